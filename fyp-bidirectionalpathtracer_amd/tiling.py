"""Multi-GPU tiling of the frame: contiguous row bands, scene replicated, and the
one exchange step of the path — summing the fixed-point splat buffers so every rank holds the
splats that landed on its own band.

The sum is over uint64 fixed-point values, so it is exact and independent of reduction order:
N ranks produce bit-identical pixels to one rank rendering the whole frame.
"""


def band_rows(height, world):
    """Rows per band; bands are padded to equal height so reduce-scatter chunks are equal."""
    return (height + world - 1) // world


def band(height, world, rank):
    rows = band_rows(height, world)
    return min(rank * rows, height), min((rank + 1) * rows, height)


def exchange_splats(dist, splat_full, splat_mine):
    """splat_full: int64[world * rows * W * 4] of this rank's splats over the (padded) full frame.
    splat_mine: int64[rows * W * 4] receives the sum over ranks of this rank's band.
    RCCL ("nccl" backend) does it as one reduce-scatter; gloo (CPU tests) has no reduce-scatter,
    so it all-reduces and slices."""
    if dist.get_backend() == "nccl":
        dist.reduce_scatter_tensor(splat_mine, splat_full, op=dist.ReduceOp.SUM)
    else:
        dist.all_reduce(splat_full, op=dist.ReduceOp.SUM)
        n = splat_mine.numel()
        r = dist.get_rank()
        splat_mine.copy_(splat_full[r * n:(r + 1) * n])
    return splat_mine


def exchange_splats_async(dist, splat_full, splat_mine):
    """Start the exchange and return a handle whose wait() orders the current stream after it (RCCL), so the
    caller can enqueue bdpt_execute_tail in between.  gloo has no stream to overlap: it runs synchronously."""
    if dist is None:  # one rank, no process group: the band is the frame
        splat_mine.copy_(splat_full[:splat_mine.numel()])
        return None
    if dist.get_backend() == "nccl":
        return dist.reduce_scatter_tensor(splat_mine, splat_full, op=dist.ReduceOp.SUM, async_op=True)
    exchange_splats(dist, splat_full, splat_mine)
    return None
