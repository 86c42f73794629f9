"""Multi-GPU tiling of the frame: interleaved stripes of rows, scene replicated, and the one
exchange step of the path — summing the fixed-point splat buffers so every rank holds the splats that landed
on its own rows.

Rows are dealt to the ranks in stripes of ``stripe_rows(H, world)`` rows (rank r renders stripes r, r + world,
...), so cost that varies by row (sky above, geometry below) spreads evenly; a context created with
``bdpt_resize_stripes`` keeps its splat accumulators owner-major, i.e. chunk r of the buffer holds rank r's
rows in order, and ONE ``reduce_scatter_tensor(SUM)`` hands every rank exactly its own chunk.  The sum is over
uint64 fixed-point values, so it is exact and independent of reduction order: N ranks produce bit-identical
pixels to one rank rendering the whole frame.

(Contiguous row bands — ``band`` / ``band_rows`` — remain for hosts that tile that way; their splat buffers
are in plain frame order.)
"""


def stripe_rows(height, world):
    """Rows per stripe: small enough that every rank gets at least four stripes (balance to within a few
    rows), at most 8 (a stripe is only a run of rows in the tile's pixel list; size costs nothing)."""
    return max(1, min(8, height // max(1, world * 4)))


def stripes_of(height, world, rank):
    """[first, last) row ranges rank `rank` renders."""
    r = stripe_rows(height, world)
    n = (height + r - 1) // r
    return [(s * r, min(height, s * r + r)) for s in range(rank, n, world)]


def chunk_rows(height, world):
    """Rows of one rank's chunk of the owner-major splat buffer (its stripes, zero-padded to equal size)."""
    r = stripe_rows(height, world)
    n = (height + r - 1) // r
    return ((n + world - 1) // world) * r


def band_rows(height, world):
    """Rows per contiguous band; bands are padded to equal height so reduce-scatter chunks are equal."""
    return (height + world - 1) // world


def band(height, world, rank):
    rows = band_rows(height, world)
    return min(rank * rows, height), min((rank + 1) * rows, height)


def exchange_splats(dist, splat_full, splat_mine):
    """splat_full: int64[world * chunk] of this rank's splats over the whole frame, rank-major chunks.
    splat_mine: int64[chunk] receives the sum over ranks of this rank's chunk.
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
    if dist is None:  # one rank, no process group: the only chunk is the frame
        splat_mine.copy_(splat_full[:splat_mine.numel()])
        return None
    if dist.get_backend() == "nccl":
        return dist.reduce_scatter_tensor(splat_mine, splat_full, op=dist.ReduceOp.SUM, async_op=True)
    exchange_splats(dist, splat_full, splat_mine)
    return None


def gather_frame(dist, torch, image, height, world, rank):
    """Assemble the full frame from every rank's stripes ("tile framebuffers gathered").  `image` is this rank's
    full-frame [H, W, 4] tensor of which only its own rows are meaningful; returns the assembled [H, W, 4] tensor
    (on every rank)."""
    if dist is None or world == 1:
        return image
    W = image.shape[1]
    rows = chunk_rows(height, world)
    gdev = image.device if dist.get_backend() == "nccl" else torch.device("cpu")  # gloo gathers host tensors
    mine = torch.zeros(rows, W, 4, dtype=image.dtype, device=gdev)
    at = 0
    for a, b in stripes_of(height, world, rank):
        mine[at:at + (b - a)] = image[a:b].to(gdev)
        at += b - a
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    full = torch.zeros(height, W, 4, dtype=image.dtype, device=gdev)
    for r in range(world):
        at = 0
        for a, b in stripes_of(height, world, r):
            full[a:b] = parts[r][at:at + (b - a)]
            at += b - a
    return full
