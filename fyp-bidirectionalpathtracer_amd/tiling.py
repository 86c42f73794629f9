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

``TileRenderer`` is the per-rank frame loop built on these pieces (what ``bench.py --gpus N`` times and what a
Python host would run): two-phase execute, the exchange overlapped with the tail, frames in flight, the running mean
in frame order.  Its C++ counterpart is ``RenderingPipeline::setTiling`` (host/Passes.cpp) over RCCL directly.
"""
import ctypes as C


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


class TileRenderer:
    """One rank's share of the tiled frame loop (the reference's per-frame order, SharedUtils/RenderingPipeline.cpp:611-695
    — G-buffer pass, BDPT pass, accumulation pass — with the single DispatchRays of
    Falcor API/D3D12/D3D12RenderContext.cpp:350-384 replaced by this rank's stripes + one splat exchange).

    `inflight` contexts render the same tile (interleaved stripes of rows; with one rank the whole frame), each with one
    frame in flight on its own stream: the persistent launches of one frame ramp up and drain (a sub-path is up to D rays
    in series) and a tile leaves the chip underfilled, so frames overlap; they stay independent
    until the running mean, which is applied in frame order through an event chain.

    dist: an initialised ``torch.distributed`` module (backend "nccl" = RCCL, or gloo for rehearsals) or None for a
    single rank without a process group (nothing is exchanged then)."""

    def __init__(self, scene, width, height, max_depth, mat_index, device, world=1, rank=0, dist=None, inflight=3,
                 accum_limit=1 << 30):
        import torch
        from . import FramePipeline, abi
        self.torch, self.abi, self.dist = torch, abi, dist
        self.W, self.H, self.world, self.rank = int(width), int(height), int(world), int(rank)
        self.dev = torch.device("cuda", device)
        self.inflight = max(1, int(inflight))
        stripes = (stripe_rows(self.H, self.world), self.world, self.rank)
        self.pipes = [FramePipeline(scene, self.W, self.H, max_depth=max_depth, mat_index=mat_index, device=device,
                                    stripes=stripes, accum_limit=accum_limit) for _ in range(self.inflight)]
        self.pipe = self.pipes[0]
        self.ctx = self.pipe.ctx
        self.rows = self.pipe.rows
        self.num_pixels = sum(b - a for a, b in self.rows) * self.W
        self.state = {"frame": 0, "accum": 0, "accum_event": None}
        info = self.ctx.tile_info()
        self.exchange_bytes = int(info.splatU64) * 8  # what one rank hands to the reduce-scatter per frame
        self.streams = [torch.cuda.Stream(self.dev) for _ in range(self.inflight)]
        self.splat_full = [torch.zeros(info.splatU64, dtype=torch.int64, device=self.dev) for _ in range(self.inflight)]
        # without a process group nothing is exchanged: this rank's own chunk is resolved where it is (one rank: the frame)
        self.splat_mine = [self.splat_full[i][self.rank * info.chunkU64:(self.rank + 1) * info.chunkU64] if dist is None else
                           torch.zeros(info.chunkU64, dtype=torch.int64, device=self.dev) for i in range(self.inflight)]
        for pp, sf in zip(self.pipes, self.splat_full):
            pp.ctx.set_splat_buffer(C.c_void_p(sf.data_ptr()), sf.numel())
        self.last_frame = self.pipe.last_frame  # the running mean is shared by all frames in flight
        self.exchange_events = None               # a list makes step() record (tail0, tail1, ex0, ex1) events per frame

    def step(self, flags=0):
        """One pipeline frame on this rank's tile; returns the context that took it."""
        torch, abi, state = self.torch, self.abi, self.state
        f = state["frame"]
        state["frame"] += 1
        i = f % self.inflight
        pp, s = self.pipes[i], self.streams[i]
        pp.gbuffer_frame, pp.bdpt_frame = 0xdeadbeef + f, 0x1337 + f
        timing = self.exchange_events is not None
        with torch.cuda.stream(s):
            # phase 1: everything that writes the splat buffer; then the exchange starts on RCCL's stream while
            # phase 2 (zero-valued connection rounds) runs on ours
            _, p = pp.render_frame(accumulate=False, extra_flags=flags | abi.PARAM_DEFER_RESOLVE | abi.PARAM_DEFER_TAIL)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timing else None
            if timing:
                ev[2].record(s)
            work = exchange_splats_async(self.dist, self.splat_full[i], self.splat_mine[i]) if self.dist is not None else None
            st = C.c_void_p(s.cuda_stream)
            if timing:
                ev[0].record(s)
            pp.ctx.execute_tail(p, pp.gb, C.c_void_p(pp.output.data_ptr()), st)
            if timing:
                ev[1].record(s)
            if work is not None:
                work.wait()
            if timing:
                ev[3].record(s)
                self.exchange_events.append(ev)
            pp.ctx.resolve_tile(C.c_void_p(self.splat_mine[i].data_ptr()), C.c_void_p(pp.output.data_ptr()), st)
            if state["accum_event"] is not None:
                s.wait_event(state["accum_event"])  # running mean in frame order
            n = state["accum"]
            state["accum"] += 1
            pp.ctx.accumulate_tile(C.c_void_p(self.last_frame.data_ptr()), C.c_void_p(pp.output.data_ptr()), n, pp.accum_limit, st)
            done = torch.cuda.Event()
            done.record(s)
            state["accum_event"] = done
        return pp.ctx

    def barrier(self):
        self.torch.cuda.synchronize(self.dev)
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize(self.dev)

    def rewind(self, frame, accum):
        self.state["frame"], self.state["accum"] = frame, accum

    def gather(self):
        """The accumulated frame of all ranks ("tile framebuffers gathered"), on every rank."""
        self.barrier()
        return gather_frame(self.dist, self.torch, self.last_frame, self.H, self.world, self.rank)

    def close(self):
        for pp in self.pipes:
            pp.close()
