"""GPU parity tests on the BASELINE.json configurations.

Where the oracle finishes in seconds the HIP path is compared with it bit for bit (same seeds, same frame
counters) and gated on north_star's RMSE <= 1e-3; at the configurations' full sizes the checks are the
size-independent properties of the path: row bands + the exact integer splat sum reproduce the full frame,
every value is finite, and the ray tallies stay within the (D+1)^2 bound.  The assets the configs name
(Sponza, Bistro, San Miguel) are in neither the reference tree nor this image; the seeded procedural
stand-ins have their triangle counts, material classes and alpha-masked foliage (host/Atrium.cpp).
"""
import ctypes as C
import os

import numpy as np
import pytest

from test_gpu_parity import RMSE_TOL, _bands_equal_full, _oracle_frame, _rmse

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RAY_KEYS = ("raysPrimary", "raysEyeExtend", "raysLightExtend", "raysNee", "raysSplat", "raysConnect")


def _assert_frame_equals_oracle(pkg, ob, scene, pipe, gp, p, label):
    orc, cnt = _oracle_frame(pkg, ob, scene, pipe, gp, p)
    orc.resolve()
    gpu = pipe.output.cpu().numpy()[pipe.y0:pipe.y1]
    ref = orc.image()[pipe.y0:pipe.y1]
    rmse = _rmse(gpu[..., :3], ref[..., :3])
    assert rmse <= RMSE_TOL, (label, rmse)
    assert np.array_equal(gpu.view(np.uint32), ref.view(np.uint32)), \
        f"{label}: {(gpu != ref).any(axis=-1).sum()} pixels differ, rmse {rmse:.3e}"
    c, o = pipe.ctx.counters().as_dict(), cnt.as_dict()
    # identical walks; the GPU path only skips shadow rays whose outcome cannot change the image
    assert c["raysEyeExtend"] == o["raysEyeExtend"] and c["raysLightExtend"] == o["raysLightExtend"], label
    assert c["raysSplat"] <= o["raysSplat"] and c["raysNee"] <= o["raysNee"] and c["raysConnect"] <= o["raysConnect"], label
    assert c["pixelsValid"] == o["pixelsValid"] and c["splatsLanded"] == o["splatsLanded"], label
    orc.close()
    return c, o


def test_config3_textured_ggx_depth8_matches_oracle(pkg, ob):
    """configs[2]: the bench scene itself (262,144 triangles, textured GGX, three lights) at depth 8, at a size the
    oracle renders in seconds (192x108 = the 1080p frame sub-sampled by ten)."""
    import torch
    scene = pkg.Scene.atrium(1, 262144)
    pipe = pkg.FramePipeline(scene, 192, 108, max_depth=8, mat_index=0)
    for frame in range(2):
        gp, p = pipe.render_frame()
        torch.cuda.synchronize()
        c, o = _assert_frame_equals_oracle(pkg, ob, scene, pipe, gp, p, f"config3 frame {frame}")
        assert c["pixelsValid"] > 0.5 * 192 * 108
        assert sum(o[k] for k in RAY_KEYS) <= 192 * 108 * 81  # (D+1)^2 per pixel
    pipe.close()
    scene.close()


def test_heavy_tailed_atrium_matches_oracle(pkg, ob):
    """The stress variant of the stand-in (bdpt_scene_create_atrium_uneven: two-triangle walls and floors of up to
    180 m^2 beside ornaments of square millimetres, bench.py's "stress" line): a tree with leaves six decades apart in
    size answers every query as the oracle's own tree does — frames bit-identical at depth 6, both material models."""
    import torch
    scene = pkg.Scene.atrium_uneven(1, 60000)
    for mat in (0, 1):
        pipe = pkg.FramePipeline(scene, 128, 72, max_depth=6, mat_index=mat)
        gp, p = pipe.render_frame()
        torch.cuda.synchronize()
        c, _ = _assert_frame_equals_oracle(pkg, ob, scene, pipe, gp, p, f"uneven atrium mat {mat}")
        assert c["pixelsValid"] > 0.5 * 128 * 72
        pipe.close()
    scene.close()


def test_config3_bench_frame_bands_equal_full_frame(pkg):
    """configs[2] at full size — exactly the frame bench.py times (1920x1080, depth 8, 262,144 triangles)."""
    scene = pkg.Scene.atrium(1, 262144)
    W, H, D = 1920, 1080, 8
    img, cnt = _bands_equal_full(pkg, scene, W, H, D, 0, [(0, 400), (400, 1080)])
    assert np.isfinite(img).all() and img[..., :3].mean() > 0.01
    rays = sum(cnt[k] for k in RAY_KEYS)
    assert 0 < rays <= W * H * (D + 1) ** 2
    assert cnt["pixelsValid"] > 0.5 * W * H
    scene.close()


def test_config2_cornell_1080p_depth8_lambert_accumulated(pkg, ob):
    """configs[1]: Cornell box 1920x1080, depth 8, Lambertian, frames accumulated through bdpt_accumulate.
    A band of rows of the full-size frame (tile context) against the oracle over the same band, 24 frames of the
    0x1337+k / kMSAA jitter sequence (three turns of the jitter table), accumulated image compared bit for bit after
    every frame: a frame sequence that is bit-identical frame by frame stays so at any sample count."""
    import torch
    scene = pkg.Scene.cornell()
    W, H, D, frames = 1920, 1080, 8, 24
    y0, y1 = 524, 532
    pipe = pkg.FramePipeline(scene, W, H, max_depth=D, mat_index=1, tile=(y0, y1), accum_limit=256)
    lib = ob.load_oracle(pkg.abi)
    last_o = np.zeros(((y1 - y0) * W, 4), np.float32)
    for k in range(frames):
        gp, p = pipe.render_frame(accumulate=True)
        torch.cuda.synchronize()
        orc, _ = _oracle_frame(pkg, ob, scene, pipe, gp, p)
        orc.resolve()
        cur_o = orc.image()[y0:y1].reshape(-1, 4).copy()
        lib.oracle_accumulate(last_o.ctypes.data, cur_o.ctypes.data, k, 256, cur_o.shape[0])
        gpu = pipe.output.cpu().numpy()[y0:y1].reshape(-1, 4)
        assert _rmse(gpu[:, :3], cur_o[:, :3]) <= RMSE_TOL
        assert np.array_equal(gpu.view(np.uint32), cur_o.view(np.uint32)), (k, int((gpu != cur_o).any(axis=-1).sum()))
        assert np.array_equal(pipe.last_frame.cpu().numpy()[y0:y1].reshape(-1, 4).view(np.uint32), last_o.view(np.uint32)), k
        orc.close()
    assert pipe.accum_count == frames
    pipe.close()
    scene.close()


def test_config2_full_size_bands_and_stripes_equal_full_frame(pkg):
    """configs[1] at full size (Cornell 1920x1080, depth 8, Lambertian): two bands and four owners' stripes reproduce
    the one-context frame; rays within the (D+1)^2 bound."""
    scene = pkg.Scene.cornell()
    W, H, D = 1920, 1080, 8
    img, cnt = _bands_equal_full(pkg, scene, W, H, D, 1, [(0, 500), (500, 1080)])
    assert np.isfinite(img).all() and img[..., :3].mean() > 0.01
    assert 0 < sum(cnt[k] for k in RAY_KEYS) <= W * H * (D + 1) ** 2
    ref = _stripes_equal_full(pkg, scene, W, H, D, 1, 4)
    assert np.array_equal(ref.view(np.uint32), img.view(np.uint32))
    scene.close()


def test_config5_alpha_masked_foliage_depth16_matches_oracle(pkg, ob):
    """configs[4] arithmetic: depth 16 on a textured scene in which 60 % of the triangles are alpha-masked leaf
    cards (any-hit alpha test on extension, NEE, splat and connection rays), against the oracle."""
    import torch
    scene = pkg.Scene.courtyard(3, 40000, 0.6)
    pipe = pkg.FramePipeline(scene, 80, 45, max_depth=16, mat_index=0)
    gp, p = pipe.render_frame()
    torch.cuda.synchronize()
    c, o = _assert_frame_equals_oracle(pkg, ob, scene, pipe, gp, p, "config5 foliage depth 16")
    assert sum(o[k] for k in RAY_KEYS) <= 80 * 45 * 17 ** 2
    # foliage really is in the way: some primary hits carry the leaf material's opacity-tested surface
    assert c["pixelsValid"] > 0.5 * 80 * 45
    pipe.close()
    scene.close()


def test_config5_full_size_bands_equal_full_frame(pkg):
    """configs[4] shape: 10 M triangles (half of them alpha-masked foliage), 3840x2160, depth 16 — the
    size-independent checks at full size (device memory: about 90 GB)."""
    scene = pkg.Scene.courtyard(1, 10_000_000, 0.5)
    W, H, D = 3840, 2160, 16
    img, cnt = _bands_equal_full(pkg, scene, W, H, D, 0, [(0, H // 2), (H // 2, H)])
    assert np.isfinite(img).all() and img[..., :3].mean() > 0.005
    rays = sum(cnt[k] for k in RAY_KEYS)
    assert 0 < rays <= W * H * (D + 1) ** 2
    assert cnt["pixelsValid"] > 0.5 * W * H
    scene.close()


def test_two_contexts_two_threads_one_process(pkg, ob):
    """contexts are independent.  Two contexts alive in one process, driven from two host
    threads on their own streams with different scenes, sizes and depths, frames interleaved; each reproduces what
    it renders alone, and each one's counters are its own."""
    import threading
    import torch
    sa, sb = pkg.Scene.atrium(7, 20000), pkg.Scene.cornell()
    cfg = {"a": (sa, 160, 90, 6, 0), "b": (sb, 96, 96, 4, 1)}
    ref = {}
    for name, (scene, W, H, D, mat) in cfg.items():
        pipe = pkg.FramePipeline(scene, W, H, max_depth=D, mat_index=mat)
        imgs = []
        for _ in range(4):
            pipe.render_frame()
            torch.cuda.synchronize()
            imgs.append((pipe.output.cpu().numpy().copy(), pipe.ctx.counters().as_dict()))
        ref[name] = imgs
        pipe.close()
    pipes = {name: pkg.FramePipeline(scene, W, H, max_depth=D, mat_index=mat) for name, (scene, W, H, D, mat) in cfg.items()}
    errors = []
    gate = threading.Barrier(2)

    def run(name):
        try:
            pipe = pipes[name]
            stream = torch.cuda.Stream()
            for k in range(4):
                gate.wait(timeout=60)
                with torch.cuda.stream(stream):
                    pipe.render_frame()
                stream.synchronize()
                img, cnt = pipe.output.cpu().numpy(), pipe.ctx.counters().as_dict()
                if not np.array_equal(img.view(np.uint32), ref[name][k][0].view(np.uint32)):
                    errors.append((name, k, "image"))
                if cnt != ref[name][k][1]:
                    errors.append((name, k, "counters"))
        except Exception as e:  # noqa: BLE001 - surfaced through the assert below
            errors.append((name, repr(e)))

    ts = [threading.Thread(target=run, args=(n,)) for n in cfg]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not errors, errors
    for pp in pipes.values():
        pp.close()
    sa.close()
    sb.close()


def test_prepare_makes_builtin_primary_stage_and_bmfr_capturable(pkg):
    """bdpt_prepare allocates the optional buffers up front; without it a capturing stream is refused
    (BDPT_E_STATE) instead of having its capture invalidated by a hipMalloc."""
    import torch
    scene = pkg.Scene.atrium(4, 15000)
    pipe = pkg.FramePipeline(scene, 96, 64, max_depth=4, mat_index=0)
    p = pipe.bdpt_params()
    lib = pkg.load_library()
    out_ptr = C.c_void_p(pipe.output.data_ptr())
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        graph.capture_begin()
        pipe.last_frame.zero_()  # something to capture: the refused call below enqueues nothing
        rc = lib.bdpt_execute(pipe.ctx._h, C.byref(p), None, out_ptr, C.c_void_p(side.cuda_stream))
        graph.capture_end()
    assert rc == -2, rc  # BDPT_E_STATE: not prepared
    assert b"bdpt_prepare" in lib.bdpt_last_error(pipe.ctx._h)
    del graph
    pipe.ctx.prepare(pkg.abi.PREPARE_PRIMARY)
    assert lib.bdpt_execute(pipe.ctx._h, C.byref(p), None, out_ptr, pipe._stream_ptr()) == 0
    torch.cuda.synchronize()
    ref = pipe.output.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        graph.capture_begin()
        rc = lib.bdpt_execute(pipe.ctx._h, C.byref(p), None, out_ptr, C.c_void_p(side.cuda_stream))
        graph.capture_end()
    assert rc == 0
    pipe.output.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(pipe.output, ref)
    del graph
    pipe.close()
    scene.close()


def test_rccl_exchange_path_single_rank_reproduces_plain_run(pkg, tmp_path):
    """The N>1 frame loop on the real `nccl` (= RCCL) backend: one rank launched by torch.distributed.run with
    BDPT_BENCH_TILED_AT_1=1 takes bench.py's tiled path — caller-owned int64 splat tensors, two-phase execute,
    dist.reduce_scatter_tensor(int64, async_op=True) issued between the phases, work.wait() ordering the render
    stream after RCCL's, bdpt_execute_tail overlapped, resolve of the reduced band, three frames in flight — and
    must accumulate the image of the plain single-context loop bit for bit.  (More than one RCCL rank needs more
    than one GPU; the driver's 8-GPU run is the only place that happens.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--dump-frames", "5", "--width", "480", "--height", "270", "--depth", "6", "--triangles", "40000"]
    plain, tiled = tmp_path / "plain.npy", tmp_path / "rccl.npy"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--plain-loop", "--dump-path", str(plain)] + common,
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, BDPT_BENCH_TILED_AT_1="1")
    env.pop("BDPT_BENCH_BACKEND", None)
    env.pop("BDPT_BENCH_DEVICE", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "1", "--dump-path", str(tiled)] + common,
                       capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    info = json.loads(line)
    assert info["frames_in_flight"] == 3 and info.get("backend") == "nccl", info
    a, b = np.load(plain), np.load(tiled)
    assert a.shape == (270, 480, 4) and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.isfinite(a).all() and a[..., :3].mean() > 0.01


def _stripes_equal_full(pkg, scene, W, H, depth, mat, world, flags=0):
    """`world` contexts rendering interleaved stripes (bdpt_resize_stripes) with owner-major splat buffers; the
    integer sum of the buffers (what the reduce-scatter computes), each owner's chunk resolved with
    bdpt_resolve_tile, reproduces the single-context frame bit for bit."""
    import torch
    full = pkg.FramePipeline(scene, W, H, max_depth=depth, mat_index=mat, flags=flags)
    full.ctx.set_environment(None, 0, 0, (0.6, 0.5, 0.4, 1.0))
    full.render_frame()
    torch.cuda.synchronize()
    ref = full.output.cpu().numpy()
    full.close()
    R = pkg.tiling.stripe_rows(H, world)
    pipes = [pkg.FramePipeline(scene, W, H, max_depth=depth, mat_index=mat, stripes=(R, world, r), flags=flags) for r in range(world)]
    total = None
    for r, pp in enumerate(pipes):
        pp.ctx.set_environment(None, 0, 0, (0.6, 0.5, 0.4, 1.0))
        assert pp.rows == pkg.tiling.stripes_of(H, world, r)
        info = pp.ctx.tile_info()
        assert info.chunkRows == pkg.tiling.chunk_rows(H, world) and info.splatU64 == world * info.chunkU64
        buf = torch.zeros(info.splatU64, dtype=torch.int64, device="cuda")
        pp.ctx.set_splat_buffer(C.c_void_p(buf.data_ptr()), buf.numel())
        _, p = pp.render_frame(extra_flags=pkg.abi.PARAM_DEFER_RESOLVE | pkg.abi.PARAM_DEFER_TAIL)
        torch.cuda.synchronize()
        snap = buf.clone()
        pp.ctx.execute_tail(p, pp.gb, C.c_void_p(pp.output.data_ptr()), pp._stream_ptr())
        torch.cuda.synchronize()
        assert torch.equal(snap, buf)  # phase 2 must not touch the splat buffer
        total = snap if total is None else total + snap
        pp._keep = buf
    img = np.zeros_like(ref)
    for r, pp in enumerate(pipes):
        n = pp.ctx.tile_info().chunkU64
        chunk = total[r * n:(r + 1) * n].contiguous()
        pp.ctx.resolve_tile(C.c_void_p(chunk.data_ptr()), C.c_void_p(pp.output.data_ptr()))
        torch.cuda.synchronize()
        out = pp.output.cpu().numpy()
        for a, b in pp.rows:
            img[a:b] = out[a:b]
        pp.close()
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), f"{(img != ref).any(axis=-1).sum()} pixels differ"
    return ref


def test_striped_tiles_equal_full_frame(pkg):
    """Interleaved stripes incl. a partial last stripe, more owners than stripes (an owner with an EMPTY tile), and
    the 8-owner split of a 1080p-shaped frame."""
    cornell, atrium = pkg.Scene.cornell(), pkg.Scene.atrium(2, 20000)
    _stripes_equal_full(pkg, cornell, 40, 37, 4, 0, 3)
    # the environment / emissive terms of the eye walk (BDPT_PARAM_ENV_ON_MISS | _EMISSIVE_HITS) go to the tile's own pixels
    a = _stripes_equal_full(pkg, atrium, 64, 45, 5, 0, 3, flags=pkg.abi.PARAM_ENV_ON_MISS | pkg.abi.PARAM_EMISSIVE_HITS)
    b = _stripes_equal_full(pkg, atrium, 64, 45, 5, 0, 3)
    assert not np.array_equal(a, b)
    _stripes_equal_full(pkg, cornell, 24, 3, 3, 1, 4)       # 3 rows, 4 owners: owner 3 renders nothing
    _stripes_equal_full(pkg, atrium, 192, 108, 5, 0, 8)
    cornell.close()
    atrium.close()


def test_tile_pack_all_gather_unpack_is_the_identity_for_every_world_size(pkg):
    """bdpt_tile_pack / bdpt_tile_unpack — the device side of "tile framebuffers gathered" that the tiled denoiser uses every
    frame — for worlds of 1, 2, 3, 5 and 8 ranks on one GPU: every rank's stripe context packs its rows of a whole-frame
    buffer into its chunk (chunkRows x W, zero padded), the chunks are laid out rank-major as ncclAllGather leaves them,
    and ONE context unpacks every owner's chunk: the frame must come back bit for bit, for 16-, 8- and 4-byte pixels and
    frames whose last stripe is short or whose last ranks have no rows (RCCL with more than one rank cannot run here; the
    index maths of the world > 1 path is what can be pinned)."""
    import torch
    scene = pkg.Scene.cornell()
    lib = pkg.load_library()
    for (W, H) in ((40, 37), (24, 3), (64, 45)):
        for world in (1, 2, 3, 5, 8):
            R = pkg.tiling.stripe_rows(H, world)
            chunk_rows = pkg.tiling.chunk_rows(H, world)
            pipes = [pkg.FramePipeline(scene, W, H, max_depth=2, mat_index=1, stripes=(R, world, r)) for r in range(world)]
            for bpp, dt in ((16, torch.float32), (8, torch.float16), (4, torch.int32)):
                ch = {16: 4, 8: 4, 4: 1}[bpp]
                frame = torch.arange(W * H * ch, device="cuda").remainder(30011).to(dt).reshape(H, W, ch).contiguous()
                gathered = torch.full((world, chunk_rows, W, ch), 77, dtype=dt, device="cuda")
                for r, pp in enumerate(pipes):
                    assert pp.ctx.tile_info().chunkRows == chunk_rows
                    gathered[r].zero_()
                    assert lib.bdpt_tile_pack(pp.ctx._h, C.c_void_p(frame.data_ptr()), C.c_void_p(gathered[r].data_ptr()), bpp, pp._stream_ptr()) == 0
                back = torch.full_like(frame, 55)
                for r in range(world):
                    assert lib.bdpt_tile_unpack(pipes[0].ctx._h, r, C.c_void_p(gathered[r].data_ptr()), C.c_void_p(back.data_ptr()), bpp, pipes[0]._stream_ptr()) == 0
                torch.cuda.synchronize()
                assert torch.equal(back.view(torch.uint8), frame.view(torch.uint8)), (W, H, world, bpp)
            assert lib.bdpt_tile_unpack(pipes[0].ctx._h, world, C.c_void_p(gathered.data_ptr()), C.c_void_p(back.data_ptr()), 16, None) != 0  # no such owner
            assert lib.bdpt_tile_pack(pipes[0].ctx._h, C.c_void_p(frame.data_ptr()), C.c_void_p(gathered.data_ptr()), 12, None) != 0  # 4, 8 or 16 bytes
            for pp in pipes:
                pp.close()
    scene.close()


def test_config4_full_size_stripes_equal_full_frame(pkg):
    """configs[3] shape (Bistro-class: 2.8 M triangles, 3840x2160, depth 12) tiled the way the 8-GPU run tiles it:
    eight owners' interleaved stripes + the exact integer sum of their owner-major splat buffers == the full frame."""
    scene = pkg.Scene.atrium(1, 2_800_000)
    img = _stripes_equal_full(pkg, scene, 3840, 2160, 12, 0, 8)
    assert np.isfinite(img).all() and img[..., :3].mean() > 0.01
    scene.close()


def test_execute_at_lower_depth_than_the_context_is_sized_for(pkg, ob):
    """bdpt_resize fixes the storage depth (and with it the generators' lanes-per-pixel shape: 16 above depth 8);
    gMaxDepth may then be any smaller value per frame (the reference's GUI slider, BDPTPass.cpp:64-66)."""
    import torch
    scene = pkg.Scene.atrium(8, 16000)
    for sized, depths in ((12, (12, 5, 2, 1)), (8, (8, 3))):
        pipe = pkg.FramePipeline(scene, 72, 40, max_depth=sized, mat_index=0)
        for d in depths:
            pipe.max_depth = d
            gp, p = pipe.render_frame()
            assert p.maxDepth == d
            torch.cuda.synchronize()
            orc, _ = _oracle_frame(pkg, ob, scene, pipe, gp, p)
            orc.resolve()
            gpu, ref = pipe.output.cpu().numpy(), orc.image()
            assert np.array_equal(gpu.view(np.uint32), ref.view(np.uint32)), (sized, d, int((gpu != ref).any(axis=-1).sum()))
            orc.close()
        pipe.close()
    scene.close()


def test_device_builder_builds_what_the_host_builder_builds(pkg, monkeypatch):
    """bdpt_set_scene builds the acceleration structure on the device (csrc/bvh_device.hip) — the reference's BLAS / TLAS
    builds are GPU work too (RtModel.cpp:181-254, RtScene.cpp:220-308): classification and split priorities, the references (alpha-clipped, pre-split), the
    binned-SAH binary tree, the four-wide collapse and the quantised, packed records.  Every stage has to produce the host builder's result bit for bit:
      * bdpt_bvh_build_hash with the device tree builder plugged in (bdpt_test_tree_builder) = with the host one: nodes,
        leaf-ordered triangles, reference boxes, packed records, depth / stack / SAH cost;
      * bdpt_bvh_recs_hash of the whole device pipeline = of the host pipeline: every packed record.
    Scenes: the bench scene, the heavy-tailed one, the alpha-masked one, tiny inputs (one wave per slot from the root on), a
    degenerate pile of identical centroids plus an 18-decade line (median fallback: pivot by one wave's ranking and by the radix select,
    depth budget), and builds with other split budgets (opaque outliers split too; eight splits per alpha card)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("device_tree_check", os.path.join(ROOT, "tools", "device_tree_check.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    lib = pkg.load_library()
    scenes = [("cornell", pkg.Scene.cornell(), {}), ("atrium", pkg.Scene.atrium(1, 262144), {}), ("uneven", pkg.Scene.atrium_uneven(1, 262144), {}),
              ("courtyard", pkg.Scene.courtyard(1, 200000), {}), ("soup 3", pkg.Scene.soup(2, 3), {}), ("soup 1500", pkg.Scene.soup(2, 1500), {}),
              ("soup 40000", pkg.Scene.soup(2, 40000), {}), ("skewed", chk.skewed(pkg, 50000), {}),
              ("uneven +split", pkg.Scene.atrium_uneven(1, 262144), {"BDPT_SPLIT_BUDGET": "1"}),
              ("courtyard +8", pkg.Scene.courtyard(3, 100000, 0.7), {"BDPT_SPLIT_BUDGET": "0.5", "BDPT_SPLIT_BUDGET_ALPHA": "8"})]
    try:
        for name, sc, env in scenes:
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            got, recs = [], []
            for dev in (-1, 0):
                assert lib.bdpt_test_tree_builder(dev) == 0
                h = C.c_uint64()
                info = pkg.abi.BvhInfo()
                assert lib.bdpt_bvh_build_hash(C.byref(sc.desc), 0, C.byref(h), C.byref(info)) == 0, name
                got.append((h.value, info.numNodes, info.maxDepth, info.maxStack, info.sahCost, info.numReferences))
            lib.bdpt_test_tree_builder(-1)
            for dev in (-1, 0):
                h = C.c_uint64()
                info = pkg.abi.BvhInfo()
                assert lib.bdpt_bvh_recs_hash(C.byref(sc.desc), dev, C.byref(h), C.byref(info)) == 0, name
                recs.append((h.value, info.numNodes, info.maxDepth, info.maxStack, info.sahCost, info.numReferences, info.numDropped, info.reserved))
            assert got[0] == got[1], (name, got)
            assert recs[0] == recs[1], (name, recs)
            assert recs[0][1] == got[0][1] and recs[0][5] == got[0][5], name
            for k in env:
                monkeypatch.delenv(k)
    finally:
        lib.bdpt_test_tree_builder(-1)
        for _, sc, _ in scenes:
            sc.close()


def test_a_failed_allocation_in_the_device_build_comes_back_as_an_error_code():
    """Every stage of the device build (references, tree, pack) takes its scratch from the build's arena; a request that
    fails must end bdpt_set_scene with BDPT_E_NOMEM and a message — no fault, no partial scene — and the next call in
    the same context must work.  (The hook counts requests per process, so every case runs in a child process.)"""
    import subprocess
    import sys
    code = r'''
import ctypes as C, os, sys
sys.path.insert(0, %r)
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); lib = pkg.load_library()
sc = pkg.Scene.courtyard(1, 30000)
ctx = pkg.Context(0)
rc = lib.bdpt_set_scene(ctx._h, C.byref(sc.desc))
msg = lib.bdpt_last_error(ctx._h).decode()
rc2 = lib.bdpt_set_scene(ctx._h, C.byref(sc.desc))
info = ctx.bvh_info()
print("RESULT", rc, rc2, info.numNodes > 0, "|", msg)
''' % ROOT
    for n in (1, 9, 30, 60, 75):  # requests in the reference stage, the tree stage (level arrays) and the pack stage
        env = dict(os.environ, BDPT_TEST_FAIL_DEVICE_ALLOC=str(n))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        assert line, (n, out.stdout[-500:], out.stderr[-500:])
        _, rc, rc2, ok, _, msg = line[0].split(" ", 5)
        assert (int(rc), int(rc2), ok) == (-4, 0, "True") and "out of device memory" in msg, (n, line[0])  # BDPT_E_NOMEM


def _many_cards_scene(pkg, n_cards, tex, uv_scale, rng, threshold=0.5, jitter=0.3):
    """n_cards alpha-masked unit cards (two triangles each, one shared texture) scattered and rotated in a box, plus an opaque
    floor: a small stand-in for foliage whose texture and uv tiling the caller chooses."""
    a = pkg.abi
    quad = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32)
    pos, uv, idx, tri_mat = [], [], [], []
    for k in range(n_cards):
        ang = rng.uniform(0, 2 * np.pi, 3)
        cx, sx, cy, sy = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1])
        R = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], np.float32) @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], np.float32)
        p = (quad * rng.uniform(0.2, 1.5)) @ R.T + rng.uniform(-3, 3, 3).astype(np.float32)
        base = len(pos)
        pos.extend(p.astype(np.float32))
        off = rng.uniform(-jitter, jitter, 2)
        uv.extend(np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32) * uv_scale + np.array([off[0], off[1], 0], np.float32))
        idx.extend([base, base + 1, base + 2, base, base + 2, base + 3])
        tri_mat.extend([0, 0])
    base = len(pos)
    pos.extend(np.array([[-6, -6, -4], [6, -6, -4], [6, 6, -4], [-6, 6, -4]], np.float32))
    uv.extend(np.zeros((4, 3), np.float32))
    idx.extend([base, base + 1, base + 2, base, base + 2, base + 3])
    tri_mat.extend([1, 1])
    pos, uv = np.ascontiguousarray(pos, np.float32), np.ascontiguousarray(uv, np.float32)
    nrm = np.tile(np.array([[0, 0, 1]], np.float32), (len(pos), 1))
    idx, tri_mat = np.asarray(idx, np.uint32), np.asarray(tri_mat, np.uint32)
    mats = (a.Material * 2)()
    for m, alpha_mode, tex_id in ((mats[0], 1, 0), (mats[1], 0, -1)):
        for k in range(4):
            m.baseColor[k] = 1.0
        m.alphaThreshold = threshold
        m.IoR = 1.5
        m.flags = (0 | ((2 if tex_id >= 0 else 1) << 3) | (1 << 6) | (alpha_mode << 17) | (1 << 19))
        m.texBaseColor, m.texSpecular, m.texEmissive, m.texNormal = tex_id, -1, -1, -1
    texs = (a.Texture * 1)()
    texs[0].rgba8 = tex.ctypes.data_as(C.POINTER(C.c_uint8))
    texs[0].width, texs[0].height, texs[0].srgb = tex.shape[1], tex.shape[0], 1
    lights = (a.Light * 1)()
    lights[0].posW[2] = 3.0
    lights[0].intensity[0] = lights[0].intensity[1] = lights[0].intensity[2] = 1.0
    d = a.SceneDesc()
    d.numVertices, d.numTriangles, d.numMaterials, d.numTextures, d.numLights = len(pos), len(tri_mat), 2, 1, 1
    fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
    d.positions, d.normals, d.texcoords = fp(pos), fp(nrm), fp(uv)
    d.indices = idx.ctypes.data_as(C.POINTER(C.c_uint32))
    d.triMaterial = tri_mat.ctypes.data_as(C.POINTER(C.c_uint32))
    d.materials, d.textures, d.lights = mats, texs, lights
    return d, [tex, pos, nrm, uv, idx, tri_mat, mats, texs, lights]


def test_device_builder_on_adversarial_inputs(pkg, monkeypatch):
    """The device build against the host build (packed records + summary, bdpt_bvh_recs_hash) where the code paths fork:
    reference counts on either side of the one-wave-per-node bound (1024) and of the wave sort's bound (2048), every
    triangle identical (median fallback through the radix select), geometry on a line and in a plane (one or two centroid
    axes without extent), coordinates over 24 decades, and alpha-masked cards — random blobs, a checkerboard, one opaque
    texel, tiled and shifted texture coordinates, up to 64 splits per card."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("device_tree_check", os.path.join(ROOT, "tools", "device_tree_check.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    lib = pkg.load_library()
    rng = np.random.default_rng(11)

    def soup(n, make_centres, size=0.01):
        c = make_centres(n).astype(np.float32)
        d = rng.normal(0, size, (n, 3, 3)).astype(np.float32)
        pos = (c[:, None, :] + d).reshape(-1, 3)
        return chk.RawScene(pkg, pos, np.arange(n * 3, dtype=np.uint32).reshape(n, 3))

    cases = []
    for n in (1, 2, 3, 1023, 1024, 1025, 2047, 2048, 2049, 4097):
        cases.append(("uniform %d" % n, soup(n, lambda m: rng.uniform(-1, 1, (m, 3))), {}))
    same = np.tile(np.array([[0.1, 0.2, 0.3], [0.4, 0.2, 0.3], [0.1, 0.6, 0.3]], np.float32), (5000, 1))
    cases.append(("identical 5000", chk.RawScene(pkg, same, np.arange(15000, dtype=np.uint32).reshape(5000, 3)), {}))
    cases.append(("line 3000", soup(3000, lambda m: np.stack([rng.uniform(-5, 5, m), np.zeros(m), np.zeros(m)], 1), size=0.0), {}))
    cases.append(("plane 3000", soup(3000, lambda m: np.stack([rng.uniform(-5, 5, m), rng.uniform(-5, 5, m), np.full(m, 2.0)], 1), size=0.0), {}))
    cases.append(("24 decades", soup(6000, lambda m: np.exp(rng.uniform(-27, 27, (m, 3))) * rng.choice([-1.0, 1.0], (m, 3)), size=1e-3), {}))
    blobs = np.zeros((64, 64, 4), np.uint8)
    yy, xx = np.mgrid[0:64, 0:64]
    for _ in range(12):
        cx, cy, r = rng.uniform(0, 64, 3)
        blobs[..., 3] = np.maximum(blobs[..., 3], np.where((xx - cx) ** 2 + (yy - cy) ** 2 < (4 + r / 8) ** 2, 255, 0))
    checker = np.zeros((16, 16, 4), np.uint8)
    checker[..., 3] = np.where((xx[:16, :16] + yy[:16, :16]) % 2 == 0, 255, 0)
    dot = np.zeros((32, 32, 4), np.uint8)
    dot[17, 5, 3] = 255
    keepalive = []
    for name, tex, uv_scale, env in (("cards blobs", blobs, 1.0, {}), ("cards blobs x3.7", blobs, 3.7, {"BDPT_SPLIT_BUDGET_ALPHA": "16"}),
                                     ("cards checker", checker, 1.0, {"BDPT_SPLIT_BUDGET_ALPHA": "64"}), ("cards dot", dot, 1.0, {}),
                                     ("cards dot x0.5", dot, 0.5, {"BDPT_SPLIT_BUDGET": "2", "BDPT_SPLIT_BUDGET_ALPHA": "8"})):
        d, keep = _many_cards_scene(pkg, 400, tex, uv_scale, rng)
        keepalive.append(keep)

        class Holder:
            pass
        hsc = Holder()
        hsc.desc = d
        hsc.close = lambda: None
        cases.append((name, hsc, env))
    for name, sc, env in cases:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = []
        for dev in (-1, 0):
            h = C.c_uint64()
            info = pkg.abi.BvhInfo()
            assert lib.bdpt_bvh_recs_hash(C.byref(sc.desc), dev, C.byref(h), C.byref(info)) == 0, name
            got.append((h.value, info.numNodes, info.maxDepth, info.maxStack, info.sahCost, info.numReferences, info.numDropped, info.reserved))
        assert got[0] == got[1], (name, got)
        for k in env:
            monkeypatch.delenv(k)
        sc.close()
