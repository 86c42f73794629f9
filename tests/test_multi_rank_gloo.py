"""world_size-2 (and 3, 5, 8) gloo test of the N>1 path on CPU: stripe tiles, owner-major splat exchange, resolve, gather.

Each rank renders its interleaved stripes with the ORACLE (there is no GPU here; the oracle is the checker
and, in this test only, also the stand-in renderer), lays its fixed-point splats out owner-major, sums them through
the package's tiling.exchange_splats over torch.distributed (gloo), resolves its stripes, and the frame assembled
by tiling.gather_frame must be bit-identical to the single-rank golden image.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import importlib.util
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import oracle_binding as ob
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = ge.load_package()
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    gold = np.load(os.path.join(HERE, "golden", "oracle_golden.npz"))
    W = H = 64
    scene = pkg.Scene.cornell()
    cam = mg.camera_from_array(pkg, gold["cornell_camera"])
    gp, p = mg.frame_params(pkg, 0, 3, 0)
    # this rank's tile = interleaved stripes (what bench.py's N > 1 path renders); the oracle takes one row range at a
    # time, so there is one OracleRender per stripe and their frame-order splat buffers add up
    t = pkg.tiling
    mine_rows = t.stripes_of(H, world, rank)
    chunk = t.chunk_rows(H, world) * W * 4
    orcs = []
    frame_splat = np.zeros((W * H, 4), np.uint64)
    for (ya, yb) in mine_rows:
        orc = ob.OracleRender(pkg.abi, scene.desc, W, H, ya, yb)
        orc.gbuffer(cam, gp, threads=1 if world > 4 else 2)
        orc.bdpt(cam, p, threads=1 if world > 4 else 2)
        frame_splat += orc.splat
        orcs.append(orc)
    # frame order -> owner-major (the layout bdpt_resize_stripes gives the HIP path's splat buffer)
    full = torch.zeros(world * chunk, dtype=torch.int64)
    fs = torch.from_numpy(frame_splat.reshape(H, W * 4).view(np.int64).copy())
    for r in range(world):
        at = r * chunk
        for (ya, yb) in t.stripes_of(H, world, r):
            n = (yb - ya) * W * 4
            full[at:at + n] = fs[ya:yb].reshape(-1)
            at += n
    mine = torch.zeros(chunk, dtype=torch.int64)
    if world == 3:  # the entry point bench.py uses; synchronous on gloo, so it returns no handle
        assert t.exchange_splats_async(dist, full, mine) is None
    else:
        t.exchange_splats(dist, full, mine)
    image = torch.zeros(H, W, 4, dtype=torch.float32)
    at = 0
    for orc, (ya, yb) in zip(orcs, mine_rows):
        n = (yb - ya) * W * 4
        orc.splat[:] = 0
        orc.splat.reshape(-1)[ya * W * 4: ya * W * 4 + n] = mine.numpy().view(np.uint64)[at:at + n]
        at += n
        orc.resolve()
        image[ya:yb] = torch.from_numpy(orc.image()[ya:yb].copy())
        orc.close()
    gathered = t.gather_frame(dist, torch, image, H, world, rank)  # "tile framebuffers gathered"
    if rank == 0:
        img = gathered.numpy()
        ok = np.array_equal(img.view(np.uint32), gold["cornell64_d3_ggx_image"].view(np.uint32))
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 5, 8])  # 8 = the driver's scaling run (8 CPUs here: one oracle thread per rank would do)
def test_two_rank_tiled_frame_equals_single_rank(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(180)
        assert pr.exitcode == 0
    assert q.get(timeout=5) is True
