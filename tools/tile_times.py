#!/usr/bin/env python3
"""Per-frame time of ONE rank's tile (interleaved stripes) for world sizes 1, 2, 4, 8 on one GPU, with 1 and 3 frames in
flight: what strong scaling can be at most before any exchange cost (the exchange itself needs N GPUs).
Usage: python tools/tile_times.py [width height depth triangles]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
import bench

pkg = ge.load_package()
W, H, D, T = (int(x) for x in (sys.argv[1:5] + ["1920", "1080", "8", "262144"][len(sys.argv) - 1:]))
scene = pkg.Scene.atrium(1, T)
base = {}
for inflight in (1, 3):
    for world in (1, 2, 4, 8):
        worst = 0.0
        for rank in ((0,) if world == 1 else (0, world - 1)):
            R = pkg.tiling.TileRenderer(scene, W, H, D, 0, 0, world, rank, None, inflight)
            for _ in range(4):
                R.step()
            R.barrier()
            n = 12
            t0 = time.perf_counter()
            for _ in range(n):
                R.step()
            R.barrier()
            worst = max(worst, (time.perf_counter() - t0) / n * 1e3)
            R.close()
        base.setdefault(inflight, worst)
        print("frames in flight %d, 1/%d of the frame (%d rows): %.2f ms/frame -> speed-up bound %.2fx of %d (%.2f of ideal)" % (
            inflight, world, sum(b - a for a, b in pkg.tiling.stripes_of(H, world, 0)), worst, base[inflight] / worst, world,
            base[inflight] / worst / world), flush=True)
