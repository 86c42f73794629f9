#!/usr/bin/env python3
"""One rank's tile of the N-GPU frame loop on ONE GPU (the package's tiling.TileRenderer without a process group): what
a rank does per frame apart from the exchange itself.  Meant to run under rocprofv3 --kernel-trace (tools/prof_tile.sh).
Usage: python tools/tile_loop.py [--world 8] [--rank 0] [--inflight 3] [--frames 24] [--warmup 6]
Prints one JSON line: ms per frame of the timed frames (host clock around a synchronised region)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--inflight", type=int, default=3)
ap.add_argument("--frames", type=int, default=24)
ap.add_argument("--warmup", type=int, default=6)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--depth", type=int, default=8)
ap.add_argument("--triangles", type=int, default=262144)
args = ap.parse_args()
pkg = ge.load_package()
scene = pkg.Scene.atrium(1, args.triangles)
R = pkg.tiling.TileRenderer(scene, args.width, args.height, args.depth, 0, 0, args.world, args.rank, None, args.inflight)
for _ in range(args.warmup):
    R.step()
R.barrier()
t0 = time.perf_counter()
for _ in range(args.frames):
    R.step()
R.barrier()
ms = (time.perf_counter() - t0) / args.frames * 1e3
print(json.dumps({"world": args.world, "rank": args.rank, "rows": sum(b - a for a, b in R.rows), "frames_in_flight": args.inflight,
                  "frames": args.frames, "lazy_rounds_env": os.environ.get("BDPT_LAZY_ROUNDS"), "ms_per_frame": round(ms, 3)}), flush=True)
R.close()
