#!/usr/bin/env python3
"""Frame time and ray rate of the other BASELINE shapes on ONE GPU (stand-in generators at the configs' triangle
counts): configs[1] Cornell box 1920x1080 depth 8 Lambertian; configs[3] 2.8 M triangles 3840x2160 depth 12 (atrium
generator); configs[4] 10 M triangles, half of them alpha-masked cards, 3840x2160 depth 16 (courtyard generator).
Usage: python tools/big_configs.py [2|4|5|both]   (numbers = BASELINE.json configs, 1-based)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge

pkg = ge.load_package()


def run(name, scene, W, H, D, frames=3, mat=0):
    t0 = time.time()
    pipe = pkg.FramePipeline(scene, W, H, max_depth=D, mat_index=mat, accum_limit=10000)
    torch.cuda.synchronize()
    setup = time.time() - t0
    pipe.ctx.enable_stage_timing(True)
    pipe.render_frame()
    torch.cuda.synchronize()
    agg = {}
    t0 = time.time()
    for _ in range(frames):
        pipe.render_frame()
        for k, v in pipe.ctx.stage_times():
            agg[k] = agg.get(k, 0.0) + v / frames
    torch.cuda.synchronize()
    dt = (time.time() - t0) / frames
    pipe.render_frame(extra_flags=pkg.abi.PARAM_COUNTERS)
    c = pipe.ctx.counters().as_dict()
    hinted = c["hintedNee"] + c["hintedSplat"]  # any-hit queries answered by their occluder hint: queries, but not traversals
    rays = sum(c[k] for k in ("raysPrimary", "raysEyeExtend", "raysLightExtend", "raysNee", "raysSplat", "raysConnect")) + hinted
    shadow = max(1, c["raysNee"] + c["raysSplat"] + c["raysConnect"])
    closest = max(1, c["raysEyeExtend"] + c["raysLightExtend"])
    info = pipe.ctx.bvh_info()
    free, total = torch.cuda.mem_get_info()
    print("%s: %d triangles %dx%d depth %d | set-up %.1f s (%s) | %.1f ms/frame, %.0f Mrays/s, %.0f M rays/frame | stages %s | "
          "visits/ray closest %.1f nodes %.1f tris %.2f alpha tests, shadow %.1f nodes %.1f tris %.2f alpha tests | hinted %.1f M of %.1f M NEE, %.1f M of %.1f M light-tracing queries | bvh %d nodes depth %d | device memory in use %.1f GB" % (
              name, scene.desc.numTriangles, W, H, D, setup, ", ".join("%s %.2f" % (k, v) for k, v in pipe.setup_times.items()), dt * 1e3, rays / dt / 1e6, rays / 1e6,
              {k: round(v, 1) for k, v in agg.items() if v >= 0.05},
              c["nodeVisitsClosest"] / closest, c["triTestsClosest"] / closest, c["alphaTestsClosest"] / closest,
              c["nodeVisitsShadow"] / shadow, c["triTestsShadow"] / shadow, c["alphaTestsShadow"] / shadow,
              c["hintedNee"] / 1e6, (c["hintedNee"] + c["raysNee"]) / 1e6, c["hintedSplat"] / 1e6, (c["hintedSplat"] + c["raysSplat"]) / 1e6,
              info.numNodes, info.maxDepth, (total - free) / 2 ** 30), flush=True)
    pipe.close()


which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which == "2":
    s = pkg.Scene.cornell()
    run("configs[1] (Cornell 1080p depth 8, Lambertian)", s, 1920, 1080, 8, frames=8, mat=1)
    s.close()
if which in ("4", "both"):
    s = pkg.Scene.atrium(1, 2800000)
    run("configs[3] shape (Bistro-class)", s, 3840, 2160, 12)
    s.close()
if which in ("5", "both"):
    s = pkg.Scene.courtyard(2, 10000000, 0.5)
    run("configs[4] shape (San-Miguel-class)", s, 3840, 2160, 16)
    s.close()
