#!/usr/bin/env python3
"""Stage times (HIP events, bdpt_get_stage_times) of the bench frame; one line.  Used by tools/variants.sh."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge

pkg = ge.load_package()
scene = pkg.Scene.atrium(1, int(os.environ.get("BDPT_STAGES_TRIS", "262144")))
pipe = pkg.FramePipeline(scene, 1920, 1080, max_depth=8, mat_index=0, accum_limit=10000)
pipe.ctx.enable_stage_timing(True)
agg = {}
N = 6
for it in range(N + 2):
    pipe.render_frame()
    st = pipe.ctx.stage_times()
    if it >= 2:
        for k, v in st:
            agg[k] = agg.get(k, 0.0) + v / N
pipe.render_frame(extra_flags=pkg.abi.PARAM_COUNTERS)
c = pipe.ctx.counters().as_dict()
info = pipe.ctx.bvh_info()
tot = sum(agg.values())
side = {k: v for k, v in agg.items() if k.startswith("side:")}  # kernels of the second stream: beside the stages, not in the sum
for k in side:
    del agg[k]
tot = sum(agg.values())
gen = agg.get("gen_nee", 0) + agg.get("splat_wait", 0)
print("frame %.2f ms | walk %.2f gen_nee+wait %.2f trace_terms %.2f connect_wait %.2f trace_pairs %.2f lazy %.2f other %.2f | side: gen_splat %.2f gen_connect %.2f | rays %.1fM" % (
    tot, agg.get("walk", 0), gen, agg.get("trace_terms", 0), agg.get("connect_wait", 0), agg.get("trace_pairs", 0),
    agg.get("lazy_gen", 0) + agg.get("lazy_trace", 0) + agg.get("lazy_check", 0),
    tot - gen - sum(agg.get(k, 0) for k in ("walk", "trace_terms", "connect_wait", "trace_pairs", "lazy_gen", "lazy_trace", "lazy_check")),
    side.get("side:gen_splat", 0), side.get("side:gen_connect", 0),
    sum(c[k] for k in ("raysEyeExtend", "raysLightExtend", "raysNee", "raysSplat", "raysConnect")) / 1e6),
    "| visits/ray closest %.2f nodes %.2f tris, shadow %.2f nodes %.2f tris | sah %.2f nodes %d" % (
        c["nodeVisitsClosest"] / max(1, c["raysEyeExtend"] + c["raysLightExtend"]), c["triTestsClosest"] / max(1, c["raysEyeExtend"] + c["raysLightExtend"]),
        c["nodeVisitsShadow"] / max(1, c["raysNee"] + c["raysSplat"] + c["raysConnect"]), c["triTestsShadow"] / max(1, c["raysNee"] + c["raysSplat"] + c["raysConnect"]),
        info.sahCost, info.numNodes),
    "| any-hit NEE %.2fM + %.2fM hinted, splat %.2fM + %.2fM hinted, connect %.2fM (lazy %.2fM)" % (
        c["raysNee"] / 1e6, c["hintedNee"] / 1e6, c["raysSplat"] / 1e6, c["hintedSplat"] / 1e6, c["raysConnect"] / 1e6, c["raysConnectLazy"] / 1e6))
