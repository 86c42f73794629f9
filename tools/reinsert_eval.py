import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
for name, mk in (("even", lambda: pkg.Scene.atrium(1, 262144)), ("uneven", lambda: pkg.Scene.atrium_uneven(1, 262144))):
    for passes in (0, 30, 100):
        os.environ["BDPT_REINSERT_PASSES"] = str(passes)
        os.environ["BDPT_REINSERT_BATCH"] = "0.02"
        sc = mk()
        t0 = time.time()
        pipe = pkg.FramePipeline(sc, 1920, 1080, max_depth=8, mat_index=0, accum_limit=1 << 30)
        torch.cuda.synchronize(); setup = time.time() - t0
        for _ in range(3): pipe.render_frame(accumulate=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8): pipe.render_frame(accumulate=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 8 * 1e3
        pipe.render_frame(extra_flags=pkg.abi.PARAM_COUNTERS); torch.cuda.synchronize()
        c = pipe.ctx.counters().as_dict()
        cl = max(1, c["raysEyeExtend"] + c["raysLightExtend"]); sh = max(1, c["raysNee"] + c["raysSplat"] + c["raysConnect"])
        info = pipe.ctx.bvh_info()
        print("%s passes %3d: %.2f ms/frame (1 in flight) | closest %.2f nodes %.2f tris | shadow %.2f nodes %.2f tris | sah %.2f | set-up %.2f s" % (
            name, passes, ms, c["nodeVisitsClosest"] / cl, c["triTestsClosest"] / cl, c["nodeVisitsShadow"] / sh, c["triTestsShadow"] / sh, info.sahCost, setup), flush=True)
        pipe.close(); sc.close()
