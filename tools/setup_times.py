import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
scene = pkg.Scene.courtyard(1, 10000000)
W, H, D = 3840, 2160, 16
for order in ("scene_first", "resize_first"):
    torch.cuda.synchronize()
    t0 = time.time()
    ctx = pkg.Context(0)
    t1 = time.time()
    if order == "scene_first":
        ctx.set_scene(scene.desc); torch.cuda.synchronize(); t2 = time.time()
        ctx.resize(W, H, 0, H, D); torch.cuda.synchronize(); t3 = time.time()
        print(order, "ctx %.2f set_scene %.2f resize %.2f" % (t1 - t0, t2 - t1, t3 - t2), flush=True)
    else:
        ctx.resize(W, H, 0, H, D); torch.cuda.synchronize(); t2 = time.time()
        ctx.set_scene(scene.desc); torch.cuda.synchronize(); t3 = time.time()
        print(order, "ctx %.2f resize %.2f set_scene %.2f" % (t1 - t0, t2 - t1, t3 - t2), flush=True)
    t4 = time.time()
    ctx.close() if hasattr(ctx, "close") else None
    del ctx
    torch.cuda.synchronize()
    print("  close %.2f" % (time.time() - t4), flush=True)
