"""One class of shadow rays at a time on the bench frame (run under rocprofv3 --kernel-trace --stats): what a generator
kernel and its trace kernel take when nothing else of the frame runs beside them.  `python tools/solo_stage.py nee|splat|connect|all`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge

pkg = ge.load_package()
a = pkg.abi
which = sys.argv[1] if len(sys.argv) > 1 else "connect"
flags = {"nee": a.PARAM_NO_SPLAT | a.PARAM_NO_CONNECT, "splat": a.PARAM_NO_NEE | a.PARAM_NO_CONNECT,
         "connect": a.PARAM_NO_NEE | a.PARAM_NO_SPLAT, "all": 0}[which]
scene = pkg.Scene.atrium(1, 262144)
pipe = pkg.FramePipeline(scene, 1920, 1080, max_depth=8, mat_index=0, accum_limit=10000, flags=flags)
for _ in range(8):
    pipe.render_frame()
    torch.cuda.synchronize()
print("done", which)
