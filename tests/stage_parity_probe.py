"""(Test infrastructure, run by hand: `python tests/stage_parity_probe.py` on the GPU box.)  Which stage of the frame differs from the oracle (NEE only / splat only / connect only / all) — a debugging aid for
kernel variants: prints the number of differing pixels per case."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
from test_gpu_parity import _oracle_frame  # noqa: E402

import oracle_binding as ob  # noqa: E402  (test infrastructure: this script lives under tests/)

pkg = ge.load_package()
ob.load_oracle(pkg.abi)
scene = pkg.Scene.cornell()
a = pkg.abi
for name, depth, mat, flags in (("nee", 3, 0, a.PARAM_NO_SPLAT | a.PARAM_NO_CONNECT), ("splat", 3, 0, a.PARAM_NO_NEE | a.PARAM_NO_CONNECT),
                                ("connect", 4, 0, a.PARAM_NO_NEE | a.PARAM_NO_SPLAT), ("all d3", 3, 0, 0), ("all d8 lambert", 8, 1, 0), ("all d12 ggx", 12, 0, 0)):
    pipe = pkg.FramePipeline(scene, 64, 64, max_depth=depth, mat_index=mat, flags=flags)
    gp, p = pipe.render_frame()
    torch.cuda.synchronize()
    orc, cnt = _oracle_frame(pkg, ob, scene, pipe, gp, p)
    orc.resolve()
    gpu = pipe.output.cpu().numpy()
    bad = (gpu.view(np.uint32) != orc.image().view(np.uint32)).any(axis=-1)
    c = pipe.ctx.counters().as_dict() if False else {}
    print(name, "differing pixels:", int(bad.sum()), "of", bad.size, flush=True)
    orc.close()
    pipe.close()
