"""Random scenes through both builders: bdpt_bvh_recs_hash of the device pipeline against the host pipeline's — soups of
random size (1 .. 80 000 triangles, so that the collapse's deferral threshold of 4 x 2^15 binary nodes is crossed both
ways), clustered / flat / duplicated geometry, alpha-masked cards with random textures, tilings and split budgets.
`python tools/device_build_fuzz.py [cases] [seed]` on the GPU box; prints every mismatch and a summary."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as ge  # noqa: E402
import device_tree_check as chk  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    pkg = ge.load_package()
    lib = pkg.load_library()
    bad = 0
    for i in range(cases):
        kind = rng.integers(0, 6)
        env = {}
        if kind <= 2:
            n = int(rng.choice([rng.integers(1, 40), rng.integers(40, 3000), rng.integers(3000, 80000), rng.integers(60000, 70000)]))
            if kind == 0:
                c = rng.uniform(-1, 1, (n, 3))
            elif kind == 1:  # clusters + duplicates
                k = int(rng.integers(1, 6))
                c = rng.normal(0, 0.01, (n, 3)) + rng.uniform(-3, 3, (k, 3))[rng.integers(0, k, n)]
                c[: n // 3] = c[0]
            else:  # flat
                c = np.stack([rng.uniform(-5, 5, n), rng.uniform(-5, 5, n) * (rng.random() < 0.5), np.zeros(n)], 1)
            size = float(rng.choice([0.0, 1e-4, 0.02, 0.5]))
            pos = (c[:, None, :] + rng.normal(0, 1, (n, 3, 3)) * size).astype(np.float32).reshape(-1, 3)
            sc = chk.RawScene(pkg, pos, np.arange(n * 3, dtype=np.uint32).reshape(n, 3))
            if rng.random() < 0.3:
                env = {"BDPT_SPLIT_BUDGET": str(rng.choice([0.3, 1, 3]))}
            label = "soup kind %d n %d size %g %s" % (kind, n, size, env)
        else:
            n = int(rng.choice([20000, 50000, 100000, 200000]))
            f = float(rng.choice([0.2, 0.5, 0.8]))
            sc = pkg.Scene.courtyard(int(rng.integers(1, 1000)), n, f)
            env = {"BDPT_SPLIT_BUDGET_ALPHA": str(rng.choice([0, 1, 4, 9]))}
            if rng.random() < 0.3:
                env["BDPT_SPLIT_BUDGET"] = "0.5"
            label = "courtyard n %d foliage %g %s" % (n, f, env)
        os.environ.update(env)
        got = []
        for dev in (-1, 0):
            h = C.c_uint64()
            info = pkg.abi.BvhInfo()
            rc = lib.bdpt_bvh_recs_hash(C.byref(sc.desc), dev, C.byref(h), C.byref(info))
            got.append((rc, h.value, info.numNodes, info.maxDepth, info.maxStack, info.sahCost, info.numReferences, info.numDropped, info.reserved))
        for k in env:
            del os.environ[k]
        if got[0] != got[1] or got[0][0] != 0:
            bad += 1
            print("MISMATCH", label, got, flush=True)
        sc.close()
    print("%d cases, %d mismatches" % (cases, bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
