"""The device builder (csrc/bvh_device.hip) against the host one, scene by scene, with build times: bdpt_bvh_build_hash
under both settings of bdpt_test_tree_builder (the binary-tree stage alone), and bdpt_bvh_recs_hash of the whole device
pipeline (references, tree, quantise + pack) against the host pipeline's packed records.
`python tools/device_tree_check.py [big]`; DEVTREE_ONLY=<scene name> runs one scene on the device only (for profiles)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


class RawScene:
    """A scene description over numpy arrays: opaque triangles, one material, one light."""

    def __init__(self, pkg, pos, idx):
        a = pkg.abi
        n_v, n_t = len(pos), len(idx)
        self.keep = [np.ascontiguousarray(pos, np.float32), np.tile(np.array([[0, 0, 1]], np.float32), (n_v, 1)), np.zeros((n_v, 3), np.float32),
                     np.ascontiguousarray(idx, np.uint32).reshape(-1), np.zeros(n_t, np.uint32)]
        mats = (a.Material * 1)()
        for k in range(4):
            mats[0].baseColor[k] = 1.0
        mats[0].IoR = 1.5
        mats[0].flags = (1 << 3) | (1 << 6) | (1 << 19)
        mats[0].texBaseColor = mats[0].texSpecular = mats[0].texEmissive = mats[0].texNormal = -1
        lights = (a.Light * 1)()
        lights[0].posW[2] = 3.0
        lights[0].intensity[0] = lights[0].intensity[1] = lights[0].intensity[2] = 1.0
        d = a.SceneDesc()
        d.numVertices, d.numTriangles, d.numMaterials, d.numTextures, d.numLights = n_v, n_t, 1, 0, 1
        fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
        d.positions, d.normals, d.texcoords = fp(self.keep[0]), fp(self.keep[1]), fp(self.keep[2])
        d.indices = self.keep[3].ctypes.data_as(C.POINTER(C.c_uint32))
        d.triMaterial = self.keep[4].ctypes.data_as(C.POINTER(C.c_uint32))
        d.materials, d.lights = mats, lights
        self.keep += [mats, lights]
        self.desc = d

    def close(self):
        pass


def skewed(pkg, n, seed=3):
    """Half of the triangles piled on one point, a quarter on a log-uniform line over 18 decades, the rest a Gaussian
    cloud: identical centroids force the median fallback (by the wave sort and, for the pile, the host sort), the line the
    depth budget."""
    rng = np.random.default_rng(seed)
    c = np.zeros((n, 3), np.float32)
    h, q = n // 2, n // 4
    c[:h] = (0.25, 0.5, 0.75)
    c[h:h + q, 0] = np.exp(rng.uniform(-40, 3, q)).astype(np.float32)
    c[h + q:] = rng.normal(0, 1, (n - h - q, 3)).astype(np.float32)
    d = np.float32(0.01)
    pos = np.empty((n * 3, 3), np.float32)
    pos[0::3] = c + np.array((d, 0, 0), np.float32)
    pos[1::3] = c + np.array((0, d, 0), np.float32)
    pos[2::3] = c + np.array((0, 0, d), np.float32)
    return RawScene(pkg, pos, np.arange(n * 3, dtype=np.uint32).reshape(n, 3))


def main():
    pkg = ge.load_package()
    lib = pkg.load_library()
    big = len(sys.argv) > 1 and sys.argv[1] == "big"
    scenes = [("cornell", lambda: pkg.Scene.cornell()), ("atrium 262144", lambda: pkg.Scene.atrium(1, 262144)),
              ("atrium uneven", lambda: pkg.Scene.atrium_uneven(1, 262144)), ("courtyard", lambda: pkg.Scene.courtyard(1, 200000)),
              ("soup 3", lambda: pkg.Scene.soup(2, 3)), ("soup 40000", lambda: pkg.Scene.soup(2, 40000)),
              ("skewed 50000", lambda: skewed(pkg, 50000)),
              # opaque outliers split too (BDPT_SPLIT_BUDGET, read by every build), more splits per alpha card
              ("uneven +split", lambda: pkg.Scene.atrium_uneven(1, 262144), {"BDPT_SPLIT_BUDGET": "1"}),
              ("courtyard +8", lambda: pkg.Scene.courtyard(3, 100000, 0.7), {"BDPT_SPLIT_BUDGET": "0.5", "BDPT_SPLIT_BUDGET_ALPHA": "8"})]
    if big:
        scenes += [("atrium 2.8M", lambda: pkg.Scene.atrium(1, 2800000)), ("atrium 10M", lambda: pkg.Scene.atrium(1, 10000000)),
                   ("courtyard 10M", lambda: pkg.Scene.courtyard(1, 10000000))]
    only = os.environ.get("DEVTREE_ONLY")  # one scene by name, device build only (for profiles)
    if only:
        scenes = [x for x in scenes if x[0] == only]
    bad = 0
    for entry in scenes:
        name, make = entry[0], entry[1]
        env = entry[2] if len(entry) > 2 else {}
        os.environ.update(env)
        sc = make()
        if sc is None:
            print(name, "skipped")
            continue
        res = []
        for dev in ((0, 0) if only else (-1, 0)):
            assert lib.bdpt_test_tree_builder(dev) == 0
            h = C.c_uint64()
            info = pkg.abi.BvhInfo()
            t0 = time.time()
            rc = lib.bdpt_bvh_build_hash(C.byref(sc.desc), 0, C.byref(h), C.byref(info))
            res.append((rc, h.value, info.numNodes, info.maxDepth, info.maxStack, info.sahCost, time.time() - t0))
        lib.bdpt_test_tree_builder(-1)
        # the whole device pipeline (tree + quantise + pack) against the host's packed records
        rh = []
        for dev in ((0, 0) if only else (-1, 0)):
            h = C.c_uint64()
            info = pkg.abi.BvhInfo()
            t0 = time.time()
            rc = lib.bdpt_bvh_recs_hash(C.byref(sc.desc), dev, C.byref(h), C.byref(info))
            rh.append((rc, h.value, info.numNodes, info.maxDepth, info.maxStack, info.sahCost, info.reserved, time.time() - t0))
        same = res[0][:6] == res[1][:6] and rh[0][:7] == rh[1][:7] and rh[0][0] == 0
        bad += 0 if same else 1
        print(f"{name:16s} host {res[0][1]:#018x} {res[0][2]:9d} nodes {res[0][6]:7.3f} s | device {res[1][1]:#018x} {res[1][2]:9d} nodes {res[1][6]:7.3f} s | records {rh[0][1]:#018x} {rh[0][7]:6.3f} s / {rh[1][1]:#018x} {rh[1][7]:6.3f} s | {'SAME' if same else 'DIFFERENT'}", flush=True)
        sc.close()
        for k in env:
            del os.environ[k]
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
