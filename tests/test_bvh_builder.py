"""Host-side tests of the acceleration-structure builder (csrc/bvh_build.cpp, alpha_clip.cpp, scene_bvh.cpp) through
the CPU trace hook bdpt_host_bvh_*: spatial pre-splitting, per-triangle alpha classification and the clipping /
dropping of pieces the alpha test can never pass must leave every query's answer exactly what the linear scan over
all triangles (with the reference's per-material opacity, Falcor Raytracing/RtModel.cpp:221-224) gives.
No GPU, no oracle: both sides are the product's own host code."""
import ctypes as C
import os

import numpy as np
import pytest


def _rays(rng, n, lo, hi, tmax=None):
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.zeros((n, 8), np.float32)
    r[:, 0:3] = o
    r[:, 3:6] = d
    r[:, 6] = 1e-4
    r[:, 7] = 1e38 if tmax is None else rng.uniform(0.2, tmax, n).astype(np.float32)
    return r


def _trace(lib, h, rays, mode, brute):
    n = rays.shape[0]
    prim = np.zeros(n, np.int32)
    tuv = np.zeros((n, 3), np.float32)
    vis = (C.c_uint64 * 2)()
    assert lib.bdpt_host_bvh_trace(h, rays.ctypes.data, n, mode, brute, 0, prim.ctypes.data, tuv.ctypes.data, vis) == 0
    return prim, tuv, (int(vis[0]), int(vis[1]))


def _bounds(desc):
    p = np.ctypeslib.as_array(desc.positions, shape=(desc.numVertices, 3))
    return p.min(axis=0), p.max(axis=0)


def _check_against_scan(pkg, scene, lo, hi, variants, n_rays=1500, seed=5):
    lib = pkg.load_library()
    rng = np.random.default_rng(seed)
    sets = {0: _rays(rng, n_rays, lo, hi), 1: _rays(rng, n_rays, lo, hi), 2: _rays(rng, n_rays, lo, hi, tmax=float(np.max(hi - lo)))}
    truth = None
    stats = []
    for (b, ba, cl) in variants:
        info = pkg.abi.BvhInfo()
        h = lib.bdpt_host_bvh_create(C.byref(scene.desc), 2, b, ba, cl, C.byref(info))
        assert h
        if truth is None:
            truth = {m: _trace(lib, h, r, m, 1)[:2] for m, r in sets.items()}
            assert (truth[0][0] >= 0).sum() > n_rays // 10, "the sample must actually hit the scene"
        for m, r in sets.items():
            prim, tuv, vis = _trace(lib, h, r, m, 0)
            if m == 2:
                assert ((prim >= 0) == (truth[m][0] >= 0)).all(), (b, ba, cl, m)
            else:
                assert (prim == truth[m][0]).all() and (tuv.view(np.uint32) == truth[m][1].view(np.uint32)).all(), (b, ba, cl, m)
        stats.append((info.numReferences, info.numDropped, info.numAlwaysPass, info.numAlphaMode))
        lib.bdpt_host_bvh_destroy(h)
    return stats


def test_split_and_clipped_tree_answers_like_the_linear_scan_foliage(pkg):
    """Alpha-masked leaf cards (the courtyard stand-in of BASELINE configs[4]), dense enough to overlap."""
    scene = pkg.Scene.courtyard(2, 12000, 0.6)
    lo, hi = _bounds(scene.desc)
    stats = _check_against_scan(pkg, scene, lo, hi, [(0.0, 0.0, 0), (0.0, 0.0, 1), (-1.0, -1.0, 1), (0.5, 6.0, 1), (2.0, 16.0, 1), (1.0, 3.0, 0)])
    refs = [s[0] for s in stats]
    assert refs[0] == refs[1] == scene.desc.numTriangles
    assert refs[2] > refs[0] and refs[3] > refs[2] and refs[4] > refs[3]
    assert stats[0][3] > 3000  # alpha-mode triangles present
    scene.close()


def test_split_and_clipped_tree_answers_like_the_linear_scan_atrium_and_soup(pkg):
    scene = pkg.Scene.atrium(1, 6000)
    lo, hi = _bounds(scene.desc)
    _check_against_scan(pkg, scene, lo, hi, [(0.0, 0.0, 0), (-1.0, -1.0, 1), (1.0, 4.0, 1)])
    scene.close()
    scene = pkg.Scene.soup(11, 4000, 0.4)  # long overlapping triangles: the opaque budget has something to cut
    lo, hi = _bounds(scene.desc)
    stats = _check_against_scan(pkg, scene, lo, hi, [(0.0, 0.0, 0), (1.0, 0.0, 1), (4.0, 0.0, 1)])
    scene.close()
    assert stats[1][0] >= stats[0][0]


def _card_scene(pkg, alpha_rows, threshold=0.5, uv_scale=1.0):
    """One unit card (two triangles) in the plane z = 0 with an alpha texture made of the given rows, plus a large
    opaque floor behind it; returns (desc, keep-alive list)."""
    a = pkg.abi
    tex = np.zeros((len(alpha_rows), len(alpha_rows[0]), 4), np.uint8)
    tex[..., :3] = 200
    tex[..., 3] = np.array(alpha_rows, np.uint8)
    pos = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [-4, -4, -1], [5, -4, -1], [5, 5, -1], [-4, 5, -1]], np.float32)
    nrm = np.tile(np.array([[0, 0, 1]], np.float32), (8, 1))
    uv = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32)
    uv[:, :2] *= uv_scale
    idx = np.array([0, 1, 2, 0, 2, 3, 4, 5, 6, 4, 6, 7], np.uint32)
    tri_mat = np.array([0, 0, 1, 1], np.uint32)
    mats = (a.Material * 2)()
    for m, alpha_mode, tex_id in ((mats[0], 1, 0), (mats[1], 0, -1)):
        for k in range(4):
            m.baseColor[k] = 1.0
        m.alphaThreshold = threshold
        m.IoR = 1.5
        m.flags = (0 | ((2 if tex_id >= 0 else 1) << 3) | (1 << 6) | (alpha_mode << 17) | (1 << 19))
        m.texBaseColor, m.texSpecular, m.texEmissive, m.texNormal = tex_id, -1, -1, -1
    texs = (a.Texture * 1)()
    texs[0].rgba8 = tex.ctypes.data_as(C.POINTER(C.c_uint8))
    texs[0].width, texs[0].height, texs[0].srgb = tex.shape[1], tex.shape[0], 1
    lights = (a.Light * 1)()
    lights[0].posW[2] = 3.0
    lights[0].intensity[0] = lights[0].intensity[1] = lights[0].intensity[2] = 1.0
    d = a.SceneDesc()
    d.numVertices, d.numTriangles, d.numMaterials, d.numTextures, d.numLights = 8, 4, 2, 1, 1
    fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
    d.positions, d.normals, d.texcoords = fp(pos), fp(nrm), fp(uv)
    d.indices = idx.ctypes.data_as(C.POINTER(C.c_uint32))
    d.triMaterial = tri_mat.ctypes.data_as(C.POINTER(C.c_uint32))
    d.materials, d.textures, d.lights = mats, texs, lights
    return d, [tex, pos, nrm, uv, idx, tri_mat, mats, texs, lights]


def _grid_rays(n=64, z0=1.0):
    """Rays straight down onto the card, a regular grid over [-0.1, 1.1]^2 (also just outside the card)."""
    xs = np.linspace(-0.1, 1.1, n, dtype=np.float32)
    X, Y = np.meshgrid(xs, xs)
    r = np.zeros((n * n, 8), np.float32)
    r[:, 0], r[:, 1], r[:, 2] = X.ravel(), Y.ravel(), z0
    r[:, 5] = -1.0
    r[:, 6], r[:, 7] = 1e-4, 1e38
    return r


@pytest.mark.parametrize("uv_scale", [1.0, 3.0])
def test_alpha_classification_all_pass_all_fail_and_mixed(pkg, uv_scale):
    lib = pkg.load_library()
    rays = _grid_rays()
    cases = {
        "all_pass": ([[255] * 8] * 8, (2, 0)),   # both card triangles lose their non-opaque flag
        "all_fail": ([[0] * 8] * 8, (0, 2)),     # both are dropped from the tree
        "left_half": ([[255, 255, 255, 255, 0, 0, 0, 0]] * 8, (0, 0)),
        "one_texel": ([[0] * 8] * 3 + [[0, 0, 0, 0, 0, 255, 0, 0]] + [[0] * 8] * 4, (0, 0)),
    }
    for name, (rows, (n_pass, n_drop)) in cases.items():
        if uv_scale != 1.0 and name == "one_texel":
            continue
        d, keep = _card_scene(pkg, rows, uv_scale=uv_scale)
        info0, info1 = pkg.abi.BvhInfo(), pkg.abi.BvhInfo()
        h0 = lib.bdpt_host_bvh_create(C.byref(d), 1, 0.0, 0.0, 0, C.byref(info0))
        h1 = lib.bdpt_host_bvh_create(C.byref(d), 1, 0.0, 8.0, 1, C.byref(info1))
        assert h0 and h1
        assert info1.numAlphaMode == 2 and info1.numAlwaysPass == n_pass and info1.numDropped == n_drop, name
        for mode in (0, 1, 2):
            truth = _trace(lib, h0, rays, mode, 1)
            for h in (h0, h1):
                got = _trace(lib, h, rays, mode, 0)
                if mode == 2:
                    assert ((got[0] >= 0) == (truth[0] >= 0)).all(), (name, mode)
                else:
                    assert (got[0] == truth[0]).all() and (got[1].view(np.uint32) == truth[1].view(np.uint32)).all(), (name, mode)
        if name == "left_half" and uv_scale == 1.0:
            prim = _trace(lib, h1, rays, 0, 0)[0]
            on_card = (rays[:, 0] > 0.02) & (rays[:, 0] < 0.98) & (rays[:, 1] > 0.02) & (rays[:, 1] < 0.98)
            assert (prim[on_card & (rays[:, 0] < 0.40)] < 2).all() and (prim[on_card & (rays[:, 0] > 0.60)] >= 2).all()
            # the clipped references bound the passing half only: rays over the cut-away half never test the card
            right = rays[on_card & (rays[:, 0] > 0.70)]
            v1 = _trace(lib, h1, right, 0, 0)[2]
            v0 = _trace(lib, h0, right, 0, 0)[2]
            assert v1[1] < v0[1], (v0, v1)
        lib.bdpt_host_bvh_destroy(h0)
        lib.bdpt_host_bvh_destroy(h1)


def test_pre_splitting_cuts_visits_on_overlapping_cards(pkg):
    """What the alpha budget is for: on dense overlapping leaf cards the split + clipped tree needs clearly fewer
    triangle tests per ray than the one-reference-per-triangle tree (tools/bvh_eval.py gives the full-size figures)."""
    lib = pkg.load_library()
    scene = pkg.Scene.courtyard(2, 60000, 0.8)
    lo, hi = _bounds(scene.desc)
    rng = np.random.default_rng(3)
    # rays through the crowns: start and end inside the trees' band above the floor
    rays = _rays(rng, 4000, np.array([-13.0, 2.5, -2.5], np.float32), np.array([13.0, 6.5, 2.5], np.float32))
    res = []
    for b, ba, cl in ((0.0, 0.0, 0), (-1.0, -1.0, 1)):
        info = pkg.abi.BvhInfo()
        h = lib.bdpt_host_bvh_create(C.byref(scene.desc), 2, b, ba, cl, C.byref(info))
        res.append(_trace(lib, h, rays, 0, 0))
        lib.bdpt_host_bvh_destroy(h)
    assert (res[0][0] == res[1][0]).all() and (res[0][1].view(np.uint32) == res[1][1].view(np.uint32)).all()
    assert res[1][2][1] < 0.75 * res[0][2][1], (res[0][2], res[1][2])
    scene.close()


def test_builder_edge_cases_all_dropped_no_texcoords_huge_tiling_degenerate(pkg):
    """Corners of the reference-building code: a scene whose every triangle fails the alpha test everywhere (an empty
    tree), an alpha-mode material on a mesh without texture coordinates (nothing can be classified: every hit is
    tested, at uv = 0), texture coordinates tiled hundreds of times over one triangle (the footprint covers the whole
    texture), and zero-area triangles among the split ones — the tree must still answer like the linear scan."""
    lib = pkg.load_library()
    rays = _grid_rays(24)
    # 1. only the card, all texels transparent
    d, keep = _card_scene(pkg, [[0] * 4] * 4)
    d.numTriangles = 2
    info = pkg.abi.BvhInfo()
    h = lib.bdpt_host_bvh_create(C.byref(d), 1, -1.0, -1.0, 1, C.byref(info))
    assert h and info.numDropped == 2 and info.numReferences == 0
    for mode in (0, 1, 2):
        prim, _, vis = _trace(lib, h, rays, mode, 0)
        assert (prim < 0).all() and vis[1] == 0  # nothing to test
        assert (_trace(lib, h, rays, mode, 1)[0] < 0).all()
    lib.bdpt_host_bvh_destroy(h)
    # 2. no texture coordinates at all
    d, keep = _card_scene(pkg, [[255, 0, 255, 0]] * 4)
    d.texcoords = None
    h0 = lib.bdpt_host_bvh_create(C.byref(d), 1, 0.0, 0.0, 0, C.byref(info))
    h1 = lib.bdpt_host_bvh_create(C.byref(d), 1, 1.0, 6.0, 1, C.byref(info))
    assert info.numAlwaysPass == 0 and info.numDropped == 0
    for mode in (0, 2):
        a, b = _trace(lib, h0, rays, mode, 1), _trace(lib, h1, rays, mode, 0)
        assert ((a[0] >= 0) == (b[0] >= 0)).all() and (mode == 2 or (a[0] == b[0]).all())
    lib.bdpt_host_bvh_destroy(h0)
    lib.bdpt_host_bvh_destroy(h1)
    # 3. the texture repeated 300 times across the card: mixed texels -> nothing classified, everything answers the same
    d, keep = _card_scene(pkg, [[255, 0, 0, 255]] * 4, uv_scale=300.0)
    h0 = lib.bdpt_host_bvh_create(C.byref(d), 1, 0.0, 0.0, 0, C.byref(info))
    h1 = lib.bdpt_host_bvh_create(C.byref(d), 1, 0.0, 16.0, 1, C.byref(info))
    assert info.numAlwaysPass == 0 and info.numDropped == 0
    fine = _grid_rays(96)
    for mode in (0, 2):
        a, b = _trace(lib, h0, fine, mode, 1), _trace(lib, h1, fine, mode, 0)
        assert ((a[0] >= 0) == (b[0] >= 0)).all() and (mode == 2 or ((a[0] == b[0]).all() and (a[1].view(np.uint32) == b[1].view(np.uint32)).all()))
    first = _trace(lib, h1, fine, 0, 0)[0]
    on_card = (fine[:, 0] > 0.02) & (fine[:, 0] < 0.98) & (fine[:, 1] > 0.02) & (fine[:, 1] < 0.98)
    assert 0.2 < (first[on_card] < 2).mean() < 0.8  # about half of the card lets rays through to the floor
    lib.bdpt_host_bvh_destroy(h0)
    lib.bdpt_host_bvh_destroy(h1)
    # 4. zero-area triangles: collapse the card's second triangle onto an edge; it can never be hit, the rest still is
    d, keep = _card_scene(pkg, [[255] * 4, [0] * 4, [255] * 4, [0] * 4])
    pos = keep[1]
    pos[3] = pos[2]  # vertex 3 := vertex 2: triangle (0, 2, 3) degenerates
    h0 = lib.bdpt_host_bvh_create(C.byref(d), 1, 0.0, 0.0, 0, C.byref(info))
    h1 = lib.bdpt_host_bvh_create(C.byref(d), 1, 2.0, 12.0, 1, C.byref(info))
    for mode in (0, 1, 2):
        a, b = _trace(lib, h0, rays, mode, 1), _trace(lib, h1, rays, mode, 0)
        assert ((a[0] >= 0) == (b[0] >= 0)).all() and (mode == 2 or (a[0] == b[0]).all())
    lib.bdpt_host_bvh_destroy(h0)
    lib.bdpt_host_bvh_destroy(h1)


def test_an_exception_on_a_builder_worker_thread_comes_back_as_an_error_code(pkg):
    """The builder's worker threads make the large allocations of a big scene; std::bad_alloc thrown on one of them
    must not reach std::terminate.  BDPT_TEST_THROW_IN_WORKER makes every worker throw (bvh.h WorkerScope): the host
    hooks return BDPT_E_NOMEM instead of aborting the process, and a normal build works again afterwards."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, os, sys\n"
        "sys.path.insert(0, %r)\n"
        "import __graft_entry__ as ge\n"
        "pkg = ge.load_package(); lib = pkg.load_library()\n"
        "sc = pkg.Scene.atrium(1, 80000)\n"
        "h = C.c_uint64(); info = pkg.abi.BvhInfo()\n"
        "os.environ['BDPT_TEST_THROW_IN_WORKER'] = '1'\n"
        "a = lib.bdpt_bvh_build_hash(C.byref(sc.desc), 4, C.byref(h), C.byref(info))\n"
        "msg = C.create_string_buffer(64)\n"
        "b = lib.bdpt_bvh_build_check(C.byref(sc.desc), C.byref(info), msg, 64)\n"
        "del os.environ['BDPT_TEST_THROW_IN_WORKER']\n"
        "c = lib.bdpt_bvh_build_hash(C.byref(sc.desc), 4, C.byref(h), C.byref(info))\n"
        "print('codes', a, b, c, info.numTriangles)\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]  # (an uncaught exception on a std::thread would have aborted: rc -6)
    assert "codes -4 -4 0 80000" in r.stdout, r.stdout


def test_recs_hash_hook_on_the_host_pipeline_and_device_hooks_without_a_gpu(pkg):
    """bdpt_bvh_recs_hash (device < 0: everything by the host code) is what the GPU suite compares the device build with:
    it has to be deterministic, to see a change of the scene, and to report the record count; without a GPU the
    device-side settings of the test hooks fail with BDPT_E_HIP instead of falling back to the host code."""
    import torch
    lib = pkg.load_library()
    a, b = pkg.Scene.courtyard(1, 20000), pkg.Scene.courtyard(2, 20000)
    got = []
    for sc in (a, a, b):
        h = C.c_uint64()
        info = pkg.abi.BvhInfo()
        assert lib.bdpt_bvh_recs_hash(C.byref(sc.desc), -1, C.byref(h), C.byref(info)) == 0
        got.append((h.value, info.numNodes, info.numReferences, info.reserved))
    assert got[0] == got[1] and got[0][0] != got[2][0]
    # records = one per node + one per reference + the pad records
    assert got[0][3] == got[0][1] + got[0][2] + 4
    if not torch.cuda.is_available():
        h = C.c_uint64()
        assert lib.bdpt_bvh_recs_hash(C.byref(a.desc), 0, C.byref(h), None) == -3  # BDPT_E_HIP
        assert lib.bdpt_test_tree_builder(0) == -3
    assert lib.bdpt_test_tree_builder(-1) == 0
    a.close()
    b.close()
