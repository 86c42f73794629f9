"""BMFR denoise pass (DenoisePass.cpp:146-279 + preprocess / regressionCP / postprocess shaders).

The reference ships no images or vectors for this pass (it is off by default, DenoisePass.h:71), so the oracle
(oracle/bmfr_oracle.cpp) is a restatement pinned only by the properties checked below — PARITY UNPINNED — and the
GPU kernels are compared with it bit for bit.
"""
import ctypes as C
import math

import numpy as np
import pytest


def _params(pkg, frame, flags, vp=None):
    p = pkg.abi.BmfrParams()
    p.frameNumber, p.flags = frame, flags
    vp = vp if vp is not None else [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]
    for i in range(16):
        p.prevViewProj[i] = vp[i]
    return p


def _plane_scene(W, H, rng, noise=0.3):
    """A tilted textured plane seen head-on: positions vary linearly over the image, normal constant."""
    y, x = np.mgrid[0:H, 0:W].astype(np.float32)
    pos = np.zeros((H, W, 4), np.float32)
    pos[..., 0] = (x + 0.5 - W / 2) * 0.02  # pixel centres (PIXEL_OFFSET, preprocess.ps.hlsl:22)
    pos[..., 1] = (H / 2 - y - 0.5) * 0.02
    pos[..., 2] = -3.0 + 0.1 * pos[..., 0]
    pos[..., 3] = 1.0
    nrm = np.zeros((H, W, 4), np.float32)
    nrm[..., 2] = 1.0
    alb = np.ones((H, W, 4), np.float32)
    alb[..., 0] = 0.25 + 0.5 * ((x // 8 + y // 8) % 2)
    alb[..., 1] = 0.5
    alb[..., 2] = 0.75
    alb = alb.astype(np.float16).astype(np.float32)
    # irradiance: smooth in the features (linear + quadratic in position), radiance = albedo * irradiance + noise
    irr = 0.6 + 0.2 * pos[..., 0:1] - 0.1 * pos[..., 1:2] + 0.05 * pos[..., 0:1] ** 2
    clean = alb[..., :3] * irr
    noisy = np.ones((H, W, 4), np.float32)
    noisy[..., :3] = clean * (1.0 + noise * rng.standard_normal((H, W, 3)).astype(np.float32))
    return pos.reshape(-1, 4), nrm.reshape(-1, 4), alb.reshape(-1, 4), noisy.reshape(-1, 4), clean


def test_view_proj_matches_numpy(pkg):
    pos, tgt, up = (1.0, 2.0, 5.0), (0.2, 1.0, -1.0), (0.0, 1.0, 0.0)
    m = np.array(pkg.camera_view_proj(pos, tgt, up, 21.0, 24.0, 1.5, 0.1, 1000.0)).reshape(4, 4)
    f = np.array(tgt) - np.array(pos)
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    # a point straight ahead projects to the image centre, a point along +s to +x, along +u to +y, w = distance
    c = m @ np.append(np.array(pos) + 4 * f, 1.0)
    assert abs(c[0]) < 1e-5 and abs(c[1]) < 1e-5 and c[3] == pytest.approx(4.0, rel=1e-5)
    c = m @ np.append(np.array(pos) + 4 * f + s, 1.0)
    th = 12.0 / 21.0
    assert c[0] / c[3] == pytest.approx(1.0 / (4 * 1.5 * th), rel=1e-4) and abs(c[1]) < 1e-5
    c = m @ np.append(np.array(pos) + 4 * f + u, 1.0)
    assert c[1] / c[3] == pytest.approx(1.0 / (4 * th), rel=1e-4)


def test_oracle_regression_recovers_smooth_irradiance(pkg, ob):
    """Fitting albedo-demodulated colour to the feature basis removes noise where the signal lies in the basis."""
    rng = np.random.default_rng(20260104)
    W, H = 96, 64
    pos, nrm, alb, noisy, clean = _plane_scene(W, H, rng)
    before = noisy.copy()
    b = ob.OracleBmfr(pkg.abi, W, H)
    flags = pkg.abi.BMFR_REGRESSION | pkg.abi.BMFR_FULL_FRAME
    b.execute(_params(pkg, 0, flags), pos, nrm, alb, noisy)
    err0 = np.sqrt(np.mean((before.reshape(H, W, 4)[..., :3] - clean) ** 2))
    err1 = np.sqrt(np.mean((noisy.reshape(H, W, 4)[..., :3] - clean) ** 2))
    assert np.isfinite(noisy).all() and err1 < 0.25 * err0, (err0, err1)
    # the reference's half-image mode leaves the right half untouched
    noisy2 = before.copy()
    b.reset()
    b.execute(_params(pkg, 0, pkg.abi.BMFR_REGRESSION), pos, nrm, alb, noisy2)
    a2, b2 = noisy2.reshape(H, W, 4), before.reshape(H, W, 4)
    # blocks start at negative offsets, so the fitted region ends a little before the middle (DenoisePass.cpp:255-262)
    assert np.array_equal(a2[:, W // 2 + 2:], b2[:, W // 2 + 2:]) and not np.array_equal(a2[:, :W // 4], b2[:, :W // 4])
    # the plain Householder variant (no rank check, feature noise) also denoises
    noisy3 = before.copy()
    b.reset()
    b.execute(_params(pkg, 0, flags | pkg.abi.BMFR_KEEP_LD_FEATURES), pos, nrm, alb, noisy3)
    err3 = np.sqrt(np.mean((noisy3.reshape(H, W, 4)[..., :3] - clean) ** 2))
    assert np.isfinite(noisy3).all() and err3 < 0.5 * err0
    b.close()


def test_oracle_temporal_accumulation_static_camera(pkg, ob):
    """Static view, identity reprojection: pre/post-process average the frames (spp counts up, blend capped)."""
    rng = np.random.default_rng(7)
    W, H = 64, 48
    pos, nrm, alb, _, clean = _plane_scene(W, H, rng)
    # a view-projection that maps every world position back onto its own pixel centre
    vp = np.zeros((4, 4), np.float32)
    vp[0, 0], vp[0, 3] = 1.0 / (0.02 * W / 2), 0.0
    vp[1, 1] = 1.0 / (0.02 * H / 2)
    vp[3, 3] = 1.0
    b = ob.OracleBmfr(pkg.abi, W, H)
    flags = pkg.abi.BMFR_PREPROCESS | pkg.abi.BMFR_POSTPROCESS | pkg.abi.BMFR_FULL_FRAME
    errs = []
    for frame in range(12):
        noisy = np.ones((H * W, 4), np.float32)
        noisy[:, :3] = (clean * (1.0 + 0.3 * rng.standard_normal((H, W, 3)))).reshape(-1, 3).astype(np.float32)
        b.execute(_params(pkg, frame, flags, list(vp.reshape(-1))), pos, nrm, alb, noisy)
        errs.append(float(np.sqrt(np.mean((noisy.reshape(H, W, 4)[..., :3] - clean) ** 2))))
        assert np.isfinite(noisy).all() and (noisy[:, 3] == 1.0).all()
    assert errs[-1] < 0.45 * errs[0], errs
    b.close()


def _pose(k):
    a = 0.05 * k
    return (278.0 + 40.0 * math.sin(a), 273.0 + 5.0 * k, -800.0 + 10.0 * k), (278.0, 273.0, 0.0), (0.0, 1.0, 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("flags_name", ["default", "regression", "regression_keep_ld_full"])
def test_bmfr_sequence_matches_oracle(pkg, ob, flags_name):
    """Five frames of G-buffer + BDPT + BMFR with a moving camera: the denoised channel equals the oracle's bit for bit."""
    import torch
    A = pkg.abi
    flags = {"default": A.BMFR_PREPROCESS | A.BMFR_POSTPROCESS,
             "regression": A.BMFR_PREPROCESS | A.BMFR_REGRESSION | A.BMFR_POSTPROCESS,
             "regression_keep_ld_full": A.BMFR_PREPROCESS | A.BMFR_REGRESSION | A.BMFR_POSTPROCESS | A.BMFR_KEEP_LD_FEATURES
             | A.BMFR_FULL_FRAME}[flags_name]
    scene = pkg.Scene.cornell()
    W, H = 100, 70
    pipe = pkg.FramePipeline(scene, W, H, max_depth=3, mat_index=1)
    orc = ob.OracleRender(A, scene.desc, W, H)
    den = ob.OracleBmfr(A, W, H)
    lib = pkg.load_library()
    v3 = lambda v: (C.c_float * 3)(*v)
    prev_vp = None
    for k in range(5):
        pos, tgt, up = _pose(k)
        cam = A.Camera()
        assert lib.bdpt_camera_look_at(v3(pos), v3(tgt), v3(up), 33.6, 24.0, W / H, 1.0, C.byref(cam)) == 0
        pipe.cam = cam
        pipe.ctx.set_camera(cam)
        gp, p = pipe.render_frame()
        bp = _params(pkg, k, flags, prev_vp)
        pipe.ctx.bmfr_execute(bp, pipe.gb, C.c_void_p(pipe.output.data_ptr()), pipe._stream_ptr())
        torch.cuda.synchronize()
        orc.gbuffer(cam, gp)
        orc.bdpt(cam, p)
        orc.resolve()
        ref = orc.image().reshape(-1, 4).copy()
        den.execute(bp, orc.chan["worldPosition"], orc.chan["worldNormal"], orc.chan["materialDiffuse"], ref)
        gpu = pipe.output.cpu().numpy().reshape(-1, 4)
        same = (gpu.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(gpu) & np.isnan(ref))
        assert same.all(), f"frame {k}: {(~same).any(axis=1).sum()} pixels differ, max |d| {np.nanmax(np.abs(gpu - ref))}"
        prev_vp = pkg.camera_view_proj(pos, tgt, up, 33.6, 24.0, W / H)
    # history is dropped by reset: frame 0 after a reset equals a fresh context's frame 0
    pipe.ctx.bmfr_reset()
    den.reset()
    gp, p = pipe.render_frame()
    bp = _params(pkg, 0, flags, prev_vp)
    pipe.ctx.bmfr_execute(bp, pipe.gb, C.c_void_p(pipe.output.data_ptr()), pipe._stream_ptr())
    torch.cuda.synchronize()
    orc.gbuffer(pipe.cam, gp)
    orc.bdpt(pipe.cam, p)
    orc.resolve()
    ref = orc.image().reshape(-1, 4).copy()
    den.execute(bp, orc.chan["worldPosition"], orc.chan["worldNormal"], orc.chan["materialDiffuse"], ref)
    gpu = pipe.output.cpu().numpy().reshape(-1, 4)
    assert np.array_equal(gpu.view(np.uint32), ref.view(np.uint32))
    den.close()
    orc.close()
    pipe.close()
    scene.close()


@pytest.mark.gpu
def test_bmfr_on_a_band_context_filters_whole_frame_buffers(pkg):
    """A context that renders a band (or one rank's stripes) still denoises: bdpt_bmfr_execute takes WHOLE-FRAME buffers
    whatever the context's tile (a tiled host gathers them first), and gives what a whole-frame context gives, bit for bit;
    the history crosses a save / load (bdpt_bmfr_save_history / _load_history), and a blob of another frame size is refused."""
    import torch
    scene = pkg.Scene.cornell()
    W = H = 64
    full = pkg.FramePipeline(scene, W, H, max_depth=2, mat_index=1)
    band = pkg.FramePipeline(scene, W, H, max_depth=2, mat_index=1, tile=(0, 16))
    lib = pkg.load_library()
    outs = {}
    for name, pipe in (("full", full), ("band", band)):
        res = []
        for k in range(3):
            full.gbuffer_frame, full.bdpt_frame = 0xdead0000 + k, 0x1337 + k
            full.render_frame()  # the whole-frame channels both contexts filter
            torch.cuda.synchronize()
            noisy = full.output.clone()
            pipe.ctx.bmfr_execute(_params(pkg, k, 7), full.gb, C.c_void_p(noisy.data_ptr()), pipe._stream_ptr())
            torch.cuda.synchronize()
            res.append(noisy.cpu().numpy().copy())
            if name == "band" and k == 1:  # the history survives a save, a reset and a load
                n = C.c_uint64()
                assert lib.bdpt_bmfr_history_bytes(pipe.ctx._h, C.byref(n)) == 0 and n.value == W * H * 16 * 4
                blob = (C.c_uint8 * n.value)()
                assert lib.bdpt_bmfr_save_history(pipe.ctx._h, blob, n.value) == 0
                assert lib.bdpt_bmfr_reset(pipe.ctx._h) == 0
                assert lib.bdpt_bmfr_load_history(pipe.ctx._h, blob, n.value - 16) != 0  # another frame size
                assert lib.bdpt_bmfr_load_history(pipe.ctx._h, blob, n.value) == 0
        outs[name] = res
    for a, b in zip(outs["full"], outs["band"]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert not np.array_equal(outs["full"][0], outs["full"][2])
    full.close()
    band.close()
    scene.close()


@pytest.mark.gpu
def test_cpp_host_denoise_pass_matches_python_sequence(pkg, tmp_path):
    """host/bdpt_render with the BMFR pass ticked (fourth pass of Main.cpp:15-18) == the same frames driven from Python."""
    import os
    import subprocess
    import torch
    import __graft_entry__ as ge
    A = pkg.abi
    exe = os.path.join(ge.PKG_DIR, "host", "bdpt_render")
    assert os.path.exists(exe), "host/bdpt_render not built (run __graft_entry__.build())"
    raw = tmp_path / "out.f32"
    W, H, frames = 80, 48, 4
    r = subprocess.run([exe, "--scene", "cornell", "--width", str(W), "--height", str(H), "--frames", str(frames), "--depth", "3",
                        "--mat", "1", "--denoise-regression", "--out", str(tmp_path / "o.pfm"), "--raw", str(raw)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    cpp = np.fromfile(raw, np.float32).reshape(H, W, 4)
    scene = pkg.Scene.cornell()
    pipe = pkg.FramePipeline(scene, W, H, max_depth=3, mat_index=1, accum_limit=100)
    # Camera::beginFrame: the (static) camera's jitter-free view-projection, near/far = Falcor defaults
    vp = pkg.camera_view_proj((278.0, 273.0, -800.0), (278.0, 273.0, 0.0), (0.0, 1.0, 0.0), 33.6, 24.0, W / H, 0.1, 1000.0)
    flags = A.BMFR_PREPROCESS | A.BMFR_REGRESSION | A.BMFR_POSTPROCESS
    for k in range(frames):
        pipe.render_frame(accumulate=True)
        pipe.ctx.bmfr_execute(_params(pkg, k, flags, vp), pipe.gb, C.c_void_p(pipe.output.data_ptr()), pipe._stream_ptr())
    torch.cuda.synchronize()
    py = pipe.output.cpu().numpy()
    assert np.array_equal(cpp.view(np.uint32), py.view(np.uint32)), np.abs(cpp - py).max()
    pipe.close()
    scene.close()


@pytest.mark.gpu
def test_bmfr_frame_smaller_than_the_block_offsets(pkg, ob):
    """A 40x20 frame: block offsets reach -32, one reflection does not bring every load back inside, and the shader
    would read outside its textures (0 in D3D).  Found by the CPU sanitizer run; both sides now define those loads as 0."""
    import torch
    A = pkg.abi
    scene = pkg.Scene.cornell()
    W, H = 40, 20
    pipe = pkg.FramePipeline(scene, W, H, max_depth=3, mat_index=1)
    orc = ob.OracleRender(A, scene.desc, W, H)
    den = ob.OracleBmfr(A, W, H)
    flags = A.BMFR_PREPROCESS | A.BMFR_REGRESSION | A.BMFR_POSTPROCESS | A.BMFR_FULL_FRAME
    vp = pkg.camera_view_proj((278.0, 273.0, -800.0), (278.0, 273.0, 0.0), (0.0, 1.0, 0.0), 33.6, 24.0, W / H)
    for k in range(3):
        gp, p = pipe.render_frame()
        bp = _params(pkg, k, flags, vp)
        pipe.ctx.bmfr_execute(bp, pipe.gb, C.c_void_p(pipe.output.data_ptr()), pipe._stream_ptr())
        torch.cuda.synchronize()
        orc.gbuffer(pipe.cam, gp)
        orc.bdpt(pipe.cam, p)
        orc.resolve()
        ref = orc.image().reshape(-1, 4).copy()
        den.execute(bp, orc.chan["worldPosition"], orc.chan["worldNormal"], orc.chan["materialDiffuse"], ref)
        gpu = pipe.output.cpu().numpy().reshape(-1, 4)
        same = (gpu.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(gpu) & np.isnan(ref))
        assert same.all(), f"frame {k}: {(~same).any(axis=1).sum()} pixels differ"
    den.close()
    orc.close()
    pipe.close()
    scene.close()
