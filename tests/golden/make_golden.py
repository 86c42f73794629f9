#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the CPU oracle (oracle/liboracle.so).

The reference repository holds no golden vectors, known-answer tests or rendered images for this
path and cannot be built or run here (HLSL/DXR + D3D12), so these fixtures pin the oracle to
ITSELF across machines, compilers and future edits — they are regression pins of the restatement,
not outputs of the reference ("parity unpinned" for the parts the reference leaves to the DXR
driver; see oracle/bdpt_oracle.h).  Run from the repo root:  python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as ge  # noqa: E402
import oracle_binding as ob  # noqa: E402

SEED = 20260104


def kat_inputs(pkg):
    rng = np.random.default_rng(SEED)
    n = 512
    v0 = rng.integers(0, 2**32, n, dtype=np.uint32)
    v1 = rng.integers(0, 2**32, n, dtype=np.uint32)
    v0[:6] = [0, 1, 255, 256 * 256 - 1, 1920 * 1080 - 1, 0xFFFFFFFF]
    v1[:6] = [0x1337, 0x1337, 0x1338, 0x1337, 0x1337 + 1023, 0xFFFFFFFF]

    def unit(k):
        v = rng.normal(size=(k, 3))
        return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)

    m = 2048
    N, V, L = unit(m), unit(m), unit(m)
    V = np.where((np.sum(N * V, axis=1, keepdims=True) < 0), -V, V).astype(np.float32)
    rec = np.zeros((m, 20), np.float32)
    rec[:, 0:3], rec[:, 3:6], rec[:, 6:9] = N, V, L
    rec[:, 9:12] = rng.uniform(0, 1, (m, 3))
    rec[:, 12:15] = rng.uniform(0, 1, (m, 3))
    rec[:, 15] = rng.uniform(0.08, 1, m) ** 2
    rec[:, 16] = rng.integers(0, 2, m)
    rec[:, 17] = rng.integers(0, 2**32, m, dtype=np.uint32).view(np.float32)
    rec[:8, 9:15] = 0.0
    o = rng.uniform(-0.5, 1.5, (m, 3)).astype(np.float32)
    d = unit(m)
    d[: m // 8] *= rng.uniform(0.1, 5.0, (m // 8, 1)).astype(np.float32)
    tmax = np.full((m, 1), 1e38, np.float32)
    tmax[m // 2:] = rng.uniform(0.05, 2.0, (m - m // 2, 1)).astype(np.float32)
    rays = np.concatenate([o, d, np.full((m, 1), 1e-4, np.float32), tmax], axis=1).astype(np.float32)
    u = rng.uniform(0, 1, 4096).astype(np.float32)
    u[:4] = [0.0, 0.25, 0.5, np.float32(1.0) - np.float32(2.0 ** -24)]
    h = np.concatenate([rng.normal(size=2000).astype(np.float32) * 10.0,
                        (rng.normal(size=2000) * 1e-6).astype(np.float32),
                        np.array([0.0, -0.0, 65504.0, 65520.0, 1e6, -1e6, 5.96e-8, 2.98e-8, 6.1e-5, np.inf, -np.inf],
                                 np.float32)])
    return dict(v0=v0, v1=v1, bsdf_in=rec, rays=rays, sincos_u=u, half_in=h)


def frame_params(pkg, k, depth, mat, flags=0):
    abi = pkg.abi
    gp = abi.GBufferParams()
    gp.pixelJitter[0], gp.pixelJitter[1] = pkg.msaa_jitter(0xdeadbeef + k)
    gp.frameCount = (0xdeadbeef + k) & 0xFFFFFFFF
    gp.focalLen, gp.lensRadius = 1.0, 1.0 / 64.0
    gp.envWidth = gp.envHeight = 128
    for i, c in enumerate((0.5, 0.5, 0.8, 1.0)):
        gp.envColor[i] = c
    p = abi.Params()
    p.minT, p.frameCount, p.matIndex, p.maxDepth = 1e-4, 0x1337 + k, mat, depth
    p.refractiveIndex, p.emitMult, p.clampUpper, p.flags = 1.0, 1.0, 0.9, flags
    p.pixelJitter[0], p.pixelJitter[1] = pkg.msaa_jitter(0x1337 + k)
    return gp, p


def camera_from_array(pkg, a):
    cam = pkg.abi.Camera()
    for i in range(3):
        cam.posW[i], cam.cameraU[i], cam.cameraV[i], cam.cameraW[i] = a[0, i], a[1, i], a[2, i], a[3, i]
    return cam


def camera_to_array(cam):
    return np.array([list(cam.posW), list(cam.cameraU), list(cam.cameraV), list(cam.cameraW)], np.float32)


def render(pkg, scene, cam, W, H, depth, mat, frames=1, flags=0, brute=False, threads=8):
    """Returns (sum of frames' resolved images, last splat buffer, last counters)."""
    orc = ob.OracleRender(pkg.abi, scene.desc, W, H)
    acc = np.zeros((H, W, 4), np.float64)
    cnt = None
    for k in range(frames):
        gp, p = frame_params(pkg, k, depth, mat, flags)
        of = ob.ORACLE_BRUTE_FORCE if brute else 0
        orc.gbuffer(cam, gp, flags=of, threads=threads)
        cnt = orc.bdpt(cam, p, flags=of, threads=threads)
        orc.resolve()
        acc += orc.image()
    out = (orc.image().copy(), orc.splat.copy(), cnt.as_dict(), {k: v.copy() for k, v in orc.chan.items()})
    orc.close()
    return out


def main():
    pkg = ge.load_package()
    abi = pkg.abi
    lib = ob.load_oracle(abi)
    out = {}
    k = kat_inputs(pkg)
    # (i) RNG streams
    draws = 12
    st = np.zeros((k["v0"].size, draws), np.uint32)
    fl = np.zeros((k["v0"].size, draws), np.float32)
    lib.oracle_rng(k["v0"].ctypes.data, k["v1"].ctypes.data, k["v0"].size, draws, st.ctypes.data, fl.ctypes.data)
    out.update(rng_v0=k["v0"], rng_v1=k["v1"], rng_states=st, rng_floats=fl)
    # (ii) BSDF KATs for matIndex 0 (GGX), 1 (Lambertian), 2 (GGX + isSpecular-from-lobe)
    out["bsdf_in"] = k["bsdf_in"]
    for mat in (0, 1, 2):
        o = np.zeros((k["bsdf_in"].shape[0], 16), np.float32)
        lib.oracle_bsdf(k["bsdf_in"].ctypes.data, k["bsdf_in"].shape[0], mat, o.ctypes.data)
        out[f"bsdf_out_{mat}"] = o
    # deterministic transcendental stand-ins and the half rounding
    s = np.zeros_like(k["sincos_u"])
    c = np.zeros_like(k["sincos_u"])
    lib.oracle_sincos2pi(k["sincos_u"].ctypes.data, k["sincos_u"].size, s.ctypes.data, c.ctypes.data)
    out.update(sincos_u=k["sincos_u"], sincos_s=s, sincos_c=c)
    hr = np.zeros_like(k["half_in"])
    lib.oracle_half_round(k["half_in"].ctypes.data, k["half_in"].size, hr.ctypes.data)
    out.update(half_in=k["half_in"], half_out=hr)
    # (iii) intersection records on a seeded soup, BVH and brute force
    soup = pkg.Scene.soup(11, 600, 0.3)
    osc = lib.oracle_scene_create(C.byref(soup.desc))
    out["trace_rays"] = k["rays"]
    for mode in (0, 1, 2):
        for name, fl_ in (("bvh", 0), ("brute", ob.ORACLE_BRUTE_FORCE)):
            prim = np.zeros(k["rays"].shape[0], np.int32)
            tuv = np.zeros((k["rays"].shape[0], 3), np.float32)
            lib.oracle_trace(osc, k["rays"].ctypes.data, k["rays"].shape[0], mode, fl_, prim.ctypes.data, tuv.ctypes.data)
            out[f"trace_prim_{mode}_{name}"] = prim
            out[f"trace_tuv_{mode}_{name}"] = tuv
    lib.oracle_scene_destroy(osc)
    # (iv)+(v) Cornell images and ray tallies.  The camera basis is stored (tan/atan come from libm).
    scene = pkg.Scene.cornell()
    cam = scene.camera(1.0)
    out["cornell_camera"] = camera_to_array(cam)
    tallies = {}
    for name, (size, depth, mat, flags) in {
        "cornell64_d3_ggx": (64, 3, 0, 0),
        "cornell64_d3_lambert": (64, 3, 1, 0),
        "cornell64_d3_ggx_nee_only": (64, 3, 0, abi.PARAM_NO_SPLAT | abi.PARAM_NO_CONNECT),
        "cornell64_d3_ggx_splat_only": (64, 3, 0, abi.PARAM_NO_NEE | abi.PARAM_NO_CONNECT),
        "cornell64_d3_ggx_connect_only": (64, 3, 0, abi.PARAM_NO_NEE | abi.PARAM_NO_SPLAT),
        "cornell48_d8_lambert": (48, 8, 1, 0),
        "cornell32_d5_ggx_lobe": (32, 5, 0, abi.PARAM_SPECULAR_FROM_LOBE),
        "cornell256_d3_ggx_config1": (256, 3, 0, 0),
    }.items():
        img, splat, cnt, chan = render(pkg, scene, cam, size, size, depth, mat, 1, flags, brute=(size <= 64))
        out[name + "_image"] = img
        out[name + "_splat"] = splat
        tallies[name] = cnt
        if name == "cornell64_d3_ggx":
            for ck, cv in chan.items():
                if ck != "out":
                    out["cornell64_gbuffer_" + ck] = cv
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **out)
    import json
    with open(os.path.join(HERE, "oracle_tallies.json"), "w") as f:
        json.dump(tallies, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "oracle_golden.npz"), os.path.getsize(os.path.join(HERE, "oracle_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
