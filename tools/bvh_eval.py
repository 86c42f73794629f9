#!/usr/bin/env python3
"""Tree quality without a GPU: node visits and triangle tests per ray of the acceleration structure bdpt_set_scene
builds, measured with the host-side trace hook (bdpt_host_bvh_*: the device's query semantics walked on the CPU) on
a ray sample shaped like the pass's own: camera rays (closest hit, culled), one bounce from their hit points
(closest hit), shadow rays from those points to the lights and between pairs of them (any hit).

Usage: python tools/bvh_eval.py [--scene atrium|courtyard] [--triangles N] [--rays K] [--variants "b,ba,c;..."]
       a variant is splitBudget,splitBudgetAlpha,classify (negative budget = build default)
Every variant's hits are compared with the first one's (prim and t must agree exactly)."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
lib = pkg.load_library()


def camera_rays(cam, W, H, n, rng):
    px = rng.integers(0, W, n)
    py = rng.integers(0, H, n)
    U, V, Wv, pos = (np.array(list(x), np.float32) for x in (cam.cameraU, cam.cameraV, cam.cameraW, cam.posW))
    ndx = 2.0 * ((px + 0.5) / W) - 1.0
    ndy = -2.0 * ((py + 0.5) / H) + 1.0
    d = ndx[:, None] * U + ndy[:, None] * V + Wv
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.zeros((n, 8), np.float32)
    r[:, 0:3] = pos
    r[:, 3:6] = d
    r[:, 6] = 0.0
    r[:, 7] = 1e38
    return r


def trace(h, rays, mode, brute=0):
    n = rays.shape[0]
    prim = np.zeros(n, np.int32)
    tuv = np.zeros((n, 3), np.float32)
    vis = (C.c_uint64 * 2)()
    rc = lib.bdpt_host_bvh_trace(h, rays.ctypes.data, n, mode, brute, 0, prim.ctypes.data, tuv.ctypes.data, vis)
    assert rc == 0
    return prim, tuv, (vis[0], vis[1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="courtyard")
    ap.add_argument("--triangles", type=int, default=1000000)
    ap.add_argument("--foliage", type=float, default=0.5)
    ap.add_argument("--rays", type=int, default=100000)
    ap.add_argument("--variants", default="0,0,0;-1,-1,1")
    args = ap.parse_args()
    scene = (pkg.Scene.courtyard(2, args.triangles, args.foliage) if args.scene == "courtyard" else
             pkg.Scene.atrium_uneven(1, args.triangles) if args.scene == "atrium_uneven" else pkg.Scene.atrium(1, args.triangles))
    cam = scene.camera(16 / 9)
    rng = np.random.default_rng(7)
    ref = None
    sets = None
    for v in args.variants.split(";"):
        b, ba, cl = v.split(",")
        info = pkg.abi.BvhInfo()
        t0 = time.time()
        h = lib.bdpt_host_bvh_create(C.byref(scene.desc), 0, float(b), float(ba), int(cl), C.byref(info))
        assert h
        build_s = time.time() - t0
        if sets is None:
            prim_rays = camera_rays(cam, 1920, 1080, args.rays, rng)
            prim, tuv, _ = trace(h, prim_rays, 1)
            ok = prim >= 0
            P = prim_rays[ok, 0:3] + prim_rays[ok, 3:6] * tuv[ok, 0:1]
            n = P.shape[0]
            d = rng.normal(size=(n, 3)).astype(np.float32)
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            bounce = np.zeros((n, 8), np.float32)
            bounce[:, 0:3] = P
            bounce[:, 3:6] = d
            bounce[:, 6] = 1e-4
            bounce[:, 7] = 1e38
            bp, bt, _ = trace(h, bounce, 0)
            ok2 = bp >= 0
            Q = bounce[ok2, 0:3] + bounce[ok2, 3:6] * bt[ok2, 0:1]
            # second bounce from Q, so the sample also holds rays that start inside the foliage
            m = Q.shape[0]
            d2 = rng.normal(size=(m, 3)).astype(np.float32)
            d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
            bounce2 = np.zeros((m, 8), np.float32)
            bounce2[:, 0:3] = Q
            bounce2[:, 3:6] = d2
            bounce2[:, 6] = 1e-4
            bounce2[:, 7] = 1e38
            closest = np.concatenate([bounce, bounce2])
            lights = np.array([list(scene.desc.lights[i].posW) for i in range(scene.desc.numLights)], np.float32)
            A = np.concatenate([P, Q])
            B = np.concatenate([lights[rng.integers(0, len(lights), P.shape[0])], P[rng.integers(0, n, m)]])
            dv = B - A
            ln = np.linalg.norm(dv, axis=1, keepdims=True)
            shadow = np.zeros((A.shape[0], 8), np.float32)
            shadow[:, 0:3] = A
            shadow[:, 3:6] = dv / np.maximum(ln, 1e-20)
            shadow[:, 6] = 1e-4
            shadow[:, 7] = ln[:, 0]
            sets = (prim_rays, closest, shadow)
        res = []
        t0 = time.time()
        line = []
        for name, rays, mode in (("primary", sets[0], 1), ("closest", sets[1], 0), ("shadow", sets[2], 2)):
            p, t, vis = trace(h, rays, mode)
            res.append((p if mode != 2 else (p >= 0), t))
            line.append("%s %.1f nodes %.1f tris" % (name, vis[0] / rays.shape[0], vis[1] / rays.shape[0]))
        same = ""
        if ref is None:
            ref = res
        else:
            bad = sum(int((a[0] != b[0]).sum() + (a[1] != b[1]).sum()) for a, b in zip(ref, res))
            same = " | differs from variant 0 in %d values" % bad if bad else " | hits identical to variant 0"
        print("[%s] build %.1f s: %d nodes, %d refs for %d tris (%d alpha-mode, %d always pass, %d dropped), sah %.1f, depth %d | %s | trace %.1f s%s" % (
            v, build_s, info.numNodes, info.numReferences, info.numTriangles, info.numAlphaMode, info.numAlwaysPass, info.numDropped,
            info.sahCost, info.maxDepth, " | ".join(line), time.time() - t0, same), flush=True)
        lib.bdpt_host_bvh_destroy(h)


if __name__ == "__main__":
    main()
