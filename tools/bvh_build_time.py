#!/usr/bin/env python3
"""Host BVH build time by thread count (bdpt_bvh_build_hash; the hash must not change)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
lib = pkg.load_library()
sizes = [int(x) for x in sys.argv[1:]] or [262144, 2_800_000, 10_000_000]
for n in sizes:
    t = time.time()
    sc = pkg.Scene.courtyard(1, n, 0.5) if n >= 5_000_000 else pkg.Scene.atrium(1, n)
    gen = time.time() - t
    row = []
    hs = set()
    for th in (1, 4, 8, 16):
        h = C.c_uint64()
        info = pkg.abi.BvhInfo()
        t = time.time()
        assert lib.bdpt_bvh_build_hash(C.byref(sc.desc), th, C.byref(h), C.byref(info)) == 0
        row.append("%d thr %.2f s" % (th, time.time() - t))
        hs.add(h.value)
    print("%d triangles (scene generated in %.1f s, default threads %d): %s; one tree: %s" %
          (n, gen, info.reserved, ", ".join(row), len(hs) == 1), flush=True)
    sc.close()
