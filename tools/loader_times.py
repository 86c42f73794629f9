#!/usr/bin/env python3
"""tools/loader_times.py [triangles] — the model loader (host/SceneLoader.cpp) on an exported atrium: `tools/export_obj.py`'s
files read back by bdpt_scene_load with 1, 2, 8 and 16 host threads; prints seconds, triangles per second and a digest
of the loaded arrays per thread count (the digests must agree: the scene does not depend on the thread count)."""
import hashlib
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import __graft_entry__ as ge
import export_obj as ex

pkg = ge.load_package()
tris = int(sys.argv[1]) if len(sys.argv) > 1 else 2800000
src = pkg.Scene.atrium(1, tris)
out = tempfile.mkdtemp(prefix="bdpt_export_")
t0 = time.time()
path = ex.export_scene(pkg, src, out)
size = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))
print("exported %d triangles: %.0f MB of OBJ / MTL / PNG in %.1f s" % (src.desc.numTriangles, size / 1e6, time.time() - t0), flush=True)
pkg.Scene.load(path, threads=16).close()  # (page cache warm: the figures below are the loader's, not the disk's)
digests = []
for threads in (1, 2, 8, 16):
    t0 = time.time()
    sc = pkg.Scene.load(path, threads=threads)
    dt = time.time() - t0
    a = ex.scene_arrays(sc.desc)
    h = hashlib.sha256()
    for k in ("positions", "normals", "bitangents", "texcoords", "indices", "tri_material"):
        h.update(np.ascontiguousarray(a[k]).tobytes())
    digests.append(h.hexdigest())
    print("%2d threads: %.3f s, %.2f M triangles/s, %d vertices, digest %s" % (threads, dt, sc.desc.numTriangles / dt / 1e6, sc.desc.numVertices, digests[-1][:16]), flush=True)
    sc.close()
print("same scene for every thread count:", len(set(digests)) == 1)
