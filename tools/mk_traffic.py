#!/usr/bin/env python3
"""tools/mk_traffic.py <fetch_dir> <write_dir> <out.json> — HBM-side bytes per frame per kernel from two rocprofv3
--pmc passes (FETCH_SIZE and WRITE_SIZE, each in its own run of the same bench.py command).  Raw counters x 1024
(the counters are in KB); FETCH_SIZE is NOT doubled here (bench.py applies the gfx950 factor of MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    f = (glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"))[0]
    tot = collections.defaultdict(float)
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        k = k[:k.index("(")] if "(" in k else k
        tot[k] += float(r["Counter_Value"]) * 1024.0
        n[k].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in n.items()}


fe, nf = load(sys.argv[1], "FETCH_SIZE")
wr, nw = load(sys.argv[2], "WRITE_SIZE")
frames = [v for k, v in nf.items() if "gbuffer_kernel" in k][0]
out = {"frames": frames, "unit": "bytes per frame (raw counters x 1024; FETCH_SIZE not yet doubled)",
       "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate runs) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-pass --no-other-configs"}
for k in sorted(fe):
    if "bdpt" in k:
        out[k] = {"fetch_bytes_per_frame": fe[k] / frames, "write_bytes_per_frame": wr.get(k, 0.0) / frames,
                  "launches_per_frame": nf[k] / frames}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    if isinstance(v, dict):
        print("%-50s fetch %8.1f MB  write %8.1f MB  launches/frame %.1f" % (k[:50], v["fetch_bytes_per_frame"] / 1e6, v["write_bytes_per_frame"] / 1e6, v["launches_per_frame"]))
