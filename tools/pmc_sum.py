#!/usr/bin/env python3
"""Per-kernel sums of every counter found in rocprofv3 --pmc csv outputs (directories given on the command line)."""
import csv, glob, os, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void bdpt::", "").replace("bdpt::", "")
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k in sorted(tot):
    if not any(s in k for s in ("walk_kernel", "trace_shadow", "gen_", "gather", "lazy")):
        continue
    print(k)
    for c in sorted(tot[k]):
        print("   %-44s %16.0f   (%d dispatches)" % (c, tot[k][c], len(disp[(k, c)])))
