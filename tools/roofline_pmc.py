#!/usr/bin/env python3
"""tools/roofline_pmc.py <out.json> <pmc dir>... — per kernel and PER FRAME hardware counts of rocprofv3 --pmc passes
of `python3 bench.py` (each pass its own run, csv): what bench.py's roofline block divides by its live kernel
durations.  Frames = dispatches of gbuffer_kernel in the pass; a counter that appears in several passes is averaged.
FETCH_SIZE / WRITE_SIZE are converted to bytes (counter x 1024); FETCH_SIZE stays RAW here — on gfx950 it tallies
64 B per 128-byte request for streams AND for 16-byte gathers (tools/microbench/fetch_calib.hip,
profiles/r3/fetch_calib.txt), so `fetch_factor` = 2 is recorded for every kernel class and applied by bench.py.
SQ_INSTS_VALU_FAST = ADD_F32 + MUL_F32 + FMA_F32: the instruction classes that issue in about 2.4 clocks per wave
(tools/microbench/valu_rate.hip); everything else is priced at 4.3."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(name):
    name = name[:name.index("(")] if "(" in name else name
    return name.replace("void ", "").replace("bdpt::", "")


def main():
    out_json, dirs = sys.argv[1], sys.argv[2:]
    command = "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-pass --no-other-configs (tools/prof_pmc.sh)"
    if dirs and dirs[0].startswith("--command="):
        command = dirs.pop(0)[len("--command="):]
    per_pass = []
    for d in dirs:
        fs = glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv")
        if not fs:
            print("no csv under", d)
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(fs[0])):
            if "bdpt" not in r["Kernel_Name"]:
                continue
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
        frames = max((len(v) for k, v in disp.items() if k.startswith("gbuffer_kernel")), default=0)
        if frames:
            per_pass.append((frames, agg, disp))
    kernels = collections.defaultdict(lambda: collections.defaultdict(list))
    launches = collections.defaultdict(list)
    for frames, agg, disp in per_pass:
        for k, cs in agg.items():
            launches[k].append(len(disp[k]) / frames)
            for c, v in cs.items():
                kernels[k][c].append(v / frames)
    out = {"command": "rocprofv3 --pmc <counters> -- " + command + " (one run per counter set)",
           "unit": "counts per frame (all dispatches of the kernel in a frame summed); *_bytes = counter x 1024",
           # the build the counts belong to: bench.py drops them (pmc_build_match false) when its sources differ
           "source_hash": __import__("__graft_entry__").load_package().source_hash(),
           "fetch_factor": {"walk_kernel": 2.0, "trace_shadow_kernel": 2.0, "gen_kernels": 2.0, "per_pixel_kernels": 2.0},
           "kernels": {}}
    for k in sorted(kernels):
        e = {"dispatches_per_frame": round(sum(launches[k]) / len(launches[k]), 3)}
        for c, vs in sorted(kernels[k].items()):
            v = sum(vs) / len(vs)
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                e[c + "_bytes"] = v * 1024.0
            else:
                e[c] = v
        if all(c in e for c in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32")):
            e["SQ_INSTS_VALU_FAST"] = e["SQ_INSTS_VALU_ADD_F32"] + e["SQ_INSTS_VALU_MUL_F32"] + e["SQ_INSTS_VALU_FMA_F32"]
        out["kernels"][k] = e
    # Per frame by kernel NAME.  A frame runs ONE template variant of each kernel (the statistics frame the variant with
    # the visit counters), so a variant's counts over ALL frames of the run understate what one of ITS frames costs
    # (round 4's file: walk_kernel<true, false, false> ran in 6 of the 7 frames, 0.857 dispatches "per frame", and every
    # count was 14 % low); summed over the variants of a name the counts are whole frames.
    by_name = collections.defaultdict(lambda: collections.defaultdict(float))
    for k, e in out["kernels"].items():
        base = k.split("<")[0]
        for c, v in e.items():
            by_name[base][c] += v
    out["by_name"] = {k: dict(v) for k, v in sorted(by_name.items())}
    json.dump(out, open(out_json, "w"), indent=1)
    for k, e in out["kernels"].items():
        if e["dispatches_per_frame"] >= 0.5:
            print("%-34s launches/frame %5.2f  valu %.3g (fast %.3g)  l1 %.3g  l1->l2 %.3g  l2miss %.3g  fetch %.1f MB  write %.1f MB" % (
                k[:34], e["dispatches_per_frame"], e.get("SQ_INSTS_VALU", 0), e.get("SQ_INSTS_VALU_FAST", 0), e.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0),
                e.get("TCP_TCC_READ_REQ_sum", 0), e.get("TCC_MISS_sum", 0), e.get("FETCH_SIZE_bytes", 0) / 1e6, e.get("WRITE_SIZE_bytes", 0) / 1e6))


if __name__ == "__main__":
    main()
