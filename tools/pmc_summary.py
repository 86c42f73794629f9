#!/usr/bin/env python3
"""tools/pmc_summary.py [--json out.json] <pmc dir>... — per-kernel summary of rocprofv3 --pmc passes (csv).

Derived figures (kernels of this library only, summed over all dispatches of the run):
  lane_util   = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU): mean fraction of the 64 lanes active per VALU instruction
  valu_wave   = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES: fraction of a resident wave's cycles spent executing VALU
  waves_simd  = resident waves per SIMD: for the persistent kernels (grid sized by the occupancy query: at most 32 waves
                per CU) the launch grid / (256 CUs x 4 SIMDs); for the dense kernels the LDS limit, at most 8
                (the csv's VGPR_Count column is not the allocation the assembler reports, so it is not used)
  valu_busy   = valu_wave * waves_simd: how busy each SIMD's VALU is (about 1 = bound by VALU issue)
  wait_any    = SQ_WAIT_ANY / SQ_WAVE_CYCLES
  insts_valu_per_wave = SQ_INSTS_VALU / SQ_WAVES
  l2_hit      = TCC_HIT / (TCC_HIT + TCC_MISS);  l1_accesses = TCP_TOTAL_CACHE_ACCESSES
"""
import collections
import csv
import glob
import json
import sys


def short(name):
    name = name[:name.index("(")] if "(" in name else name
    return name.replace("void ", "").replace("bdpt::", "")


def main():
    args = sys.argv[1:]
    out_json = None
    if args and args[0] == "--json":
        out_json = args[1]
        args = args[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    res = {}
    for d in args:
        f = (glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"))[0]
        for r in csv.DictReader(open(f)):
            if "bdpt" not in r["Kernel_Name"]:
                continue
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add((d, r["Dispatch_Id"]))
            res[k] = (int(r["Grid_Size"]), int(r["LDS_Block_Size"]), int(r["Workgroup_Size"]))
    summary = {}
    for k, v in sorted(agg.items()):
        grid, lds, wg = res[k]
        waves_per_wg = max(1, (wg + 63) // 64)
        grid_waves = grid // 64
        by_lds = (163840 // lds) * waves_per_wg / 4.0 if lds else 8.0
        persistent = grid_waves <= 32 * 256
        s = {"grid_waves": grid_waves, "lds_bytes_per_workgroup": lds,
             "waves_per_simd": round(grid_waves / 1024.0, 2) if persistent else min(8.0, by_lds)}
        if "SQ_ACTIVE_INST_VALU" in v and v.get("SQ_WAVE_CYCLES"):
            s["lane_util"] = round(v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64), 3)
            s["valu_wave"] = round(v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], 3)
            s["valu_busy"] = round(s["valu_wave"] * s["waves_per_simd"], 3)
            s["wait_any"] = round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 3)
            s["insts_valu_per_wave"] = round(v["SQ_INSTS_VALU"] / v["SQ_WAVES"], 1)
            s["waves"] = int(v["SQ_WAVES"])
        if "TCC_HIT_sum" in v:
            s["l2_hit"] = round(v["TCC_HIT_sum"] / max(1.0, v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 3)
            s["l1_accesses"] = int(v["TCP_TOTAL_CACHE_ACCESSES_sum"])
            s["l1_to_l2_read_requests"] = int(v["TCP_TCC_READ_REQ_sum"])
        for c in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
            if c in v:
                s[c.lower()] = int(v[c])
        summary[k] = s
        print("%-34s %s" % (k[:34], " ".join("%s=%s" % kv for kv in s.items())))
    if out_json:
        json.dump({"note": __doc__.split("\n\n")[1], "kernels": summary}, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
