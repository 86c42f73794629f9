#!/usr/bin/env python3
"""tools/roofline_table.py <bench.json> — the roofline blocks of a bench line as markdown tables (what DESIGN.md section 7 and
profiles/README.md quote): per kernel block ms per frame, algorithmic bytes, the contract's frac, the HBM-side frac and the
ceiling fractions; the same for every other_configs entry that carries PMC counts."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])


def table(name, roof):
    print("**%s**" % name)
    print()
    print("| kernel block | ms / frame | algorithmic GB / frame | `frac` (algorithmic / 8 TB/s) | `hbm_side_frac` (2 x FETCH + WRITE) | l2_gather | infinity_cache_gather | vmem_address_unit | valu_issue | bound |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for k, v in roof["kernels"].items():
        fr = v.get("fractions") or {}
        f = lambda x: "-" if x is None else ("%.2f" % x)
        print("| `%s` | %.2f | %.1f | %.2f | %s | %s | %s | %s | %s | %s |" % (
            k, v["ms_per_frame"], v["bytes_per_frame"] / 1e9, v["frac"], f(v.get("hbm_side_frac")), f(fr.get("l2_gather")),
            f(fr.get("infinity_cache_gather")), f(fr.get("vmem_address_unit")), f(fr.get("valu_issue")), v.get("bound", "-")))
    fr = roof["frame"]
    print("| frame (critical path, waits %.2f ms) | %.2f | %.1f | %.2f | %s | | | | | |" % (
        roof.get("waits_ms_per_frame", 0.0), fr["ms_per_frame"], fr["bytes_per_frame"] / 1e9, fr["frac"],
        "-" if not fr.get("traffic") else "%.2f" % (fr["traffic"] / (fr["ms_per_frame"] * 1e-3) / 1e9 / 8000.0)))
    print()


print("value %.1f %s, %.3f ms per step, %d rays per frame of which %d answered by a hint (traversals only: %.1f), pmc_build_match %s" % (
    d["value"], d["unit"], d["ms_per_step"], d["config"]["rays_per_frame"], d["config"].get("rays_hinted_per_frame", 0),
    d["config"].get("value_traversals_only", 0.0), d["roofline"].get("pmc_build_match")))
print()
table(d["config"]["workload"], d["roofline"])
for o in d["config"].get("other_configs", []):
    if o.get("roofline", {}).get("pmc_build_match"):
        table("%s — %.1f ms per frame, %.0f Mrays/s" % (o["workload"], o["ms_per_frame"], o["value"]), o["roofline"])
