#!/bin/bash
# tools/prof_stats.sh <tag> [bench args...] — rocprofv3 --kernel-trace --stats of `python3 bench.py`; csv summary under
# gpurun_out/<tag>_stats/.  Run on the GPU box.
set -e
tag=$1; shift
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_stats -o run --output-format csv -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-single-pass --no-other-configs "$@" > gpurun_out/${tag}_stats.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/${tag}_stats/*kernel_stats.csv")+glob.glob("gpurun_out/${tag}_stats/*/*kernel_stats.csv")
for r in csv.DictReader(open(f[0])):
    if float(r["Percentage"])>0.3: print("%-70s calls %5s avg %9.1f us  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
