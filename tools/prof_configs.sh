#!/bin/bash
# tools/prof_configs.sh <round> — rocprofv3 --kernel-trace --stats of the other single-GPU BASELINE shapes
# (tools/big_configs.py 2 | 4 | 5: one warm-up frame, timed frames, one statistics frame), one run each; the
# kernel_stats.csv of each goes to gpurun_out/profiles_<round>/config<N>/ with the script's own line beside it.
set -e
round=$1
export TMPDIR=/tmp
for cfg in 2 4 5; do
  out=gpurun_out/profiles_${round}/config${cfg}
  mkdir -p $out
  rocprofv3 --kernel-trace --stats -d gpurun_out/${round}c${cfg}_stats -o run --output-format csv -- python3 tools/big_configs.py $cfg > gpurun_out/${round}c${cfg}_stats.log 2>&1
  cp gpurun_out/${round}c${cfg}_stats/run_kernel_stats.csv $out/kernel_stats.csv
  grep "configs\[" gpurun_out/${round}c${cfg}_stats.log > $out/big_configs_line.txt || true
  cat $out/big_configs_line.txt
  python3 - <<PY
import csv
for r in csv.DictReader(open("$out/kernel_stats.csv")):
    if float(r["Percentage"]) > 2.0: print("  %-60s calls %5s avg %10.1f us  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
done
