#!/bin/bash
# scratch driver for one gpurun call
mkdir -p gpurun_out/r3c
export TMPDIR=/tmp
timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-other-configs --cpu-seconds 5 > gpurun_out/r3c/bench.log 2>&1 || { echo "bench failed"; tail -20 gpurun_out/r3c/bench.log; }
tail -c 1500 gpurun_out/r3c/bench.log; echo
for b in 4 8; do
  BDPT_SPLIT_BUDGET_ALPHA=$b timeout -k 10 300 python tools/big_configs.py 5 2>&1 | tail -1 > gpurun_out/r3c/big5_alpha$b.log; cat gpurun_out/r3c/big5_alpha$b.log
done
rocprofv3 -L > gpurun_out/r3c/counters.txt 2>&1; grep -o "SQ_INSTS_VALU[A-Z0-9_]*" gpurun_out/r3c/counters.txt | sort -u | tr '\n' ' '; echo
(cd /tmp && rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/r3c/calib_fetch -o run --output-format csv -- $GRAFT_REPO_ROOT/tools/microbench/fetch_calib > $GRAFT_REPO_ROOT/gpurun_out/r3c/calib_fetch.log 2>&1)
(cd /tmp && rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $GRAFT_REPO_ROOT/gpurun_out/r3c/calib_cache -o run --output-format csv -- $GRAFT_REPO_ROOT/tools/microbench/fetch_calib > $GRAFT_REPO_ROOT/gpurun_out/r3c/calib_cache.log 2>&1)
tail -3 gpurun_out/r3c/calib_fetch.log
python3 - <<'PY'
import csv,glob,collections
for d in ("gpurun_out/r3c/calib_fetch","gpurun_out/r3c/calib_cache"):
    fs=glob.glob(d+"/*counter_collection.csv")+glob.glob(d+"/*/*counter_collection.csv")
    if not fs: print(d,"no csv"); continue
    for r in csv.DictReader(open(fs[0])):
        print(r["Kernel_Name"][:40], r["Counter_Name"], r["Counter_Value"])
PY
