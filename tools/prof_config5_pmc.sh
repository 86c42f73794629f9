#!/bin/bash
# tools/prof_config5_pmc.sh <round> — the "any-hit rate, divergence counters" BASELINE.md section 3 row 5 asks for, on the
# configs[4] shape (10 M triangles, half of them alpha-masked leaf cards, 3840x2160, depth 16) on one GPU: two rocprofv3
# --pmc passes (SQ occupancy / lane utilisation / waits; L1 and L2 hits), each its own run without trace domains, of
# tools/big_configs.py 5, whose own line carries the alpha tests per ray (device counters, BDPT_PARAM_COUNTERS).
# Results: gpurun_out/profiles_<round>/config5/{pmc_sq.csv,pmc_cache.csv,pmc_summary.txt,big_configs_line.txt}
set -e
round=$1
export TMPDIR=/tmp
out=gpurun_out/profiles_${round}/config5
mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  -d gpurun_out/${round}c5_pmc_sq -o run --output-format csv -- python3 tools/big_configs.py 5 > gpurun_out/${round}c5_pmc_sq.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum \
  -d gpurun_out/${round}c5_pmc_cache -o run --output-format csv -- python3 tools/big_configs.py 5 > gpurun_out/${round}c5_pmc_cache.log 2>&1
cp gpurun_out/${round}c5_pmc_sq/run_counter_collection.csv $out/pmc_sq.csv
cp gpurun_out/${round}c5_pmc_cache/run_counter_collection.csv $out/pmc_cache.csv
grep "configs\[" gpurun_out/${round}c5_pmc_sq.log > $out/big_configs_line.txt || true
python3 tools/pmc_summary.py --json $out/sq_summary.json gpurun_out/${round}c5_pmc_sq gpurun_out/${round}c5_pmc_cache > $out/pmc_summary.txt
cat $out/big_configs_line.txt $out/pmc_summary.txt
