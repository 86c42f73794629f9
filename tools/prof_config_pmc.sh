#!/bin/bash
# tools/prof_config_pmc.sh <round> <4|5> — the hardware counters behind the roofline block of the configs[3] (2.8 M
# triangles, 3840x2160, depth 12) or configs[4] (10 M triangles, half of them alpha-masked cards, 3840x2160, depth 16)
# shape on one GPU: five rocprofv3 --pmc passes of tools/big_configs.py <N>, each its own run without trace domains —
# SQ occupancy / lane utilisation / waits, L1 and L2 hits, the VALU instruction classes, FETCH_SIZE, WRITE_SIZE — and
# tools/roofline_pmc.py over them (counts per frame by kernel, stamped with the build's source hash): what bench.py's
# other_configs[].roofline quotes.  For configs[4] this is also BASELINE.md section 3 row 5's "any-hit rate, divergence
# counters" (big_configs.py's own line carries the alpha tests per ray).
# Results: gpurun_out/profiles_<round>/config<N>/{pmc_*.csv,pmc_summary.txt,sq_summary.json,roofline_pmc.json,big_configs_line.txt}
set -e
round=$1
cfg=$2
export TMPDIR=/tmp
out=gpurun_out/profiles_${round}/config${cfg}
tag=gpurun_out/${round}c${cfg}
mkdir -p $out
pass() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" -d ${tag}_pmc_${name} -o run --output-format csv -- python3 tools/big_configs.py ${cfg} > ${tag}_pmc_${name}.log 2>&1
  cp ${tag}_pmc_${name}/run_counter_collection.csv $out/pmc_${name}.csv 2>/dev/null || cp ${tag}_pmc_${name}/*/run_counter_collection.csv $out/pmc_${name}.csv
  echo "pass ${name} done"
}
pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass cache TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
pass valu SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32
pass fetch_size FETCH_SIZE
pass write_size WRITE_SIZE
grep "configs\[" ${tag}_pmc_sq.log > $out/big_configs_line.txt || true
python3 tools/pmc_summary.py --json $out/sq_summary.json ${tag}_pmc_sq ${tag}_pmc_cache > $out/pmc_summary.txt
python3 tools/roofline_pmc.py $out/roofline_pmc.json --command="python3 tools/big_configs.py ${cfg} (tools/prof_config_pmc.sh)" \
  ${tag}_pmc_sq ${tag}_pmc_cache ${tag}_pmc_valu ${tag}_pmc_fetch_size ${tag}_pmc_write_size >> $out/pmc_summary.txt
cat $out/big_configs_line.txt $out/pmc_summary.txt
