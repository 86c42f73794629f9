#!/bin/bash
# tools/prof_pmc.sh <tag> [bench args...] — rocprofv3 PMC passes (own runs, no trace domains) + kernel-trace stats of
# `python3 bench.py`, written under gpurun_out/<tag>_*; summaries printed by tools/pmc_summary.py.
# Run on the GPU box:  gpurun -- 'bash tools/prof_pmc.sh r2x'
set -e
tag=$1; shift
args="--steps 2 --warmup 1 --no-cpu-baseline --no-single-pass --no-other-configs $*"
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  -d gpurun_out/${tag}_pmc_sq -o run --output-format csv -- python3 bench.py $args > gpurun_out/${tag}_pmc_sq.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum \
  -d gpurun_out/${tag}_pmc_cache -o run --output-format csv -- python3 bench.py $args > gpurun_out/${tag}_pmc_cache.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY \
  -d gpurun_out/${tag}_pmc_mix -o run --output-format csv -- python3 bench.py $args > gpurun_out/${tag}_pmc_mix.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 \
  -d gpurun_out/${tag}_pmc_valu -o run --output-format csv -- python3 bench.py $args > gpurun_out/${tag}_pmc_valu.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${tag}_pmc_fetch -o run --output-format csv -- python3 bench.py $args > gpurun_out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${tag}_pmc_write -o run --output-format csv -- python3 bench.py $args > gpurun_out/${tag}_pmc_write.log 2>&1
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_sq gpurun_out/${tag}_pmc_cache gpurun_out/${tag}_pmc_mix > gpurun_out/${tag}_pmc_summary.txt
python3 tools/mk_traffic.py gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write gpurun_out/${tag}_hbm_traffic_pmc.json >> gpurun_out/${tag}_pmc_summary.txt
python3 tools/roofline_pmc.py gpurun_out/${tag}_roofline_pmc.json gpurun_out/${tag}_pmc_sq gpurun_out/${tag}_pmc_cache gpurun_out/${tag}_pmc_mix \
  gpurun_out/${tag}_pmc_valu gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write >> gpurun_out/${tag}_pmc_summary.txt
cat gpurun_out/${tag}_pmc_summary.txt
