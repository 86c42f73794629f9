#!/bin/bash
# tools/prof_pmc_mem.sh <tag> — SQ-side view of the vector-memory pipeline (TA FIFO-full stalls, VMEM issue cycles and
# levels, instruction fetch, LDS conflicts) for `python3 bench.py`; each pass is its own rocprofv3 --pmc run without trace
# domains, under a timeout (the TA_* / TCP_* stall counters of this ROCm abort rocprofv3 on gfx950 and are not used).
tag=$1; shift
args="--steps 2 --warmup 1 --no-cpu-baseline --no-single-pass --no-other-configs $*"
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_IFETCH" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU_CVT"; do
  i=$((i+1))
  echo "pass $i: $set" >> gpurun_out/${tag}_progress.txt
  timeout -k 10 200 rocprofv3 --pmc $set -d gpurun_out/${tag}_m$i -o run --output-format csv -- python3 bench.py $args > gpurun_out/${tag}_m$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/${tag}_m$i.log; }
done
python3 tools/pmc_sum.py gpurun_out/${tag}_m1 gpurun_out/${tag}_m2 gpurun_out/${tag}_m3 > gpurun_out/${tag}_mem_summary.txt
cat gpurun_out/${tag}_mem_summary.txt
