#!/bin/bash
# second half of the pool experiment: fewer LDS rows per ray (more resident waves), then the SQ counters of the best
# variant and of the default build for the README row (lane utilisation, VALU wave-instructions per ray)
bash tools/variants.sh "-DBDPT_POOL_ANYHIT=1 -DBDPT_POOL_STACK=10" "-DBDPT_POOL_ANYHIT=1 -DBDPT_POOL_STACK=8" "-DBDPT_POOL_ANYHIT=1 -DBDPT_POOL_STACK=8 -DBDPT_POOL_NODE_MIN=40 -DBDPT_POOL_REFILL=24"
export TMPDIR=/tmp
for v in "default:" "pool:-DBDPT_POOL_ANYHIT=1 -DBDPT_POOL_STACK=12"; do
  tag=${v%%:*}; flags=${v#*:}
  (cd fyp-bidirectionalpathtracer_amd/csrc && make EXTRA="$flags" > /tmp/b.log 2>&1) || { tail -5 /tmp/b.log; continue; }
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    -d gpurun_out/r4_pool_pmc_$tag -o run --output-format csv -- python3 tools/stages.py > gpurun_out/r4_pool_pmc_$tag.log 2>&1
  echo "== $tag [$flags]"
  python3 tools/pmc_summary.py gpurun_out/r4_pool_pmc_$tag | grep -i "trace_shadow"
done
(cd fyp-bidirectionalpathtracer_amd/csrc && make > /dev/null 2>&1)
