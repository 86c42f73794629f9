#!/bin/bash
# tools/r4_pool.sh — round-4 experiment: the any-hit kernel with rays unbound from lanes (device_trace.hpp
# trace_shadow_pool_kernel, -DBDPT_POOL_ANYHIT=1).  Parity first (the tests that compare frames and visibility bytes with
# the oracle, run on the variant build), then stage times of the bench frame per variant, then the default build again.
cd fyp-bidirectionalpathtracer_amd/csrc
make EXTRA="-DBDPT_POOL_ANYHIT=1" > /tmp/pool_build.log 2>&1 || { tail -20 /tmp/pool_build.log; exit 1; }
(cd ../.. && timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "trace or anyhit or cornell or atrium or config3 or config5_alpha or partial or striped" > gpurun_out/r4_pool_tests.log 2>&1; tail -3 gpurun_out/r4_pool_tests.log)
cd ../..
bash tools/variants.sh "" "-DBDPT_POOL_ANYHIT=1" "-DBDPT_POOL_ANYHIT=1 -DBDPT_POOL_STACK=12" "-DBDPT_POOL_ANYHIT=1 -DBDPT_POOL_NODE_MIN=32" "-DBDPT_POOL_ANYHIT=1 -DBDPT_POOL_NODE_MIN=64" \
  "-DBDPT_POOL_ANYHIT=1 -DBDPT_POOL_SLOTS=192 -DBDPT_POOL_STACK=12" "-DBDPT_POOL_ANYHIT=1 -DBDPT_NODE_BURST=2" "-DBDPT_POOL_ANYHIT=1 -DBDPT_NODE_BURST=4" "-DBDPT_POOL_ANYHIT=1 -DBDPT_POOL_REFILL=16"
