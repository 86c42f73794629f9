#!/bin/bash
# round-5 second GPU call: (1) hit rate of a temporal (pixel, slot) occluder cache (throwaway probe build);
# (2) the configs[3] / configs[4] shapes with and without occluder hints
set -o pipefail
mkdir -p gpurun_out
( cd fyp-bidirectionalpathtracer_amd/csrc && make EXTRA="-DBDPT_OCC_PROBE=1" > /tmp/probe_build.log 2>&1 ) || { tail -5 /tmp/probe_build.log; exit 1; }
BDPT_LIGHT_MAP_RES=0 timeout -k 10 300 python tools/stages.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r5b_probe.log
( cd fyp-bidirectionalpathtracer_amd/csrc && make > /tmp/probe_build.log 2>&1 ) || { tail -5 /tmp/probe_build.log; exit 1; }
for v in "BDPT_NO_HINTS=1" "BDPT_LIGHT_MAP_RES=512" "BDPT_LIGHT_MAP_RES=1024"; do
  echo "== $v"
  env $v timeout -k 10 500 python tools/big_configs.py both 2>&1 | grep -v amdgpu.ids | tail -2
done | tee gpurun_out/r5b_big.log
