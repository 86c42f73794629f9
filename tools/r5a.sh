#!/bin/bash
# round-5 first GPU call: parity with hints on, stage times with / without hints and by light-map resolution
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5a_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r5a_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tests/stage_parity_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5a_probe.log
for v in "BDPT_NO_HINTS=1" "BDPT_LIGHT_MAP_RES=0" "BDPT_LIGHT_MAP_RES=128" "BDPT_LIGHT_MAP_RES=256" "BDPT_LIGHT_MAP_RES=512" "BDPT_LIGHT_MAP_RES=1024" "BDPT_LIGHT_MAP_RES=2048"; do
  echo "== $v"
  env $v timeout -k 10 180 python tools/stages.py 2>&1 | grep -v amdgpu.ids | tail -1
done | tee gpurun_out/r5a_stages.log
for v in "BDPT_NO_HINTS=1" "BDPT_LIGHT_MAP_RES=512"; do
  echo "== bench $v"
  env $v timeout -k 10 300 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-single-pass --no-other-configs 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'])"
done | tee gpurun_out/r5a_bench.log
