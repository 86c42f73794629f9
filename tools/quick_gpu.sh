#!/bin/bash
# tools/quick_gpu.sh <tag> — GPU parity tests, then a short bench; logs under gpurun_out/<tag>_*.log
tag=$1; shift
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; rc=$?
tail -4 gpurun_out/${tag}_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-single-pass --no-other-configs "$@" > gpurun_out/${tag}_bench.log 2>&1 || { tail -5 gpurun_out/${tag}_bench.log; exit 1; }
tail -1 gpurun_out/${tag}_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'])"
