#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r5e_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r5e_tests.log
[ $rc -ne 0 ] && exit $rc
BDPT_BUILD_VERBOSE=1 timeout -k 10 300 python tools/setup_times.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5e_setup.log | tail -45
timeout -k 10 600 python tools/loader_times.py 2800000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5e_loader.log
