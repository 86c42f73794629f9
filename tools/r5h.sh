#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r5h_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r5h_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/device_tree_check.py big 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5h_devtree.log | tail -16
timeout -k 10 200 python tools/device_build_fuzz.py 150 11 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r5h_fuzz.log
BDPT_BUILD_VERBOSE=1 timeout -k 10 300 python tools/setup_times.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5h_setup.log | grep -E "refs upload|refs make|levels|buildSceneBvh|indices, textures|scene_first|resize_first|priorities"
