#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r5i_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r5i_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/device_tree_check.py big 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5i_devtree.log | grep -c SAME
grep -v SAME gpurun_out/r5i_devtree.log | head -5
BDPT_BUILD_VERBOSE=1 timeout -k 10 300 python tools/setup_times.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5i_setup.log | grep -E "refs upload|refs make|levels|buildSceneBvh|indices, textures|scene_first|resize_first|priorities|light maps"
timeout -k 10 300 bash tools/prof_config_pmc.sh r5 4 2>&1 | tail -3
timeout -k 10 400 bash tools/prof_config_pmc.sh r5 5 2>&1 | tail -3
