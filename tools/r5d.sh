#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5d_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r5d_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 1000 bash tools/collect_profiles.sh r5 2>&1 | tail -25
