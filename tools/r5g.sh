#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 bash tools/collect_profiles.sh r5 2>&1 | tail -30
timeout -k 10 240 python tools/tile_times.py > gpurun_out/profiles_r5/tile_times.log 2>&1; tail -12 gpurun_out/profiles_r5/tile_times.log
