#!/bin/bash
# tools/prof_tile.sh <round> [tile_loop args...] — rocprofv3 --kernel-trace --stats of ONE rank's 1/8 tile of the 1080p
# frame loop, three frames in flight (tools/tile_loop.py): where the time of a rank goes at world 8 before any exchange.
# Results: gpurun_out/profiles_<round>/tile8/{kernel_stats.csv,kernel_trace.csv,timeline.txt,tile_loop.json}
set -e
round=$1; shift
tag=${TILE_TAG:-tile8}
export TMPDIR=/tmp
out=gpurun_out/profiles_${round}/${tag}
mkdir -p $out
rocprofv3 --kernel-trace --stats -d gpurun_out/${round}_${tag}_stats -o run --output-format csv -- python3 tools/tile_loop.py "$@" > gpurun_out/${round}_${tag}.log 2>&1
cp gpurun_out/${round}_${tag}_stats/run_kernel_stats.csv $out/kernel_stats.csv
cp gpurun_out/${round}_${tag}_stats/run_kernel_trace.csv $out/kernel_trace.csv
grep '^{' gpurun_out/${round}_${tag}.log | tail -1 > $out/tile_loop.json
python3 tools/timeline_summary.py $out/kernel_trace.csv 16 > $out/timeline.txt
cat $out/tile_loop.json $out/timeline.txt
