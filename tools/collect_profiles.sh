#!/bin/bash
# tools/collect_profiles.sh <round> — everything bench.py's roofline block quotes, measured on this build with the
# very command the driver runs (python3 bench.py, default workload), copied under profiles/<round>/:
#   kernel_stats.csv        rocprofv3 --kernel-trace --stats
#   pmc_sq.csv, pmc_cache.csv, pmc_mix.csv, pmc_fetch_size.csv, pmc_write_size.csv   rocprofv3 --pmc passes (own runs)
#   sq_summary.json         tools/pmc_summary.py over the three SQ/cache passes
#   hbm_traffic_pmc.json    tools/mk_traffic.py over the FETCH_SIZE / WRITE_SIZE passes
#   bench.json              the bench line of the same build (no profiler attached)
# Run on the GPU box (gpurun -- 'bash tools/collect_profiles.sh r2'); results come back under gpurun_out/profiles_<round>/.
set -e
round=$1
out=gpurun_out/profiles_${round}
mkdir -p $out
bash tools/prof_pmc.sh ${round}p > $out/pmc_summary.txt 2>&1
bash tools/prof_stats.sh ${round}p > $out/kernel_stats_summary.txt 2>&1
cp gpurun_out/${round}p_stats/run_kernel_stats.csv $out/kernel_stats.csv
# the same command with one frame in flight: kernels never overlap another frame's, so the averages are the solo
# durations bench.py's roofline block measures with HIP events (its stage frames run alone)
bash tools/prof_stats.sh ${round}p1 --inflight 1 > $out/kernel_stats_inflight1_summary.txt 2>&1
cp gpurun_out/${round}p1_stats/run_kernel_stats.csv $out/kernel_stats_inflight1.csv
cp gpurun_out/${round}p_pmc_sq/run_counter_collection.csv $out/pmc_sq.csv
cp gpurun_out/${round}p_pmc_cache/run_counter_collection.csv $out/pmc_cache.csv
cp gpurun_out/${round}p_pmc_mix/run_counter_collection.csv $out/pmc_mix.csv
cp gpurun_out/${round}p_pmc_valu/run_counter_collection.csv $out/pmc_valu.csv
cp gpurun_out/${round}p_roofline_pmc.json $out/roofline_pmc.json
cp gpurun_out/${round}p_pmc_fetch/run_counter_collection.csv $out/pmc_fetch_size.csv
cp gpurun_out/${round}p_pmc_write/run_counter_collection.csv $out/pmc_write_size.csv
cp gpurun_out/${round}p_hbm_traffic_pmc.json $out/hbm_traffic_pmc.json
python3 tools/pmc_summary.py --json $out/sq_summary.json gpurun_out/${round}p_pmc_sq gpurun_out/${round}p_pmc_cache gpurun_out/${round}p_pmc_mix > /dev/null
mkdir -p profiles/${round}
cp $out/hbm_traffic_pmc.json $out/sq_summary.json $out/roofline_pmc.json profiles/${round}/   # so the bench line below already quotes them
python3 bench.py --steps 16 --warmup 3 2> /dev/null | tail -1 > $out/bench.json
cat $out/kernel_stats_summary.txt | tail -14
tail -c 1500 $out/bench.json
