#!/bin/bash
# copy the judged summaries of a tools/final_profile.sh run from gpurun_out/<tag>_* into profiles/<round>_final_*
# usage: bash tools/collect_profiles.sh <tag> <round, e.g. r02>
set -e
TAG=${1:-r02}
RND=${2:-r02}
cd "$(dirname "$0")/.."
G=gpurun_out
cp $G/${TAG}_bench.json profiles/${RND}_final_bench.json
cp $G/${TAG}_bench_driver_window.json profiles/${RND}_final_bench_driver_window.json
cp $G/${TAG}_bench_no_overlap.json profiles/${RND}_final_bench_no_overlap.json
cp "$(ls -S $G/${TAG}_trace/*/*_kernel_stats.csv | head -1)" profiles/${RND}_final_kernel_stats_overlap.csv   # (largest: the bench process, not a helper child)
cp "$(ls -S $G/${TAG}_trace_serial/*/*_kernel_stats.csv | head -1)" profiles/${RND}_final_kernel_stats_serial.csv
cp $G/${TAG}_bench_under_rocprof.json profiles/${RND}_final_bench_under_rocprof_overlap.json
cp $G/${TAG}_bench_under_rocprof_serial.json profiles/${RND}_final_bench_under_rocprof_serial.json
python3 tools/pmc_summary.py $G/${TAG}_pmc --last 100 --traffic-json profiles/pmc_traffic.json > profiles/${RND}_final_pmc_walk_f64_last100.txt
python3 tools/pmc_summary.py $G/${TAG}_pmc > profiles/${RND}_final_pmc_walk_f64_all.txt
python3 tools/trace_window.py $G/${TAG}_trace_serial 1000 > profiles/${RND}_final_kernel_trace_timed_window_serial.txt
python3 tools/trace_window.py $G/${TAG}_trace 1000 > profiles/${RND}_final_kernel_trace_timed_window_overlap.txt
