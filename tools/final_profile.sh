#!/bin/bash
# Round profile set (run on the GPU box): clean bench lines, kernel-trace stats, PMC passes.
# usage: bash tools/final_profile.sh <tag>   -> files under gpurun_out/<tag>_*
set -e
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
python3 bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python3 bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver_window.json 2>> $O/${TAG}_bench.err
python3 bench.py --no-overlap --cpu-seconds 0 --no-secondary > $O/${TAG}_bench_no_overlap.json 2>> $O/${TAG}_bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -- python3 $R/bench.py --steps 1000 --warmup 20 --cpu-seconds 0 --no-secondary > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace_serial -- python3 $R/bench.py --steps 1000 --warmup 20 --cpu-seconds 0 --no-overlap --no-secondary > $O/${TAG}_bench_under_rocprof_serial.json 2>> $O/${TAG}_trace.err
cd $R
bash tools/pmc_profile.sh ${TAG}_pmc
python3 tools/pmc_summary.py $O/${TAG}_pmc > $O/${TAG}_pmc_summary.txt
find $O/${TAG}_trace $O/${TAG}_trace_serial -name "*kernel_stats.csv" | head
