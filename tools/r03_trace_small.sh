#!/bin/bash
# kernel timeline of the 512-env pipeline: one / two wavefronts per env in k_sim with 8 sim steps per launch
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
for w in 1 2; do
  export TSIDB_SIM_WAVES=$w
  rocprofv3 --kernel-trace --output-format csv -d $O/r05_tr512_w$w -- python3 $GRAFT_REPO_ROOT/bench.py --envs 512 --steps 400 --cpu-seconds 0 --no-secondary --sim-batch 8 > $O/r05_tr512_w$w.json 2> $O/r05_tr512_w$w.err
  python3 $GRAFT_REPO_ROOT/tools/trace_timeline.py $O/r05_tr512_w$w 60
done
