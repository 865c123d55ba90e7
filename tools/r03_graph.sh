#!/bin/bash
# HIP graph replay against eager step_pipelined at small batches: steps per graph x sim steps per launch inside the graph
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
run() { python3 bench.py "$@" --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gsb', os.environ.get('TSIDB_GRAPH_SIM_BATCH'), '$*', round(d['value']/1e6,3), 'M', round(d['ms_per_step'],4))"; }
for n in 512 1024; do
  run --envs $n --steps 1280
  for gsb in 1 2 8; do
    export TSIDB_GRAPH_SIM_BATCH=$gsb
    for g in 16 64; do run --envs $n --steps 1280 --graph $g; done
  done
  unset TSIDB_GRAPH_SIM_BATCH
done
run --envs 4096 --steps 1280
TSIDB_GRAPH_SIM_BATCH=1 run --envs 4096 --steps 1280 --graph 64
python3 -m pytest tests -m gpu -x -q -k "graph or pipelined" 2>&1 | tail -3
