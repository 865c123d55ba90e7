#!/bin/bash
# A/B of two builds on the same box: driver window x5, 2000 steps, 512 envs, alternating
# usage: bash tools/r03_ab.sh <libA> <libB>
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5; do
  for lib in "$@"; do
    TSIDB_LIB_PATH=$lib python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('drv  $lib', round(d['value']/1e6,3), round(r['k_tick_ms'],4), round(r['k_sim_ms'],4))"
  done
done
for lib in "$@" "$@"; do
  TSIDB_LIB_PATH=$lib python3 bench.py --steps 2000 --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('2000 $lib', round(d['value']/1e6,3), round(r['k_tick_ms'],4), round(r['k_sim_ms'],4))"
  TSIDB_LIB_PATH=$lib python3 bench.py --envs 512 --steps 800 --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('512  $lib', round(d['value']/1e6,3), round(r['k_tick_ms'],4), round(r['k_sim_ms'],4))"
done
