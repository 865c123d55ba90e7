#!/bin/bash
# throughput against resident workgroups per CU: unused dynamic LDS lowers the residency from 8 (pad 0) to 7, 6, 5, 4
# (160 KB LDS per CU; k_tick / k_sim use 20.0 KB static).  4096 walkers, pipelined and back to back.
cd $GRAFT_REPO_ROOT
run() { python3 bench.py "$@" --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('pad', os.environ.get('TSIDB_LDS_PAD'), '$*', round(d['value']/1e6,3), 'M', round(d['ms_per_step'],4), 'tick', round(r['k_tick_ms'],4), 'sim', round(r['k_sim_ms'],4))"; }
for pad in 0 2900 6800 12200 20400; do
  export TSIDB_LDS_PAD=$pad
  run --steps 1000
  run --steps 1000 --no-overlap
  run --steps 1000 --envs 2048 --no-overlap
done
