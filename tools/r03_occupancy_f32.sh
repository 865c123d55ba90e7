#!/bin/bash
# what a third resident wavefront per SIMD is worth for this code shape: the float32 k_tick (150 VGPRs, 10 KB LDS: 12 workgroups
# per CU) with unused LDS taking it down to 8 and 4 per CU.  6144 envs = 24 per CU: 2 / 3 / 6 rounds.  Back to back.
cd $GRAFT_REPO_ROOT
run() { python3 bench.py "$@" --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('pad', os.environ.get('TSIDB_LDS_PAD'), '$*', round(d['value']/1e6,3), 'M', round(d['ms_per_step'],4), 'tick', round(r['k_tick_ms'],4), 'sim', round(r['k_sim_ms'],4))"; }
for pad in 0 3400 6000 10200 16000 30000; do
  export TSIDB_LDS_PAD=$pad
  run --dtype f32 --envs 6144 --steps 600 --no-overlap
done
