#!/bin/bash
# why is the 20-step driver window slower per step than the long run?  vary warm-up, length, overlap
cd $GRAFT_REPO_ROOT
run() { python3 bench.py "$@" --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$*', round(d['value']/1e6,3), 'M', round(d['ms_per_step'],4), 'tick', round(r['k_tick_ms'],4), 'sim', round(r['k_sim_ms'],4), 'newton', round(d['config']['last_step_stats']['newton_iters_mean'],2))"; }
for rep in 1 2 3; do
run --steps 20 --warmup 5
run --steps 20 --warmup 5 --event-every 7
run --steps 20 --warmup 5 --event-every 100
run --steps 100 --warmup 5
run --steps 100 --warmup 5 --event-every 1000
done
