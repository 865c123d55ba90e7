#!/bin/bash
out=gpurun_out/r04_sim_waves_512.txt; : > $out
pr='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(round(d["value"]/1e6,3), round(r["k_tick_ms"],4), round(r["k_sim_ms"],4))'
for rep in 1 2 3; do for n in 512 448; do for w in 1 2; do
  echo "envs $n sim waves $w: $(TSIDB_SIM_WAVES=$w python3 bench.py --envs $n --steps 1500 --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "$pr")" >> $out
done; done; done
sort -s -k2,5 $out
