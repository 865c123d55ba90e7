#!/bin/bash
# 512 / 384 / 256 walkers: CUs of the tick stream (the sim stream takes the rest) x wavefronts per env in the sim kernel
out=gpurun_out/r04_cu_share.txt; : > $out
pr='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(round(d["value"]/1e6,3), round(r["k_tick_ms"],4), round(r["k_sim_ms"],4))'
for n in 512 384 256; do for cu in 128 112 96 80 64; do for w in 1 2; do
  echo "envs $n tick CUs $cu sim waves $w: $(TSIDB_CU_TICK=$cu TSIDB_SIM_WAVES=$w python3 bench.py --envs $n --steps 800 --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "$pr")" >> $out
done; done; done
cat $out
