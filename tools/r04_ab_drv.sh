#!/bin/bash
# the driver's 20-step window, several builds alternating, REPS times each; plus the device-plan window and 512 / 2048 envs
out=gpurun_out/$1.txt; shift; : > $out
pr='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(round(d["value"]/1e6,3), round(r["k_tick_ms"],4), round(r["k_sim_ms"],4))'
for rep in 1 2 3 4 5 6; do for lib in "$@"; do
  L=$lib; [ "$lib" = product ] && L=tsid_control_amd/libtsidb.so
  echo "drv   $lib $(TSIDB_LIB_PATH=$L python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "$pr")" >> $out
done; done
for lib in "$@"; do
  L=$lib; [ "$lib" = product ] && L=tsid_control_amd/libtsidb.so
  for n in 512 2048; do
  echo "n$n $lib $(TSIDB_LIB_PATH=$L python3 bench.py --envs $n --steps 800 --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "$pr")" >> $out
  done
  echo "65536 $lib $(TSIDB_LIB_PATH=$L python3 bench.py --envs 65536 --randomize --steps 100 --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "$pr")" >> $out
done
sort -s -k1,2 $out
