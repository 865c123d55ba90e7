#!/bin/bash
# A/B of several builds on one box, alternating: driver window x REPS, then 2000 ticks pipelined and back to back
# usage: bash tools/r04_ab.sh <out-name> <lib> [<lib> ...]      (lib = path, or "product")
out=gpurun_out/$1.txt; shift; : > $out
pr='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(round(d["value"]/1e6,3), round(r["k_tick_ms"],4), round(r["k_sim_ms"],4))'
for rep in 1 2 3; do for lib in "$@"; do
  L=$lib; [ "$lib" = product ] && L=tsid_control_amd/libtsidb.so
  echo "drv   $lib $(TSIDB_LIB_PATH=$L python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "$pr")" >> $out
done; done
for lib in "$@" "$@"; do
  L=$lib; [ "$lib" = product ] && L=tsid_control_amd/libtsidb.so
  echo "2000  $lib $(TSIDB_LIB_PATH=$L python3 bench.py --steps 2000 --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "$pr")" >> $out
  echo "2000s $lib $(TSIDB_LIB_PATH=$L python3 bench.py --steps 2000 --cpu-seconds 0 --no-secondary --no-overlap 2>/dev/null | python3 -c "$pr")" >> $out
done
sort -k1,2 -s $out
