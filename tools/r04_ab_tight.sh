#!/bin/bash
# builds alternating: the walking batch with torque bounds tight enough that 38 % of the envs iterate, and the plain walk
out=gpurun_out/$1.txt; shift; : > $out
pr='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(round(d["value"]/1e6,3), round(r["k_tick_ms"],4), round(r["k_sim_ms"],4))'
for rep in 1 2 3; do for lib in "$@"; do
  L=$lib; [ "$lib" = product ] && L=tsid_control_amd/libtsidb.so
  echo "tight $lib $(TSIDB_LIB_PATH=$L python3 bench.py --steps 400 --preroll 800 --tau-max-scaling 0.12 --dephase 0.5 --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "$pr")" >> $out
  echo "tightS $lib $(TSIDB_LIB_PATH=$L python3 bench.py --steps 400 --preroll 800 --tau-max-scaling 0.12 --dephase 0.5 --cpu-seconds 0 --no-secondary --no-overlap 2>/dev/null | python3 -c "$pr")" >> $out
  echo "walkS $lib $(TSIDB_LIB_PATH=$L python3 bench.py --steps 1000 --cpu-seconds 0 --no-secondary --no-overlap 2>/dev/null | python3 -c "$pr")" >> $out
done; done
sort -s -k1,2 $out
