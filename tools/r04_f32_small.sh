#!/bin/bash
# float32 sim kernel at three (product) against two wavefronts per SIMD over the batch sizes, pipelined and closed loop
out=gpurun_out/r04_f32_wpe3_sizes.txt; : > $out
pr='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(round(d["value"]/1e6,3), r["k_tick_ms"] and round(r["k_tick_ms"],4), r["k_sim_ms"] and round(r["k_sim_ms"],4))'
for n in 512 1024 2048 4096 16384; do for lib in tsid_control_amd/libtsidb.so tools/_diag/lib_f32wpe2.so; do
  echo "f32 $n envs 1500 ticks $lib $(TSIDB_LIB_PATH=$lib python3 bench.py --dtype f32 --envs $n --steps 1500 --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "$pr")" >> $out
done; done
for lib in tsid_control_amd/libtsidb.so tools/_diag/lib_f32wpe2.so; do
  echo "f32 4096 closed loop $lib $(TSIDB_LIB_PATH=$lib python3 bench.py --dtype f32 --closed-loop --steps 1500 --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "$pr")" >> $out
  echo "f32 4096 stand $lib $(TSIDB_LIB_PATH=$lib python3 bench.py --dtype f32 --workload stand --steps 1500 --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "$pr")" >> $out
done
cat $out
