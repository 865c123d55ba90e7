#!/bin/bash
# float32: kernels register-allocated for three wavefronts per SIMD (-DTSIDB_WPE=3; 10 KB of LDS per env allows 12-15 per CU)
out=gpurun_out/r04_f32_wpe3.txt; : > $out
pr='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(round(d["value"]/1e6,3), round(r["k_tick_ms"],4), round(r["k_sim_ms"],4))'
for rep in 1 2; do for lib in tsid_control_amd/libtsidb.so tools/_diag/lib_wpe3.so; do
  echo "f32 2000        $lib $(TSIDB_LIB_PATH=$lib python3 bench.py --dtype f32 --steps 2000 --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "$pr")" >> $out
  echo "f32 2000 serial $lib $(TSIDB_LIB_PATH=$lib python3 bench.py --dtype f32 --steps 2000 --cpu-seconds 0 --no-secondary --no-overlap 2>/dev/null | python3 -c "$pr")" >> $out
  echo "f32 drv         $lib $(TSIDB_LIB_PATH=$lib python3 bench.py --dtype f32 --steps 20 --warmup 5 --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "$pr")" >> $out
done; done
sort -s -k1,3 $out
