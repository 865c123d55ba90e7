#!/bin/bash
# stamp profiles (diagnostic build) of the sim kernel: one env per wavefront against two, touch-down window and mid-swing
out=gpurun_out/r04_stamps_pack.txt; : > $out
for tick in 620 900; do for p in 0 1; do
  echo "== TSIDB_SIM_PACK=$p walk 4096 envs, tick $tick" >> $out
  TSIDB_SIM_PACK=$p TSIDB_LIB_PATH=tools/_diag/libtsidb_stamps.so python tools/stamp_profile.py f64 4096 walk $tick 2>&1 | grep -A12 "^k_sim" >> $out
done; done
cat $out
