#!/bin/bash
# stamp profile of k_tick with tight torque bounds (42 % of the envs in the dual active-set loop) and in the plain walk
out=gpurun_out/${1:-r04_stamps_tick}.txt; : > $out
echo "== tight torque bounds (tau_max_scaling 0.12, dephase 0.5), tick 850" >> $out
TSIDB_TAU_MAX_SCALING=0.12 TSIDB_DEPHASE=0.5 TSIDB_LIB_PATH=${2:-tools/_diag/libtsidb_stamps.so} python tools/stamp_profile.py f64 4096 walk 850 2>&1 | grep -B30 "^k_sim" | grep -v "^k_sim" >> $out
echo "== plain walk, tick 620" >> $out
TSIDB_LIB_PATH=${2:-tools/_diag/libtsidb_stamps.so} python tools/stamp_profile.py f64 4096 walk 620 2>&1 | grep -B30 "^k_sim" | grep -v "^k_sim" >> $out
cat $out
