#!/bin/bash
# stamp profiles of diagnostic build variants tools/_diag/libtsidb_stamps*.so (touch-down window, 4096 envs; mid-swing, 512 envs)
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
for lib in tools/_diag/libtsidb_stamps*.so; do
  n=$(basename $lib .so)
  for spec in "4096 620" "512 900"; do
    set -- $spec
    TSIDB_LIB_PATH=$lib python3 tools/stamp_profile.py f64 $1 walk $2 > $O/var_${n}_$1_$2.txt 2>&1
    echo "== $n $1 $2"; grep -A12 "k_sim" $O/var_${n}_$1_$2.txt | grep -v "^k_sim"
  done
done
