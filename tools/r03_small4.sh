#!/bin/bash
# final kernels: wavefronts per env in k_sim x batch size, 8 sim steps per launch (and 1 for the two-wavefront variant)
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
for spec in "128 8 1" "128 8 2" "256 8 1" "256 8 2" "384 8 1" "384 8 2" "512 8 1" "512 8 2" "512 1 2" "768 8 1" "768 8 2" "1024 8 1" "1024 8 2" "1024 1 2"; do
  set -- $spec
  TSIDB_SIM_WAVES=$3 python3 bench.py --envs $1 --steps 1200 --cpu-seconds 0 --no-secondary --sim-batch $2 > $O/r03w.json 2>/dev/null
  python3 - <<PY
import json
d = json.load(open("$O/r03w.json")); r = d["roofline"]
print("envs $1 batch $2 waves $3:", round(d["value"] / 1e6, 3), "M", round(d["ms_per_step"], 4), "ms tick", round(r["k_tick_ms"], 4), "sim", round(r["k_sim_ms"], 4))
PY
done
