#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
for spec in "512 1 1" "512 4 1" "512 8 1" "512 1 2" "512 4 2" "256 4 1" "256 1 2" "256 4 2"; do
  set -- $spec
  TSIDB_SIM_WAVES=$3 python3 bench.py --envs $1 --steps 1200 --cpu-seconds 0 --no-secondary --sim-batch $2 > $O/r03w.json 2>/dev/null
  python3 - <<PY
import json
d = json.load(open("$O/r03w.json")); r = d["roofline"]
print("envs $1 batch $2 waves $3:", round(d["value"] / 1e6, 3), "M", round(d["ms_per_step"], 4), "ms tick", round(r["k_tick_ms"], 4), "sim", round(r["k_sim_ms"], 4))
PY
done
