#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
for spec in "512 4 0 8" "512 4 16 8" "512 4 16 100000" "512 1 0 100000" "512 8 16 100000" "1024 8 16 100000" "1024 4 0 100000"; do
  set -- $spec
  TSIDB_RING_SLOTS=$3 python3 bench.py --envs $1 --steps 1200 --cpu-seconds 0 --no-secondary --sim-batch $2 --event-every $4 > $O/r03w.json 2>/dev/null
  python3 - <<PY
import json
d = json.load(open("$O/r03w.json")); r = d["roofline"]
print("envs $1 batch $2 ring $3 events-every $4:", round(d["value"] / 1e6, 3), "M", round(d["ms_per_step"], 4), "ms")
PY
done
