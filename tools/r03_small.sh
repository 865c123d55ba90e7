#!/bin/bash
# small-batch sweep: env counts x sim batch sizes
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q > $O/r03t_gputests.log 2>&1 || { tail -30 $O/r03t_gputests.log; exit 1; }
tail -2 $O/r03t_gputests.log
for n in 512 1024 2048 4096; do
  for b in 1 2 4 8; do
    python3 bench.py --envs $n --steps 1200 --cpu-seconds 0 --no-secondary --sim-batch $b > $O/r03t_${n}_${b}.json 2>/dev/null
    python3 - <<PY
import json
d = json.load(open("$O/r03t_${n}_${b}.json")); r = d["roofline"]
print("envs $n batch $b:", round(d["value"] / 1e6, 3), "M", round(d["ms_per_step"], 4), "ms tick", round(r["k_tick_ms"], 4), "sim", round(r["k_sim_ms"], 4))
PY
  done
done
