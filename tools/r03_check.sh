#!/bin/bash
# gpu tests + driver-window bench + stamp profile (touch-down window): quick check of a kernel change
set -e
O=$GRAFT_REPO_ROOT/gpurun_out
TAG=${1:-r03b}
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q > $O/${TAG}_gputests.log 2>&1 || { tail -40 $O/${TAG}_gputests.log; exit 1; }
tail -3 $O/${TAG}_gputests.log
python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-secondary > $O/${TAG}_bdrv.json 2> $O/${TAG}_bench.err
python3 bench.py --steps 2000 --cpu-seconds 0 --no-secondary > $O/${TAG}_b2000.json 2>> $O/${TAG}_bench.err
python3 bench.py --envs 512 --steps 800 --cpu-seconds 0 --no-secondary > $O/${TAG}_b512.json 2>> $O/${TAG}_bench.err
for spec in "4096 620" "512 900"; do
  set -- $spec
  TSIDB_LIB_PATH=tools/_diag/libtsidb_stamps.so python3 tools/stamp_profile.py f64 $1 walk $2 > $O/${TAG}_stamps_$1_$2.txt 2>&1
done
python3 tools/dbg_f32_step.py > $O/${TAG}_f32_step.txt 2>&1 || true
python3 - <<PY
import json
for f in ("bdrv", "b2000", "b512"):
    d = json.load(open("$O/${TAG}_%s.json" % f)); r = d["roofline"]
    print(f, round(d["value"] / 1e6, 3), "M", round(d["ms_per_step"], 4), "ms tick", round(r["k_tick_ms"], 4), "sim", round(r["k_sim_ms"], 4), "newton", d["config"]["last_step_stats"]["newton_iters_mean"])
PY
grep -A12 "k_sim" $O/${TAG}_stamps_4096_620.txt
