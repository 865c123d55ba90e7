#!/bin/bash
# A/B on one box: sim kernel with one env per wavefront (TSIDB_SIM_PACK=0) against two envs per wavefront (=1), the driver's
# 20-step window, the 5000-tick phase average, tick and sim back to back.  Output: gpurun_out/r04_pack_ab.txt
out=gpurun_out/r04_pack_ab.txt; : > $out
for rep in 1 2; do
for p in 0 1; do
  echo "== TSIDB_SIM_PACK=$p driver window (rep $rep)" >> $out
  TSIDB_SIM_PACK=$p python bench.py --steps 20 --warmup 5 --no-secondary --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('value','ms_per_step','k_tick_ms','k_sim_ms')})" >> $out
done; done
for p in 0 1; do
  echo "== TSIDB_SIM_PACK=$p 5000 ticks" >> $out
  TSIDB_SIM_PACK=$p python bench.py --steps 5000 --warmup 5 --no-secondary --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('value','ms_per_step','k_tick_ms','k_sim_ms')})" >> $out
  echo "== TSIDB_SIM_PACK=$p 5000 ticks --no-overlap" >> $out
  TSIDB_SIM_PACK=$p python bench.py --steps 5000 --warmup 5 --no-secondary --cpu-seconds 0 --no-overlap 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('value','ms_per_step','k_tick_ms','k_sim_ms')})" >> $out
done
cat $out
