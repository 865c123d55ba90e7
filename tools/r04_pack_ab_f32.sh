#!/bin/bash
# float32: the packed sim kernel at TWO wavefronts per SIMD (10 KB of LDS per env: 16 envs per CU) against one env per wavefront
# (8 per CU): what the float64 kernel would need an LDS diet for.  tick and sim back to back, 2000 ticks.
out=gpurun_out/r04_pack_ab_f32.txt; : > $out
for p in 0 1; do
  echo "== f32 TSIDB_SIM_PACK=$p 2000 ticks --no-overlap" >> $out
  TSIDB_SIM_PACK=$p python bench.py --dtype f32 --steps 2000 --warmup 5 --no-secondary --cpu-seconds 0 --no-overlap 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('value','ms_per_step')}, d.get('roofline',{}).get('kernel'), d.get('kernels'))" >> $out
  echo "== f32 TSIDB_SIM_PACK=$p 2000 ticks pipelined" >> $out
  TSIDB_SIM_PACK=$p python bench.py --dtype f32 --steps 2000 --warmup 5 --no-secondary --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('value','ms_per_step')})" >> $out
done
cat $out
