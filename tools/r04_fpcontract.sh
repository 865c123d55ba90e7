#!/bin/bash
# -ffp-contract=on (fusion only where the source writes a * b + c in one expression: the same in every instantiation of a
# kernel) against the default fast (the backend also fuses across statements, context dependent): GPU tests + bench A/B
out=gpurun_out/r04_fpcontract.txt; : > $out
for lib in "" tools/_diag/libtsidb_fpon.so; do
  for rep in 1 2; do
  echo "== lib=${lib:-product} driver window" >> $out
  TSIDB_LIB_PATH=$lib python bench.py --steps 20 --warmup 5 --no-secondary --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('value','ms_per_step')})" >> $out
  done
  echo "== lib=${lib:-product} 3000 ticks no-overlap" >> $out
  TSIDB_LIB_PATH=$lib python bench.py --steps 3000 --warmup 5 --no-secondary --cpu-seconds 0 --no-overlap 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('value','ms_per_step')})" >> $out
done
cat $out
TSIDB_LIB_PATH=tools/_diag/libtsidb_fpon.so timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_v0_robot.py > gpurun_out/r04_fpcontract_tests.log 2>&1; tail -15 gpurun_out/r04_fpcontract_tests.log
