#!/bin/bash
# the pipelined step with two wavefronts per env in k_sim and 8 sim steps per launch, 512 / 1024 envs
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
for n in 512 1024; do for w in 1 2; do
  export TSIDB_SIM_WAVES=$w
  rocprofv3 --kernel-trace --output-format csv -d $O/r04_tr${n}_w$w -- python3 $GRAFT_REPO_ROOT/bench.py --envs $n --steps 400 --cpu-seconds 0 --no-secondary --sim-batch 8 > $O/r04_tr${n}_w$w.json 2> $O/r04_tr${n}_w$w.err
  echo "== $n envs, $w wavefront(s) per env in k_sim, 8 sim steps per launch" > $O/r04_tr${n}_w$w.txt
  python3 -c "import json,sys; d=json.loads(open('$O/r04_tr${n}_w$w.json').read().strip().splitlines()[-1]); print('bench:', d['value'], 'env-steps/s', d['ms_per_step'], 'ms/step')" >> $O/r04_tr${n}_w$w.txt
  python3 $GRAFT_REPO_ROOT/tools/trace_timeline.py $O/r04_tr${n}_w$w 45 >> $O/r04_tr${n}_w$w.txt
  rm -rf $O/r04_tr${n}_w$w
done; done
cat $O/r04_tr512_w2.txt $O/r04_tr1024_w2.txt
