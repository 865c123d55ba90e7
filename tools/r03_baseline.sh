#!/bin/bash
# round-3 starting point: gpu tests, stamp profiles at 512 / 4096 envs, small-batch bench lines
set -e
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q > $O/r03a_gputests.log 2>&1 || { tail -30 $O/r03a_gputests.log; exit 1; }
tail -3 $O/r03a_gputests.log
for spec in "512 620" "512 900" "4096 620" "4096 900"; do
  set -- $spec
  TSIDB_LIB_PATH=tools/_diag/libtsidb_stamps.so python3 tools/stamp_profile.py f64 $1 walk $2 > $O/r03a_stamps_$1_$2.txt 2>&1
done
python3 bench.py --envs 512 --steps 800 --cpu-seconds 0 --no-secondary > $O/r03a_b512.json 2> $O/r03a_b512.err
python3 bench.py --envs 1024 --steps 800 --cpu-seconds 0 --no-secondary > $O/r03a_b1024.json 2>> $O/r03a_b512.err
python3 bench.py --envs 2048 --steps 800 --cpu-seconds 0 --no-secondary > $O/r03a_b2048.json 2>> $O/r03a_b512.err
python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-secondary > $O/r03a_bdrv.json 2>> $O/r03a_b512.err
cat $O/r03a_stamps_512_900.txt
