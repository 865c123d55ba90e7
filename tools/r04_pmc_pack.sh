#!/bin/bash
# instruction counters of the packed sim kernel (two envs per wavefront) beside the one-env kernel: passes p1 and p5 of pmc_profile.sh
set -e
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_pmc_pack
rm -rf $O; mkdir -p $O
cd /tmp
ARGS="$GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --cpu-seconds 0 --no-overlap --no-secondary"
export TSIDB_SIM_PACK=1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p1 -- python3 $ARGS > $O.p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d $O/p5 -- python3 $ARGS > $O.p5.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $O/p6 -- python3 $ARGS > $O.p6.log 2>&1 || true
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $O --last 100 > gpurun_out/r04_pmc_pack_last100.txt
cat gpurun_out/r04_pmc_pack_last100.txt
