#!/bin/bash
# PMC passes over bench.py (counters in their own runs, no tracing domains besides the kernel trace).
# usage (on the GPU box): bash tools/pmc_profile.sh <outdir-under-gpurun_out>
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc}
export TMPDIR=/tmp
cd /tmp
ARGS="$GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --cpu-seconds 0 --no-overlap --no-secondary"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -- python3 $ARGS > $OUT.p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/p2 -- python3 $ARGS > $OUT.p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p3 -- python3 $ARGS > $OUT.p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/p4 -- python3 $ARGS > $OUT.p4.log 2>&1
# instruction classes: what the VALU stream is made of - float64 add / mul / fma / transcendental, 32- and 64-bit
# integer (v_readlane, v_cndmask, DPP moves, compares land here), conversions
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d $OUT/p5 -- python3 $ARGS > $OUT.p5.log 2>&1
find $OUT -name "*counter_collection.csv" | head
