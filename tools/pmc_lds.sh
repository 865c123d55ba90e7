#!/bin/bash
# one more PMC pass: where the waits are - LDS bank conflicts, LDS / scalar / vector-memory instruction and wait cycles
# usage (on the GPU box): bash tools/pmc_lds.sh <outdir-under-gpurun_out>
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_lds}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail > $OUT.avail.txt 2>&1
ARGS="$GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --cpu-seconds 0 --no-overlap --no-secondary"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $OUT/p6 -- python3 $ARGS > $OUT.p6.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/p7 -- python3 $ARGS > $OUT.p7.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INSTS_SENDMSG --output-format csv -d $OUT/p8 -- python3 $ARGS > $OUT.p8.log 2>&1
find $OUT -name "*counter_collection.csv" | head
tail -3 $OUT.p6.log $OUT.p7.log $OUT.p8.log
