#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
STEPS=6 timeout 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d /tmp/r5b_a -o p -- python $REPO/tools/r5_bm25_ab.py --batches 1024 --variants 0 > $OUT/r5_pmc_bounds_a.log 2>&1
db=$(find /tmp/r5b_a -name "*.db" | head -1)
[ -n "$db" ] && python $REPO/tools/pmc_multi.py $db bounds8 | cut -c60-200
