#!/bin/bash
# Round 5: SQ counter passes of the BM25 record scorer at a batch of 1024 (kernel trace only beside --pmc)
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
pmc() { # name, counters
  local name=$1 counters=$2
  timeout 600 rocprofv3 --pmc $counters --kernel-trace -d /tmp/r5p_$name -o p -- python $REPO/tools/r5_bm25_ab.py --batches ${B:-1024} --variants ${V:-0} > $OUT/r5_pmc_bm25_$name.log 2>&1
  local db=$(find /tmp/r5p_$name -name "*.db" | head -1)
  [ -n "$db" ] && python $REPO/tools/pmc_multi.py $db "bm25${K:-l}_kernel<1" > $OUT/r5_pmc_bm25_$name.txt 2>&1
  rm -rf /tmp/r5p_$name
  cat $OUT/r5_pmc_bm25_$name.txt
}
pmc sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
pmc sq2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
