#!/bin/bash
# Round 6: counter passes of the exhaustive FLAT shadow pass (h16_flat_kernel, 4096 queries over 1M x 768); kernel trace only beside --pmc
REPO=$(pwd); OUT=$REPO/gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
K=${K:-h16_flat_kernel}
pmc() { # name, counters
  local name=$1 counters=$2
  timeout 600 rocprofv3 --pmc $counters --kernel-trace -d /tmp/r6p_$name -o p -- python $REPO/tools/flat_batch.py 3 4096 ${OPTS} > $OUT/pmc_flat_$name.log 2>&1
  local db=$(find /tmp/r6p_$name -name "*.db" | head -1)
  [ -n "$db" ] && python $REPO/tools/pmc_multi.py $db "$K" > $OUT/pmc_flat_$name.txt 2>&1
  rm -rf /tmp/r6p_$name
  cat $OUT/pmc_flat_$name.txt
}
for p in "$@"; do
case $p in
 sq1) pmc sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT";;
 sq2) pmc sq2 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD";;
 tcc) pmc tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum";;
 fetch) pmc fetch "FETCH_SIZE";;
 tcp) pmc tcp "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum";;
esac
done
