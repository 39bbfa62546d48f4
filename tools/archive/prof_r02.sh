#!/bin/bash
# Round-2 profile set (run on the GPU box from the repo root): tools/prof_r02.sh
#   1. kernel trace + stats of the headline steps (bench.py --headline-only)    -> gpurun_out/r02/bench_kernel_trace.txt
#   2. counter passes over tools/pmc_workload.py 4 4096, each its own run (--pmc only together with --kernel-trace):
#      FETCH_SIZE | WRITE_SIZE + L2 hits | SQ busy / wait / MFMA | instruction mix   -> gpurun_out/r02/pmc_*.txt, traffic.json
REPO=$(pwd); OUT=$REPO/gpurun_out/r02; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/r02_trace -o t -- python $REPO/bench.py --headline-only --steps 20 --warmup 5 > $OUT/bench_headline.json 2> $OUT/trace.log
db=$(find /tmp/r02_trace -name "*.db" | head -1)
[ -n "$db" ] && python $REPO/tools/rocprof_summary.py $db > $OUT/bench_kernel_trace.txt 2>&1
rm -rf /tmp/r02_trace
declare -A SETS
SETS[fetch]="FETCH_SIZE"
SETS[write]="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
SETS[sq]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
SETS[mix]="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
for name in fetch write sq mix; do
  timeout 400 rocprofv3 --pmc ${SETS[$name]} --kernel-trace -d /tmp/r02_$name -o p -- python $REPO/tools/pmc_workload.py 4 4096 > $OUT/pmc_$name.log 2>&1
  db=$(find /tmp/r02_$name -name "*.db" | head -1)
  [ -n "$db" ] && python $REPO/tools/pmc_multi.py $db h16_ > $OUT/pmc_$name.txt 2>&1
  [ -n "$db" ] && cp $db /tmp/r02_$name.db
  rm -rf /tmp/r02_$name
done
cd $REPO
[ -f /tmp/r02_fetch.db ] && [ -f /tmp/r02_write.db ] && python tools/pmc_to_traffic.py /tmp/r02_fetch.db /tmp/r02_write.db 4096 1000000 768 r02 > $OUT/traffic.log 2>&1 && cp profiles/traffic.json $OUT/traffic.json
rm -f /tmp/r02_*.db
