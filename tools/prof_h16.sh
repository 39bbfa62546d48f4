#!/bin/bash
# rocprofv3 passes over tools/pmc_workload.py (run on the GPU box from the repo root):  tools/prof_h16.sh <outdir> <batch> [passes]
# kernel trace + stats, then counter passes (each its own run; --pmc only together with --kernel-trace).
OUT=${1:-gpurun_out/prof}; B=${2:-4096}; PASSES=${3:-"1 2 3 4"}; REPO=$(pwd)
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/$OUT/trace -o t -- python $REPO/tools/pmc_workload.py 6 $B > $REPO/$OUT/trace.log 2>&1
db=$(find $REPO/$OUT/trace -name "*.db" | head -1)
[ -n "$db" ] && python $REPO/tools/rocprof_summary.py $db > $REPO/$OUT/trace_summary.txt 2>&1
SETS=("" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
      "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
      "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_WR" \
      "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum")
for i in $PASSES; do
  timeout 300 rocprofv3 --pmc ${SETS[$i]} --kernel-trace -d $REPO/$OUT/pmc$i -o p -- python $REPO/tools/pmc_workload.py 3 $B > $REPO/$OUT/pmc$i.log 2>&1
  db=$(find $REPO/$OUT/pmc$i -name "*.db" | head -1)
  [ -n "$db" ] && python $REPO/tools/pmc_multi.py $db h16_scan > $REPO/$OUT/pmc$i.txt 2>&1
  rm -rf $REPO/$OUT/pmc$i
done
rm -rf $REPO/$OUT/trace
cd $REPO
