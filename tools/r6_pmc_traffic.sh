#!/bin/bash
# Round 6: HBM traffic of the kernels the verdict asked about -- FETCH_SIZE and WRITE_SIZE as SEPARATE single-counter passes (kernel trace only
# beside --pmc; the 4-counter pass of round 5 hung).  Summaries -> gpurun_out/r06/traffic_<what>_<counter>.txt
REPO=$(pwd); OUT=$REPO/gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
pass() { # what, counter, kernel substring, command...
  local what=$1 ctr=$2 kern=$3; shift 3
  timeout 900 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/r6t_${what}_$ctr -o p -- "$@" > $OUT/traffic_${what}_$ctr.log 2>&1
  local db=$(find /tmp/r6t_${what}_$ctr -name "*.db" | head -1)
  [ -n "$db" ] && python $REPO/tools/pmc_multi.py $db "$kern" > $OUT/traffic_${what}_$ctr.txt 2>&1
  rm -rf /tmp/r6t_${what}_$ctr
  grep -v "^#" $OUT/traffic_${what}_$ctr.txt | head -12
  grep "rows(model" $OUT/traffic_${what}_$ctr.log | tail -1 | cut -c1-300
}
for w in "$@"; do
case $w in
 headline) for c in FETCH_SIZE WRITE_SIZE; do pass headline $c h16_s python $REPO/tools/pmc_workload.py 4 4096; done;;
 mid)      for c in FETCH_SIZE WRITE_SIZE; do PMC_DATA=mid pass mid $c h16_s python $REPO/tools/pmc_workload.py 4 4096; done;;
 iid)      for c in FETCH_SIZE WRITE_SIZE; do PMC_DATA=iid pass iid $c h16_s python $REPO/tools/pmc_workload.py 4 4096; done;;
 flatfew)  for c in FETCH_SIZE WRITE_SIZE; do pass flatfew $c h16_flat_kernel python $REPO/tools/flat_few_latency.py --few-only; done;;
 flat)     for c in FETCH_SIZE WRITE_SIZE; do pass flat $c h16_flat_kernel python $REPO/tools/flat_batch.py 3 4096; done;;
 bm25)     for c in FETCH_SIZE WRITE_SIZE; do pass bm25 $c bm25 python $REPO/tools/bm25_ab.py --batches 64,1024 --variants 0; done;;
esac
done
