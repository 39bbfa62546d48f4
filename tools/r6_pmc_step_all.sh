#!/bin/bash
# Round 6: FETCH_SIZE / WRITE_SIZE of EVERY launch of the headline step (single-counter passes): what the non-scan launches move
REPO=$(pwd); OUT=$REPO/gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/r6s_$ctr -o p -- python $REPO/tools/pmc_workload.py 4 4096 > $OUT/step_all_$ctr.log 2>&1
  db=$(find /tmp/r6s_$ctr -name "*.db" | head -1)
  [ -n "$db" ] && python $REPO/tools/pmc_multi.py $db > $OUT/step_all_$ctr.txt 2>&1
  rm -rf /tmp/r6s_$ctr
  grep -E "rerank|coarse|cand_select|sample|prep_queries|preprune|hist|scatter|plan_scan|subset|h16_scan" $OUT/step_all_$ctr.txt | cut -c1-150
done
