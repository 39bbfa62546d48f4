#!/bin/bash
# Round 6: instruction-cache counters of the headline step's launches (is a short kernel's fixed cost its code fetch?)
REPO=$(pwd); OUT=$REPO/gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -o -E "(SQC_ICACHE[A-Z_]*|SQ_IFETCH[A-Z_]*|SQC_TC_INST[A-Z_]*|SQ_WAIT_INST_ANY|SQ_INST_LEVEL[A-Z_]*)" | sort -u | tr '\n' ' ' > $OUT/icache_counters_available.txt
cat $OUT/icache_counters_available.txt; echo
pmc() {
  local name=$1 counters=$2
  timeout 600 rocprofv3 --pmc $counters --kernel-trace -d /tmp/r6i_$name -o p -- python $REPO/bench.py --headline-only --steps 4 --warmup 2 --no-concurrent > $OUT/pmc_icache_$name.log 2>&1
  local db=$(find /tmp/r6i_$name -name "*.db" | head -1)
  [ -n "$db" ] && python $REPO/tools/pmc_multi.py $db > $OUT/pmc_icache_$name.txt 2>&1
  rm -rf /tmp/r6i_$name
  grep -E "coarse_gemm|coarse_tail|ivf_rerank|cand_select|sample_thr|h16_sample|prep_queries|preprune|hist_lds|merge_subset" $OUT/pmc_icache_$name.txt | cut -c1-150
}
pmc a "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH"
pmc b "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
