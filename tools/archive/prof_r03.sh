#!/bin/bash
# Round-3 profile set (run on the GPU box from the repo root): tools/prof_r03.sh [headline|c3|c4|c5|all]
#   headline: kernel trace + stats of the headline steps; FETCH_SIZE / WRITE_SIZE / SQ passes of the list scan  -> traffic.json
#   c3 / c4 / c5: kernel trace + stats of that bench leg (bench.py --only ...), FETCH_SIZE pass of its h16 kernels
# Counter passes never combine --pmc with sys / hip / hsa trace domains (kernel trace only) and are time-boxed.
WHAT=${1:-all}
REPO=$(pwd); OUT=$REPO/gpurun_out/r03; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
trace() { # name, bench args...
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/r03_$name -o t -- python $REPO/bench.py "$@" > $OUT/${name}_bench.json 2> $OUT/${name}_trace.log
  local db=$(find /tmp/r03_$name -name "*.db" | head -1)
  [ -n "$db" ] && python $REPO/tools/rocprof_summary.py $db > $OUT/${name}_kernel_trace.txt 2>&1
  rm -rf /tmp/r03_$name
}
pmc() { # name, counters, filter, command...
  local name=$1 counters=$2 filt=$3; shift 3
  timeout 900 rocprofv3 --pmc $counters --kernel-trace -d /tmp/r03p_$name -o p -- "$@" > $OUT/pmc_$name.log 2>&1
  local db=$(find /tmp/r03p_$name -name "*.db" | head -1)
  [ -n "$db" ] && python $REPO/tools/pmc_multi.py $db $filt > $OUT/pmc_$name.txt 2>&1
  [ -n "$db" ] && cp $db /tmp/r03_$name.db
  rm -rf /tmp/r03p_$name
}
if [ "$WHAT" = headline ] || [ "$WHAT" = all ]; then
  trace headline --headline-only --steps 20 --warmup 5
  pmc fetch "FETCH_SIZE" h16_ python $REPO/tools/pmc_workload.py 4 4096
  pmc write "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" h16_ python $REPO/tools/pmc_workload.py 4 4096
  pmc sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" h16_ python $REPO/tools/pmc_workload.py 4 4096
  ( cd $REPO && [ -f /tmp/r03_fetch.db ] && [ -f /tmp/r03_write.db ] && python tools/pmc_to_traffic.py /tmp/r03_fetch.db /tmp/r03_write.db 4096 1000000 768 r03 > $OUT/traffic.log 2>&1 && cp profiles/traffic.json $OUT/traffic.json )
fi
if [ "$WHAT" = c3 ] || [ "$WHAT" = all ]; then
  trace c3 --only c3 --steps 5 --warmup 2
  pmc c3_fetch "FETCH_SIZE" h16_ python $REPO/bench.py --only c3 --steps 5 --warmup 2
fi
if [ "$WHAT" = c4 ] || [ "$WHAT" = all ]; then
  trace c4 --only c4 --steps 5 --warmup 2
  pmc c4_fetch "FETCH_SIZE" h16_ python $REPO/bench.py --only c4 --steps 5 --warmup 2
fi
if [ "$WHAT" = c5 ] || [ "$WHAT" = all ]; then
  trace c5 --only c3,c5 --steps 5 --warmup 2
  pmc c5_fetch "FETCH_SIZE" bm25 python $REPO/bench.py --only c3,c5 --steps 5 --warmup 2
fi
rm -f /tmp/r03_*.db
ls -la $OUT
