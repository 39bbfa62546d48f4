#!/bin/bash
# Round-5 profile set (run on the GPU box from the repo root; every step time-boxed): kernel traces of the headline step, the BM25 batches
# (64 / 1024 queries, with the launch timeline), the few-query FLAT shadow path, the exhaustive FLAT pass; SQ counter passes of the BM25
# record scorer (kernel trace only beside --pmc).  Outputs under gpurun_out/r05/ -- copy what is to be kept into profiles/.
REPO=$(pwd); OUT=$REPO/gpurun_out/r05; mkdir -p $OUT; export TMPDIR=/tmp
trace() { # name, command...
  local name=$1; shift
  TIMELINE=${TL:-0} timeout 240 tools/prof_cmd.sh gpurun_out/r05/${name}_kernel_trace.txt "$@"
}
trace bench python $REPO/bench.py --headline-only --no-concurrent --steps 20 --warmup 5
TL=16 trace bm25_64 python $REPO/tools/r5_bm25_ab.py --batches 64 --variants 0
TL=16 trace bm25_1024 python $REPO/tools/r5_bm25_ab.py --batches 1024 --variants 0
TL=14 trace flat_latency python $REPO/tools/r5_flat_lat.py --few-only
trace flat python $REPO/bench.py --only iid --no-cpu-baseline --steps 5 --warmup 2
cd /tmp
for pass in "sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
            "sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  set -- $pass; name=$1; shift
  STEPS=6 timeout 150 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/r05p_$name -o p -- python $REPO/tools/r5_bm25_ab.py --batches 1024 --variants 0 > $OUT/pmc_bm25_$name.log 2>&1
  db=$(find /tmp/r05p_$name -name "*.db" | head -1)
  [ -n "$db" ] && python $REPO/tools/pmc_multi.py $db bm25 > $OUT/pmc_bm25_$name.txt 2>&1
  rm -rf /tmp/r05p_$name
done
ls -la $OUT | head -40
