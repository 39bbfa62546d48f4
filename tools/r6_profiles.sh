#!/bin/bash
# Round 6: the profiles the round commits -- kernel traces (+ timelines) of the headline step, the exhaustive FLAT pass, the BM25 batches
mkdir -p gpurun_out/r06
TIMELINE=36 TIMELINE_AFTER="h16_scan_kernel:4" tools/prof_cmd.sh gpurun_out/r06/r06_bench_kernel_trace.txt python $PWD/bench.py --headline-only --steps 6 --warmup 2 --no-concurrent
mv gpurun_out/r06/r06_bench_kernel_trace_timeline.txt gpurun_out/r06/r06_bench_step_kernel_trace_timeline.txt
tools/prof_cmd.sh gpurun_out/r06/r06_flat_kernel_trace.txt python $PWD/tools/flat_batch.py 5 4096
TIMELINE=24 tools/prof_cmd.sh gpurun_out/r06/r06_bm25_1024_kernel_trace.txt python $PWD/tools/bm25_ab.py --batches 1024 --variants 0
TIMELINE=24 tools/prof_cmd.sh gpurun_out/r06/r06_bm25_64_kernel_trace.txt python $PWD/tools/bm25_ab.py --batches 64 --variants 0
head -30 gpurun_out/r06/r06_bench_kernel_trace.txt
