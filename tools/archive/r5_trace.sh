#!/bin/bash
# Round 5: kernel traces of the BM25 batch legs and of the few-query FLAT shadow path (run on the GPU box from the repo root).
R=$(pwd); mkdir -p gpurun_out
timeout 600 tools/prof_cmd.sh gpurun_out/r5_bm25_trace64.txt python $R/tools/r5_bm25_ab.py --batches 64 --variants 0
head -24 gpurun_out/r5_bm25_trace64.txt
timeout 600 tools/prof_cmd.sh gpurun_out/r5_bm25_trace1024.txt python $R/tools/r5_bm25_ab.py --batches 1024 --variants 0
head -24 gpurun_out/r5_bm25_trace1024.txt
timeout 600 tools/prof_cmd.sh gpurun_out/r5_flat_lat_trace.txt python $R/tools/r5_flat_lat.py --few-only
head -40 gpurun_out/r5_flat_lat_trace.txt
