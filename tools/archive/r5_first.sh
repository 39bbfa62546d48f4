#!/bin/bash
# Round 5, first GPU call: the new BM25 scorer and the few-query FLAT shadow path -- parity tests, then A/B numbers, then a trace.
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py -q -m gpu -k "(bm25 or flat_shadow) and not vector_top100" --maxfail 6 --timeout 900 > gpurun_out/r5_t1.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5_t1.log
tail -15 gpurun_out/r5_t1.log
timeout 900 python tools/r5_bm25_ab.py --batches 16,64,256,1024 > gpurun_out/r5_bm25_ab.txt 2>&1
tail -30 gpurun_out/r5_bm25_ab.txt
timeout 600 python tools/r5_flat_lat.py > gpurun_out/r5_flat_lat.txt 2>&1
tail -14 gpurun_out/r5_flat_lat.txt
timeout 600 tools/prof_cmd.sh gpurun_out/r5_bm25_trace.txt python tools/r5_bm25_ab.py --batches 64,1024 --variants 0
head -30 gpurun_out/r5_bm25_trace.txt
