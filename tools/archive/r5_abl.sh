#!/bin/bash
R=$(pwd); mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py -q -m gpu -k "bm25 and not vector_top100" --maxfail 6 --timeout 900 2>&1 | tail -5
timeout 900 python tools/r5_bm25_ab.py --batches 64,1024 --variants ${1:-2,0,10} > gpurun_out/r5_bm25_abl.txt 2>&1
cat gpurun_out/r5_bm25_abl.txt | grep batch
