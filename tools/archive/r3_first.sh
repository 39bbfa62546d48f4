#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "register_tile" 2>&1 | tail -15 > gpurun_out/r3_first_tests.txt
cat gpurun_out/r3_first_tests.txt
timeout 600 python tools/ivf_sweep.py $R3_ARGS > gpurun_out/r3_first_sweep.txt 2>&1
cat gpurun_out/r3_first_sweep.txt
