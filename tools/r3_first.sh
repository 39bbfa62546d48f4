#!/bin/bash
# round 3, first GPU contact of the register-tile list scan: parity test, then A/B timing on the bench index
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "register_tile" 2>&1 | tail -15 > gpurun_out/r3_first_tests.txt
cat gpurun_out/r3_first_tests.txt
timeout 600 python tools/ivf_sweep.py B=4096,h16_reg=0 B=4096,h16_reg=1 B=4096,h16_reg=1,h16_grid=512 B=1024,h16_reg=0 B=1024,h16_reg=2 B=16384,h16_reg=0 B=16384,h16_reg=1 > gpurun_out/r3_first_sweep.txt 2>&1
cat gpurun_out/r3_first_sweep.txt
