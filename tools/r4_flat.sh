#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
SWEEP_ROWS=10000 SWEEP_DIM=128 timeout 600 python tools/flat_sweep.py B=1000,flat_h16=0 B=1000 B=1000,flat_segb=8 B=1000,flat_segb=16 B=1000,flat_segb=32 B=1000,flat_segb=64 B=1000,flat_h16=3 B=100,flat_h16=0 B=100 B=4096,flat_h16=0 B=4096 > gpurun_out/r4/flat_c1.txt 2>&1
cat gpurun_out/r4/flat_c1.txt | cut -c1-330
SWEEP_ROWS=100000 SWEEP_DIM=128 timeout 600 python tools/flat_sweep.py B=1000,flat_h16=0 B=1000 B=100,flat_h16=0 B=100 > gpurun_out/r4/flat_c1b.txt 2>&1
cat gpurun_out/r4/flat_c1b.txt | cut -c1-330
