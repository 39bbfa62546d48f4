#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python tools/ivf_sweep.py $R3_ARGS > gpurun_out/r3_ablate.txt 2>&1
cat gpurun_out/r3_ablate.txt
