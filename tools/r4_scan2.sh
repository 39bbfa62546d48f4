#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
timeout 600 python tools/ivf_sweep.py B=4096 B=4096,h16_stamps=1 B=4096,h16_join=0 B=4096,h16_join=0,h16_stamps=1 B=1024 B=1024,h16_join=0 B=256 B=256,h16_join=0 B=64 > gpurun_out/r4/scan2_latent.txt 2>&1
cat gpurun_out/r4/scan2_latent.txt
SWEEP_DATA=blobs03 timeout 600 python tools/ivf_sweep.py B=4096 B=4096,h16_stamps=1 B=4096,h16_join=0 B=1024 B=1024,h16_join=0 > gpurun_out/r4/scan2_blobs.txt 2>&1
cat gpurun_out/r4/scan2_blobs.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "handout or h16 or shadow or prun or candidate or certif or second or band" > gpurun_out/r4/parity2.txt 2>&1
tail -5 gpurun_out/r4/parity2.txt
