#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
timeout 900 python tools/flat_sweep.py B=64 B=64,h16_flush_each=1 B=64 B=16 B=16,h16_flush_each=1 B=128 B=128,h16_flush_each=1 B=256 B=256,h16_flush_each=1 B=256 B=4096 B=4096,h16_flush_each=1 > gpurun_out/r4/flat_sweep6.txt 2>&1
cat gpurun_out/r4/flat_sweep6.txt | cut -c1-300
timeout 600 python tools/ivf_sweep.py B=4096 B=4096,h16_flush_each=1 B=4096 B=4096,h16_flush_each=1 B=1024 B=1024,h16_flush_each=1 B=256 B=256,h16_flush_each=1 B=64 B=64,h16_flush_each=1 > gpurun_out/r4/scan4_latent.txt 2>&1
cat gpurun_out/r4/scan4_latent.txt | cut -c1-400
