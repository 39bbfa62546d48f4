#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
SWEEP_DATA=blobs03 timeout 600 python tools/ivf_sweep.py B=4096 B=256,h16_preprune=0 B=256 B=64,h16_preprune=0 B=64 B=16,h16_preprune=0 B=16 B=4,h16_preprune=0 B=4 2>&1 | grep -v amdgpu.ids | cut -c1-420
timeout 600 python tools/ivf_sweep.py B=64,h16_preprune=0 B=64 B=16,h16_preprune=0 B=16 2>&1 | grep -v amdgpu.ids | cut -c1-420
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_sharded_gloo.py -x -q -m gpu -k "prun or sharded or ivf or batch" 2>&1 | tail -4
