#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
SWEEP_DATA=blobs03 timeout 600 python tools/ivf_sweep.py B=4096,h16_preprune=0 B=4096 B=1024,h16_preprune=0 B=1024 B=256,h16_preprune=0 B=256 2>&1 | grep -v amdgpu.ids | cut -c1-420
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_sharded_gloo.py -x -q -m gpu -k "prun or sharded" 2>&1 | tail -4
