#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
SWEEP_IID=1 timeout 600 python tools/ivf_sweep.py B=4096 B=4096,h16_stamps=1 2>&1 | grep -v amdgpu.ids | cut -c1-420
SWEEP_DATA=blobs03 timeout 600 python tools/ivf_sweep.py B=4096 2>&1 | grep -v amdgpu.ids | cut -c1-420
timeout 600 python tools/ivf_sweep.py B=4096 2>&1 | grep -v amdgpu.ids | cut -c1-420
PROBE_PARAMS="" timeout 300 python tools/prune_probe.py 2>&1 | grep -v amdgpu.ids | head -4
