#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests/test_sharded_gloo.py tests/test_shim.py -x -q -m gpu > gpurun_out/r4/t1_sharded.txt 2>&1
tail -15 gpurun_out/r4/t1_sharded.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "prun or items_of_every" > gpurun_out/r4/t1_prune.txt 2>&1
tail -15 gpurun_out/r4/t1_prune.txt
