#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "flat or knn or few" --maxfail 5 --timeout 900 2>&1 | tail -5
timeout 600 python tools/r5_flat_lat.py --few-only 2>&1 | grep -v "^W2026\|amdgpu.ids"
