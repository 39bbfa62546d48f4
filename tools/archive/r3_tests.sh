#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
if [ -n "$R3_K" ]; then
  timeout 1700 python -m pytest tests -x -q -m gpu -k "$R3_K" 2>&1 | tail -25 > gpurun_out/r3_tests.txt
else
  timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r3_tests.txt
fi
cat gpurun_out/r3_tests.txt
