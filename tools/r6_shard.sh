#!/bin/bash
# Round 6: the multi-rank forms (W = 2 / 4 / 8 on the one GPU, gloo) + the single-rank RCCL group path
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_sharded_many_ranks.py tests/test_sharded_gloo.py -q -m gpu -x --timeout 1200 > gpurun_out/r06/shard_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06/shard_tests.log
tail -40 gpurun_out/r06/shard_tests.log
