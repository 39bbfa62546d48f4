#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r3_quick_tests.txt
cat gpurun_out/r3_quick_tests.txt
timeout 600 python bench.py --only c3,blobs03 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([x for x in sys.stdin if x.startswith('{')][-1])
print('value', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['pruned_pair_fraction'], d['roofline']['step_kernels_ms'], d['roofline']['prefilter'])
print('C3', d['other_configs']['C3'])
print('blobs03', d['blobs03']['at_recall_0.95'])
print('blobs03 cfg', d['blobs03']['at_config_nprobe'])
"
