#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "probe_pruning or c2_bench or ivfflat_matches or matrix_core or shadow or c4_shape or second_chance" 2>&1 | tail -4 > gpurun_out/r3_quick_tests.txt
cat gpurun_out/r3_quick_tests.txt
timeout 600 python bench.py --only c4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([x for x in sys.stdin if x.startswith('{')][-1])
print('value', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_kernels_ms'], d['roofline']['prefilter'])
for b,v in d['other_configs']['C4']['batches'].items(): print('C4', b, v['qps'], v['list_scan_ms'], v['roofline_frac'])
"
