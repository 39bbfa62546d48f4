#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "prun or cosine or shadow_scan_items" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_baseline_sizes.py -x -q -m gpu -k "c3 or config3 or cosine or c5 or hybrid" 2>&1 | tail -3
timeout 900 python bench.py --only c3,c5 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
c=d['other_configs']
print('C3', {k:v for k,v in c['C3'].items() if k in ('qps','ms_per_batch','list_scan_ms','roofline_frac','rows_read_per_batch','union_rows_per_batch','step_kernels_ms')})
print('C5', {k:v for k,v in c['C5'].items() if k in ('hybrid_qps','hybrid_ms_per_batch_median_mean_max')})
"
