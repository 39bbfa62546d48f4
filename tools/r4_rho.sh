#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "error_bound_on_hardware or prun or coarse_band or flat_shadow or few_query" -s 2>&1 | grep -v "^$" | tail -40
for s in 1; do
  timeout 600 python bench.py --headline-only --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d['roofline']
print('qps', d['value'], 'ms', d['ms_per_step'], 'launch', r['launch_ms'], 'pruned', r['pruned_pair_fraction'], 'rows_read', r['rows_read_per_step'], r['step_kernels_ms'])
"
done
