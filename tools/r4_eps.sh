#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
for s in 1 0.5 0.35; do
  MSVS_IVF_EPS_SCALE=$s timeout 600 python bench.py --headline-only --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d['roofline']
print('eps_scale $s', 'qps', d['value'], 'ms', d['ms_per_step'], 'recall', d['recall_at_10'], 'launch', r['launch_ms'], 'pruned', r['pruned_pair_fraction'], 'rows_read', r['rows_read_per_step'], r['step_kernels_ms'])
"
done
