#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "prun or cosine or shadow_scan_items or second_chance" 2>&1 | tail -4
timeout 1200 python bench.py --only c3,c5,c4,target --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
c=d['other_configs']
print('C3', {k:v for k,v in c['C3'].items() if k in ('qps','ms_per_batch','list_scan_ms','roofline_frac','rows_read_per_batch','union_rows_per_batch','step_kernels_ms')})
print('C5', {k:v for k,v in c['C5'].items() if k in ('hybrid_qps','hybrid_ms_per_batch_median_mean_max')})
for b,v in c['C4']['batches'].items(): print('  c4', b, v['qps'], v['ms_per_batch'], 'scan', v['list_scan_ms'], 'rows', v['rows_read_per_batch'], v['union_rows_per_batch'], 'fb', v['fallback_queries'])
print(' C4 oracle', c['C4'].get('oracle_check'))
t=d['target_100m']
for b,v in t['batches'].items(): print('  target', b, v['qps'], v['ms_per_batch'], 'scan', v['list_scan_ms'], 'rows', v['rows_read_per_batch'], v['union_rows_per_batch'], 'fb', v['fallback_queries'])
print(' target oracle', t.get('oracle_check'), t['recall_at_10'])
"
