#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "second_chance or prun or candidate or certificate or fallback" 2>&1 | tail -3
timeout 900 python bench.py --only target --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d['roofline']
print('headline qps', d['value'], 'ms', d['ms_per_step'], r['step_kernels_ms'])
t=d['target_100m']
for b,v in t['batches'].items(): print('  target', b, v['qps'], v['ms_per_batch'], 'scan', v['list_scan_ms'], 'frac', v['roofline_frac'], 'fb', v['fallback_queries'], v['step_kernels_ms'])
"
