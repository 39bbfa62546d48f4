#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for e in MSVS_H16_SEGS=1 MSVS_H16_SEGS=0; do
env $e timeout 900 python bench.py --only target,c4 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
t=d['target_100m']
print('$e target recall', t['recall_at_10'], t.get('oracle_check'))
for b,v in t['batches'].items(): print('  target', b, v['qps'], v['ms_per_batch'], 'scan', v['list_scan_ms'], 'frac', v['roofline_frac'], 'fb', v['fallback_queries'])
c=d['other_configs']['C4']
print(' C4 oracle', c.get('oracle_check'))
for b,v in c['batches'].items(): print('  c4', b, v['qps'], v['ms_per_batch'], 'scan', v['list_scan_ms'], 'frac', v['roofline_frac'], 'fb', v['fallback_queries'])
"
done
