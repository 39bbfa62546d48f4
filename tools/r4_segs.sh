#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shadow or prun or h16 or candidate or second_chance" 2>&1 | tail -3
for e in MSVS_H16_SEGS=1 MSVS_H16_SEGS=0; do
for b in 16 64 256 1024 4096; do
env $e timeout 600 python bench.py --headline-only --batch $b --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d['roofline']
print('$e batch $b qps', d['value'], 'ms', d['ms_per_step'], 'scan', r['step_kernels_ms'].get('ivf_scan'))
"
done
done
