#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shadow or prun or h16 or candidate or second_chance" 2>&1 | tail -3
run() {
  env "$@" timeout 600 python bench.py --headline-only --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d['roofline']
print('$*', 'qps', d['value'], 'ms', d['ms_per_step'], 'launch', r['launch_ms'], 'prefilter', r['prefilter'], {k:v for k,v in r['step_kernels_ms'].items() if k in ('ivf_scan','rerank','ivf_sample_scan')})
"
}
run MSVS_H16_PREFETCH=1
run MSVS_H16_PREFETCH=0
run MSVS_H16_PREFETCH=1
run MSVS_H16_PREFETCH=0
