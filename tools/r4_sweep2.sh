#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() {
  env "$@" timeout 600 python bench.py --headline-only --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d['roofline']
print('$*', 'qps', d['value'], 'ms', d['ms_per_step'], 'launch', r['launch_ms'], 'prefilter', r['prefilter'], {k:v for k,v in r['step_kernels_ms'].items() if k in ('ivf_scan','rerank','fallback_scan','ivf_sample_scan')})
"
}
run A=1
run MSVS_H16_TARGET=150
run MSVS_H16_TARGET=100
run MSVS_H16_TARGET=400
run MSVS_H16_KC=16
run MSVS_H16_KC=24
run MSVS_H16_KC=64
run MSVS_RERANK_SECOND=0
