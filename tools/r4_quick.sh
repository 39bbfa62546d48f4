#!/bin/bash
# quick check on the GPU box: a parity subset + the headline step (args: extra env assignments for a second headline run)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${PYTEST_K:-shadow or prun or h16 or candidate}" 2>&1 | tail -3
run() {
  env "$@" timeout 600 python bench.py --headline-only --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d['roofline']
print('$*', 'qps', d['value'], 'ms', d['ms_per_step'], 'launch', r['launch_ms'], 'prefilter', r['prefilter'], r['step_kernels_ms'])
"
}
run A=1
run A=2
for e in "$@"; do run $e; done
