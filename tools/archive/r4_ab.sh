#!/bin/bash
# A/B of option settings inside ONE box: tools/r4_ab.sh "ENV1=.." "ENV2=.." ... (each run twice, interleaved)
cd "$GRAFT_REPO_ROOT" || exit 1
run() {
  env "$@" timeout 600 python bench.py --headline-only --steps 40 --warmup 5 ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d['roofline']
print('$*', 'qps', d['value'], 'ms', d['ms_per_step'], 'launch', r['launch_ms'], 'prefilter', r['prefilter'], r['step_kernels_ms'])
"
}
for rep in 1 2; do for e in "$@"; do run $e; done; done
