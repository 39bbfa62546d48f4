#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() {
  env "$@" timeout 600 python bench.py --only latency --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
l=d['latency']
print('$*', 'p50', l['p50_us'], 'p99', l['p99_us'], 't1', l['threads_1']['p50_us'], 't8', l['threads_8']['qps'], 't64', l['threads_64']['qps'], l['threads_64']['combined_batches'], l['threads_64']['queries_in_batches'])
"
}
run A=1
run MSVS_COMBINE_BATCHES=2
run MSVS_COMBINE_BATCHES=4
run MSVS_COMBINE=4
run MSVS_COMBINE=16
run MSVS_COMBINE=2 MSVS_COMBINE_BATCHES=3
