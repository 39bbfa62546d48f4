#!/bin/bash
# Round 6: headline step A/B on ONE box:  tools/r6_ab.sh "ENV=.. ENV=.." "ENV=.."   (each argument = the environment of one run; "" = defaults)
for e in "$@"; do
  for rep in 1 2; do
    env $e python bench.py --headline-only --steps 40 --warmup 10 --no-concurrent ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']
print('%-40s qps %.0f  step %.4f ms  scan %.4f  non-scan %.4f  frac %.4f  recall %s' % ('$e', d['value'], d['ms_per_step'], r['launch_ms'], r['non_scan_ms_per_step'], r['frac'], d['recall_at_10']))"
  done
done
