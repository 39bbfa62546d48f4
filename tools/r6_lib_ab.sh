#!/bin/bash
# Round 6: headline step A/B of several BUILDS of libmsvs.so on ONE box (ab_libs/libmsvs_<tag>.so):  tools/r6_lib_ab.sh d2 d4 d6
for rep in 1 2; do
for t in "$@"; do
  cp ab_libs/libmsvs_$t.so myscaledb_amd/libmsvs.so
  python bench.py --headline-only --steps 40 --warmup 10 --no-concurrent ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']
det=json.load(open('bench_detail.json'))['roofline']['step_kernels_ms']
print('%-8s qps %.0f  step %.4f ms  scan %.4f  non-scan %.4f  coarse %.4f rerank %.4f' % ('$t', d['value'], d['ms_per_step'], r['launch_ms'], r['non_scan_ms_per_step'], det['coarse_pass'], det['rerank']))"
done
done
