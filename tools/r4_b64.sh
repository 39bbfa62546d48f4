#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for b in 16 64 256; do
timeout 600 python bench.py --headline-only --batch $b --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d['roofline']
print('batch $b qps', d['value'], 'ms', d['ms_per_step'], r['step_kernels_ms'])
"
done
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/b64 -o t -- python $GRAFT_REPO_ROOT/bench.py --headline-only --batch 64 --steps 200 --warmup 5 > /dev/null 2>&1
db=$(find /tmp/b64 -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db | awk '$0 ~ / (2[0-9][0-9]|4[0-9][0-9]) +[0-9.]+ +[0-9.]+ +[0-9.]+ +[0-9.]+ +[0-9.]+$/' | cut -c1-60,95-170
