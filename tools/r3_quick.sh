#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2; do
timeout 600 python bench.py --headline-only --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads([x for x in sys.stdin if x.startswith('{')][-1])
print('value', d['value'], d['ms_per_step'], d['roofline']['step_kernels_ms']['ivf_scan'], d['roofline']['prefilter'])
"
done
