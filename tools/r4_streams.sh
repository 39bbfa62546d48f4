#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for s in 1 2 3 4; do
timeout 600 python bench.py --headline-only --steps 60 --warmup 6 --streams $s 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('streams $s qps', d['value'], 'ms', d['ms_per_step'], 'recall', d['recall_at_10'])
"
done
