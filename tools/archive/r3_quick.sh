#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for ncb in 0 2 1; do
  MSVS_H16_NCB=$ncb timeout 300 python bench.py --headline-only --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads([x for x in sys.stdin if x.startswith('{')][-1])
print('ncb=$ncb value', d['value'], d['ms_per_step'], d['roofline']['step_kernels_ms']['ivf_scan'], d['roofline']['pruned_pair_fraction'], d['roofline']['prefilter'])
"
done
