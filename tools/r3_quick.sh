#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "coarse_band or c2_bench or mfma_accumulation or massive_ties or sharded" 2>&1 | tail -5
for b in 1 0; do
MSVS_COARSE_BAND=$b timeout 300 python bench.py --headline-only --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads([x for x in sys.stdin if x.startswith('{')][-1])
print('band=$b value', d['value'], d['ms_per_step'], d['roofline']['step_kernels_ms'], d['roofline']['prefilter'])
"
done
