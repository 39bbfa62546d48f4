#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
timeout 600 python bench.py --only other_batches --steps 20 > gpurun_out/r4/bench_small.json 2> gpurun_out/r4/bench_small_err.txt; echo rc=$?
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4/bench_small.json') if l.startswith('{')][-1])
print('value', d['value'], 'roofline', {k:d['roofline'][k] for k in ('achieved','frac','launch_ms','rows_read_per_step','rows_probed_union_per_step','whole_step_frac')})
print(json.dumps(d['other_batches']))
PY
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py -x -q -m gpu -k "centroid_shadow_error or c4_one_gpu or flat_shadow or radius_pruning" 2>&1 | grep -E "passed|failed|Error|assert|FAILED" | tail -8
