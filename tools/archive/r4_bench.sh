#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
timeout 1200 python bench.py > gpurun_out/r4/bench_out.json 2> gpurun_out/r4/bench_err.txt
echo "bench rc=$?"; tail -3 gpurun_out/r4/bench_err.txt
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4/bench_out.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'recall', d['recall_at_10'])
print('roofline', {k:d['roofline'][k] for k in ('achieved','frac','launch_ms','whole_step_frac','non_scan_ms_per_step','pruned_pair_fraction','step_kernels_ms')})
print('cpu', json.dumps(d['cpu_baseline'])[:400])
print('ops', json.dumps(d['operating_points']))
for k in ('other_batches','latency','blobs03','iid','latent32','target_100m'):
    print(k, json.dumps(d.get(k))[:1500])
for k,v in (d.get('other_configs') or {}).items():
    print(k, json.dumps(v)[:1200])
PY
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4/gpu_tests.txt 2>&1
tail -5 gpurun_out/r4/gpu_tests.txt
