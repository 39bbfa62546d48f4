#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shadow or candidate or flat_shadow" 2>&1 | tail -2
timeout 900 python bench.py --only iid,latent32 --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d['roofline']
print('headline', d['value'], d['ms_per_step'], r['launch_ms'], r['step_kernels_ms'])
i=d['iid']; print(' iid nprobe32', i['at_config_nprobe']['qps'], i['at_config_nprobe']['list_scan_ms'], 'ivf256', i['exhaustive_ivf256']['qps'], 'flat', i['exhaustive_flat']['qps'])
l=d['latent32']; print(' latent32 np1', l['at_recall_0.95']['qps'], 'np32', l['at_config_nprobe']['qps'], l['at_config_nprobe']['list_scan_ms'])
"
