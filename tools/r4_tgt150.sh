#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for e in MSVS_H16_TARGET=250 MSVS_H16_TARGET=150; do
env $e timeout 1200 python bench.py --only iid,latent32,target,c3 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$e headline', d['value'])
i=d['iid']; print(' iid nprobe32', i['at_config_nprobe']['qps'], i['at_config_nprobe']['fallback_queries'], 'ivf256', i['exhaustive_ivf256']['qps'], i['exhaustive_ivf256']['fallback_queries'], 'flat', i['exhaustive_flat']['qps'])
l=d['latent32']; print(' latent32 np1', l['at_recall_0.95']['qps'], l['at_recall_0.95']['fallback_queries'], 'np32', l['at_config_nprobe']['qps'], l['at_config_nprobe']['fallback_queries'])
t=d['target_100m']
for b,v in t['batches'].items(): print('  target', b, v['qps'], v['ms_per_batch'], 'scan', v['list_scan_ms'], 'fb', v['fallback_queries'])
c=d['other_configs']['C3']; print(' C3', c['qps'], c['ms_per_batch'])
"
done
