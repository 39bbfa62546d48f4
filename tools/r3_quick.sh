#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python bench.py --only c4 --no-cpu-baseline ) > gpurun_out/r3_quick.json 2> gpurun_out/r3_quick.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r3_quick.json") if x.startswith("{")][-1]
d=json.loads(l)
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
print(d["roofline"]["step_kernels_ms"], d["roofline"]["prefilter"])
c4=d["other_configs"]["C4"]
for b,v in c4["batches"].items(): print("C4", b, v["qps"], v["ms_per_batch"], v["list_scan_ms"], v["roofline_frac"])
print(c4["oracle_check"])
PY
MSVS_H16_RING=4 timeout 600 python bench.py --only c4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([x for x in sys.stdin if x.startswith('{')][-1])
print('ring4 value', d['value'], d['roofline']['step_kernels_ms']['ivf_scan'])
for b,v in d['other_configs']['C4']['batches'].items(): print('ring4 C4', b, v['qps'], v['list_scan_ms'], v['roofline_frac'])
"
