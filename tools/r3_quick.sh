#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "second_chance or candidate or shadow or register_tile or c2_bench or baseline" 2>&1 | tail -5
( timeout 900 python bench.py --only blobs03 --no-cpu-baseline ) > gpurun_out/r3_quick.json 2> gpurun_out/r3_quick.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r3_quick.json") if x.startswith("{")][-1]
d=json.loads(l)
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
print(d["roofline"]["step_kernels_ms"], d["roofline"]["prefilter"])
b=d["blobs03"]
print("blobs03 op", b["at_recall_0.95"])
print("blobs03 cfg", b["at_config_nprobe"])
PY
