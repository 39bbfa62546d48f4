#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python bench.py --only iid --no-cpu-baseline ) > gpurun_out/r3_iid.json 2> gpurun_out/r3_iid.err
tail -c 600 gpurun_out/r3_iid.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r3_iid.json") if x.startswith("{")][-1]
d=json.loads(l)
i=d["iid"]
for k in ("recall_at_10","exhaustive_ivf256","exhaustive_flat","at_recall_0.95"):
    print(k, json.dumps(i.get(k)))
PY
