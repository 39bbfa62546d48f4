#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu -k "fusion or bm25 or c5" 2>&1 | tail -6
( timeout 900 python bench.py --only c5 --no-cpu-baseline ) > gpurun_out/r3_c5.json 2> gpurun_out/r3_c5.err
tail -c 400 gpurun_out/r3_c5.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r3_c5.json") if x.startswith("{")][-1]
d=json.loads(l)
print(json.dumps(d["other_configs"]["C5"], indent=1))
PY
