#!/bin/bash
# round 3: the full bench line + the tests touched by the bench / sharding / trainer changes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 1500 python bench.py $R3_BENCH_ARGS ) > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
tail -c 600 gpurun_out/r3_bench.err
python - <<'PY'
import json
try:
    l=[x for x in open("gpurun_out/r3_bench.json") if x.startswith("{")][-1]
    d=json.loads(l)
    def short(o, depth=0):
        if isinstance(o, dict):
            return {k: short(v, depth+1) for k, v in o.items() if k not in ("note","kernel","sample","workload","data","api","driver")}
        return o
    print(json.dumps(short(d), indent=1)[:9000])
except Exception as e:
    print("no json", e)
PY
