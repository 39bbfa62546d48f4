#!/bin/bash
# Round 6: the round-end sequence on one box -- the GPU suite, smoke, the default bench line (and its size)
mkdir -p gpurun_out/r06
python -m pytest tests -q -m gpu -x --timeout 1200 > gpurun_out/r06/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06/gpu_tests.log
tail -4 gpurun_out/r06/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r06/bench_line.json 2> gpurun_out/r06/bench.err; echo "bench rc=$?"
wc -c gpurun_out/r06/bench_line.json
cp bench_detail.json gpurun_out/r06/bench_detail.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_line.json").read().strip().split("\n")[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "whole", d["roofline"]["whole_step_frac"], "traffic", d["roofline"]["traffic"])
print(json.dumps(d["legs"])[:3000])
PY
