#!/bin/bash
# full GPU suite, then the headline + blobs03/iid legs, then a kernel trace of the headline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r3_tests.txt
cat gpurun_out/r3_tests.txt
( timeout 900 python bench.py --only ${R3_ONLY:-blobs03} --no-cpu-baseline ) > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
tail -c 300 gpurun_out/r3_bench.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r3_bench.json") if x.startswith("{")][-1]
d=json.loads(l)
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "whole", d["roofline"]["whole_step_frac"], "nonscan", d["roofline"]["non_scan_ms_per_step"])
print(d["roofline"]["step_kernels_ms"], d["roofline"]["prefilter"])
b=d.get("blobs03")
if b: print("blobs03", b["at_recall_0.95"])
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/hl -o t -- python "$GRAFT_REPO_ROOT/bench.py" --headline-only --steps 20 --warmup 5 --no-cpu-baseline > /tmp/hl.log 2>&1
cd "$GRAFT_REPO_ROOT" && python tools/rocprof_summary.py $(find /tmp/hl -name "*.db" | head -1) > gpurun_out/r3_headline_trace.txt 2>&1; grep -v "at::native\|assign_kernel\|centroid_update\|gather_rows\|scatter_rows\|absmax\|h16_build\|Cijk" gpurun_out/r3_headline_trace.txt | head -36
