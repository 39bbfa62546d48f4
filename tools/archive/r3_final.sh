#!/bin/bash
# round-3 closing run: smoke(), the full bench line, the headline kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( timeout 1200 python bench.py ) > gpurun_out/r3_final_bench.json 2> gpurun_out/r3_final_bench.err
tail -c 200 gpurun_out/r3_final_bench.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r3_final_bench.json") if x.startswith("{")][-1]
d=json.loads(l)
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "whole", d["roofline"]["whole_step_frac"], "nonscan", d["roofline"]["non_scan_ms_per_step"], "p50", d["p50_ms_batch1"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["oracle_check"])
print("iid", d["iid"]["at_recall_0.95"]["qps"], "blobs", d["blobs03"]["at_recall_0.95"]["qps"])
oc=d["other_configs"]
print("C3", oc["C3"]["qps"], "C4", {b:v["qps"] for b,v in oc["C4"]["batches"].items()}, oc["C4"]["batches"]["4096"]["roofline_frac"], oc["C4"]["oracle_check"])
print("C5", oc["C5"]["hybrid_qps"], oc["C5"]["bm25_batch64"]["us_per_query"], "lat64", d["latency"]["threads_64"]["qps"])
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/hl -o t -- python "$GRAFT_REPO_ROOT/bench.py" --headline-only --steps 20 --warmup 5 --no-cpu-baseline > /tmp/hl.log 2>&1
cd "$GRAFT_REPO_ROOT" && python tools/rocprof_summary.py $(find /tmp/hl -name "*.db" | head -1) > gpurun_out/r3_headline_trace.txt 2>&1; grep "h16_scan\|h16_sample_k\|ivf_rerank_kernel\|coarse_h16" gpurun_out/r3_headline_trace.txt | head -5
