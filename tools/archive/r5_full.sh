#!/bin/bash
# Round 5: the round-end sequence on one box -- the GPU suite, smoke, the default bench line
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x --timeout 1200 > gpurun_out/r5_gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5_gpu_tests.log
tail -4 gpurun_out/r5_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench_output.json 2> gpurun_out/r5_bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r5_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5_bench_output.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "whole", d["roofline"]["whole_step_frac"])
oc = d.get("other_configs") or {}
c5 = oc.get("C5") or {}
for k in ("bm25_batch64", "bm25_batch256", "bm25_batch1024", "bm25_batch4096"):
    print(k, {kk: c5.get(k, {}).get(kk) for kk in ("us_per_query", "hbm_frac", "ms_per_batch", "error")})
iid = d.get("iid") or {}
print("iid flat_latency", iid.get("flat_latency"))
print("iid exhaustive_flat", {k: (iid.get("exhaustive_flat") or {}).get(k) for k in ("qps", "whole_step_mfma_frac")})
print("iid at_config", {k: (iid.get("at_config_nprobe") or {}).get(k) for k in ("qps", "roofline_frac", "recall")})
print("mid", json.dumps((d.get("operating_points") or {}).get("mid"))[:900])
lat = d.get("latency") or {}
print("latency", {k: lat.get(k) for k in ("p50_us", "p99_us", "host_pointer_batch4096", "host_pointer_batch4096_two_threads")}, (lat.get("threads_64") or {}).get("qps"), (lat.get("threads_128") or {}).get("qps"))
print("small batches", lat.get("host_pointer_small_batches"))
print("setup", d.get("setup_s"), "errors", [k for k, v in d.items() if isinstance(v, dict) and "error" in v])
PY
