#!/bin/bash
# Round 6, first GPU call: the suite + the default bench (is the compact line parseable?) + a timeline of the headline step.
mkdir -p gpurun_out/r06
python -m pytest tests -q -m gpu -x --timeout 1200 > gpurun_out/r06/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06/gpu_tests.log
tail -4 gpurun_out/r06/gpu_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_line.json 2> gpurun_out/r06/bench.err; echo "bench rc=$?"
wc -c gpurun_out/r06/bench_line.json; cat gpurun_out/r06/bench_line.json
cp bench_detail.json gpurun_out/r06/bench_detail_first.json
TIMELINE=120 tools/prof_cmd.sh gpurun_out/r06/step_trace.txt python bench.py --headline-only --steps 4 --warmup 2 --no-concurrent
head -40 gpurun_out/r06/step_trace.txt
