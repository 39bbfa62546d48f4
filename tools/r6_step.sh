#!/bin/bash
# Round 6: kernel trace + timeline of the headline step (single stream), optional data model:  tools/r6_step.sh <tag> [bench args]
TAG=${1:-step}; shift
mkdir -p gpurun_out/r06
TIMELINE=${TIMELINE:-60} tools/prof_cmd.sh gpurun_out/r06/${TAG}_trace.txt python $PWD/bench.py --headline-only --steps 6 --warmup 2 --no-concurrent "$@"
head -50 gpurun_out/r06/${TAG}_trace.txt
tail -70 gpurun_out/r06/${TAG}_trace_timeline.txt
