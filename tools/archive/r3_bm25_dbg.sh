#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "bm25 or c5 or text" 2>&1 | tail -5
{
echo "## default"; timeout 300 python tools/bm25_bench.py --batches 1,16,64,256 2>&1 | grep BM25
echo "## dense accumulator (bm25_posting=0)"; MSVS_BM25_POSTING=0 timeout 300 python tools/bm25_bench.py --batches 64,256 2>&1 | grep BM25 | sed 's/algorithmic.*//'
} > gpurun_out/r3_bm25_bench.txt 2>&1
cat gpurun_out/r3_bm25_bench.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/bm25prof -o t -- python "$GRAFT_REPO_ROOT/tools/bm25_bench.py" --batches 64 > /tmp/bm25prof.log 2>&1
cd "$GRAFT_REPO_ROOT" && python tools/rocprof_summary.py $(find /tmp/bm25prof -name "*.db" | head -1) > gpurun_out/r3_bm25_trace.txt 2>&1; grep -i "bm25\|merge\|fillBuffer" gpurun_out/r3_bm25_trace.txt | head -14
