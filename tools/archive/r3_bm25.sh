#!/bin/bash
# BM25 posting-as-unit scorer: parity tests, then tools/bm25_bench.py with the posting scorer and the dense one, then a kernel trace.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "bm25 or c5 or text" 2>&1 | tail -15 > gpurun_out/r3_bm25_tests.txt
cat gpurun_out/r3_bm25_tests.txt
{
  echo "## posting scorer (default)"; timeout 600 python tools/bm25_bench.py --batches 1,16,64,256 2>&1 | grep BM25
  echo "## sub_docs sweep at batch 64"
  for sd in 2048 4096 8192; do echo "# bm25_sub_docs=$sd"; MSVS_BM25_SUB_DOCS=$sd timeout 300 python tools/bm25_bench.py --batches 64 2>&1 | grep BM25 | sed 's/algorithmic.*//'; done
  echo "## experiment masks at sub_docs=4096, batch 64"
  for dbg in 8 7 3 1 4; do echo "# bm25_dbg=$dbg"; MSVS_BM25_SUB_DOCS=4096 MSVS_BM25_DBG=$dbg timeout 300 python tools/bm25_bench.py --batches 64 2>&1 | grep BM25 | sed 's/algorithmic.*//'; done
  echo "## dense accumulator (bm25_posting=0)"; MSVS_BM25_POSTING=0 timeout 600 python tools/bm25_bench.py --batches 64 2>&1 | grep BM25
} > gpurun_out/r3_bm25_bench.txt 2>&1
cat gpurun_out/r3_bm25_bench.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/bm25prof -o t -- python "$GRAFT_REPO_ROOT/tools/bm25_bench.py" --batches 64 > /tmp/bm25prof.log 2>&1
cd "$GRAFT_REPO_ROOT" && python tools/rocprof_summary.py $(find /tmp/bm25prof -name "*.db" | head -1) > gpurun_out/r3_bm25_trace.txt 2>&1; grep -i "bm25\|merge" gpurun_out/r3_bm25_trace.txt | head -12
