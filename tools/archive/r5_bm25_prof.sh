#!/bin/bash
# Round 5: kernel trace (+ timeline of the last batches) of the BM25 batch at 64 / 1024 queries (default scorer)
R=$(pwd); mkdir -p gpurun_out
for B in ${BS:-64 1024}; do
TIMELINE=16 timeout 600 tools/prof_cmd.sh gpurun_out/r5_bm25_trace$B.txt python $R/tools/r5_bm25_ab.py --batches $B --variants 0
grep -i "bm25\|merge_k" gpurun_out/r5_bm25_trace$B.txt | cut -c1-190
grep "^batch" gpurun_out/r5_bm25_trace$B.txt.log
cat gpurun_out/r5_bm25_trace${B}_timeline.txt
done
