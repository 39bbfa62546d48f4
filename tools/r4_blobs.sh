#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
for np in 32 2; do
SWEEP_DATA=blobs03 SWEEP_NPROBE=$np timeout 600 python tools/ivf_sweep.py \
  B=4096,rerank_stats=1 \
  B=4096,h16_stamps=1 \
  B=4096,h16_item_ncb=1,h16_stamps=1 \
  B=4096,h16_ncb=1,h16_item_ncb=1,h16_stamps=1 \
  B=4096,h16_prune=0 \
  B=4096,h16_prune=2,h16_item_ncb=1,h16_stamps=1 \
  B=1024 B=1024,h16_item_ncb=1 \
  > gpurun_out/r4/blobs_np$np.txt 2>&1
cat gpurun_out/r4/blobs_np$np.txt
done
