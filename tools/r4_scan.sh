#!/bin/bash
# round 4, first GPU call: where the time of the shadow list scan goes on short lists (per-item stamps) + tile / occupancy variants
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
timeout 900 python tools/ivf_sweep.py \
  B=4096 \
  B=4096,h16_stamps=1 \
  B=4096,h16_item_ncb=1 \
  B=4096,h16_item_ncb=1,h16_stamps=1 \
  B=4096,h16_ncb=2,h16_item_ncb=1 \
  B=4096,h16_ncb=2,h16_half=1,h16_item_ncb=1 \
  B=4096,h16_ncb=2,h16_half=1,h16_item_ncb=1,h16_stamps=1 \
  B=4096,h16_ncb=2,h16_half=2,h16_item_ncb=1 \
  B=4096,h16_ncb=2,h16_half=2,h16_item_ncb=1,h16_stamps=1 \
  B=4096,h16_ncb=1,h16_item_ncb=1 \
  B=4096,h16_nseg=2 \
  B=4096,h16_nseg=4,h16_item_ncb=1 \
  B=4096,h16_nseg=4,h16_item_ncb=1,h16_stamps=1 \
  B=4096,h16_ncb=2,h16_half=1,h16_item_ncb=1,h16_nseg=2 \
  B=4096,h16_ncb=2,h16_half=1,h16_item_ncb=1,h16_nseg=4 \
  B=4096,h16_ncb=3,h16_half=1,h16_item_ncb=1 \
  B=4096,h16_prune=0 \
  B=4096,h16_prune=0,h16_ncb=2,h16_half=1,h16_item_ncb=1 \
  B=1024 \
  B=1024,h16_item_ncb=1 \
  B=1024,h16_ncb=2,h16_half=1,h16_item_ncb=1 \
  B=1024,h16_ncb=1,h16_item_ncb=1 \
  B=256 \
  B=256,h16_ncb=2,h16_half=1,h16_item_ncb=1 \
  > gpurun_out/r4/scan_sweep.txt 2>&1
tail -40 gpurun_out/r4/scan_sweep.txt
# parity of the new tile shapes on the test suite's shapes (ragged d, filters, ties, overflow, pruning)
MSVS_H16_ITEM_NCB=1 MSVS_H16_HALF=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "h16 or shadow or prun or candidate or batch or certif or second or band" > gpurun_out/r4/parity_half1.txt 2>&1
tail -5 gpurun_out/r4/parity_half1.txt
MSVS_H16_ITEM_NCB=1 MSVS_H16_HALF=2 MSVS_H16_NSEG=3 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "h16 or shadow or prun or candidate or batch or certif or second or band" > gpurun_out/r4/parity_half2.txt 2>&1
tail -5 gpurun_out/r4/parity_half2.txt
