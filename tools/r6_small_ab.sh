#!/bin/bash
# Round 6: small batches A/B on one box (env per run)
for e in "$@"; do for rep in 1 2; do
  echo "== $e"; env $e SWEEP_DATA=blobs03 python tools/ivf_sweep.py B=16 B=32 B=64 B=256 2>&1 | grep "^B=" | sed -e "s/same_ids.*kernels/kernels/" | cut -c1-230
done; done
