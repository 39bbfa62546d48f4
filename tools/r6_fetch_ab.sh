#!/bin/bash
# Round 6: hipMemcpyAsync vs the copy kernel for small pinned hand-overs (option pinned_fetch), BM25 batches and small host-pointer batches
for v in 0 1 0 1; do
  echo "== pinned_fetch=$v"
  MSVS_PINNED_FETCH=$v python tools/bm25_ab.py --batches 64,256,1024 --variants 0 2>&1 | grep "^batch" | cut -c1-150
  MSVS_PINNED_FETCH=$v python bench.py --only latency --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])['legs']['latency']; print(d)"
done
