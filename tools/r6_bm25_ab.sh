#!/bin/bash
# Round 6: BM25 A/B of two builds of libmsvs.so on ONE box (ab_libs/libmsvs_<tag>.so):  tools/r6_bm25_ab.sh w4 w3
for rep in 1 2; do
for t in "$@"; do
  cp ab_libs/libmsvs_$t.so myscaledb_amd/libmsvs.so
  echo "== $t"
  python tools/bm25_ab.py --batches 64,256,1024 --variants 0 2>&1 | grep "^batch"
done
done
