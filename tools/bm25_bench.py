"""BM25 scorer throughput (seam B): batches of 3-term queries over a synthetic Zipf corpus through the stream-ordered
device entry point; algorithmic bytes = SURVEY 8d's model (8 B per posting + one fieldnorm byte per touched document).

    python tools/bm25_bench.py [--docs 10000000] [--batches 1,64,256]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myscaledb_amd.capi as capi  # noqa: E402
from bench import build_postings  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--vocab", type=int, default=200_000)
    ap.add_argument("--batches", default="1,16,64,256")
    ap.add_argument("--k", type=int, default=100)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    capi.set_device(0)
    ps, df_all, total, n_post = build_postings(a.docs, a.vocab)
    rng = np.random.default_rng(5)
    mids = np.argsort(-df_all)[50:2000]
    stream = torch.cuda.current_stream().cuda_stream
    for B in [int(b) for b in a.batches.split(",")]:
        oi = torch.empty((B, a.k), device=dev, dtype=torch.int64)
        od = torch.empty((B, a.k), device=dev, dtype=torch.float32)
        sets = []
        for _ in range(6):
            qs = [rng.choice(mids, 3, replace=False) for _ in range(B)]
            dfs_ = [df_all[q] for q in qs]
            sets.append((qs, dfs_, ps.prepare_batch(qs, dfs_, total)))
        byts = np.mean([sum(int(d.sum()) * 8 + min(int(d.sum()), a.docs) for d in dfs) for _, dfs, _ in sets])
        for qs, dfs, prep in sets[:2]:
            ps.bm25_search_batch_device(qs, dfs, a.docs, total, a.k, oi.data_ptr(), od.data_ptr(), stream, prepared=prep)
        torch.cuda.synchronize()
        capi.profile_reset()
        capi.profile_enable(True)
        t = time.perf_counter()
        steps = 12
        for i in range(steps):
            qs, dfs, prep = sets[i % len(sets)]
            ps.bm25_search_batch_device(qs, dfs, a.docs, total, a.k, oi.data_ptr(), od.data_ptr(), stream, prepared=prep)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / steps
        capi.profile_enable(False)
        cnt, ms = capi.profile_get("bm25_score")
        capi.profile_reset()
        print("BM25 %d docs %d postings, batch %d x 3 terms, k=%d: %.3f ms/batch (%.1f us/query, %.0f q/s), score kernels %.3f ms; "
              "algorithmic %.1f MB/batch -> %.0f GB/s whole call, %.0f GB/s score kernels (%.3f of 8 TB/s)"
              % (a.docs, n_post, B, a.k, dt * 1e3, dt / B * 1e6, B / dt, ms / steps, byts / 1e6, byts / dt / 1e9,
                 byts / (ms / steps * 1e-3) / 1e9, byts / (ms / steps * 1e-3) / 8e12), flush=True)  # per BATCH (a large
        # batch runs as several chunks = several profile scopes: round 2's "2.9 us/query at 1024" divided by the scopes)


if __name__ == "__main__":
    main()
