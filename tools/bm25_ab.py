"""Round 5: BM25 scorer variants side by side in ONE process (the 10M-document corpus is built once):
    python tools/bm25_ab.py [--docs 10000000] [--batches 16,64,256,1024] [--check 6]
Every variant's first batch is compared with the default variant's results (ids and score bits) -- the variants are all exact."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myscaledb_amd.capi as capi  # noqa: E402
from bench import build_postings  # noqa: E402

VARIANTS = [
    ("r5 default (records, 8192 slots, radix cut, 8-ary bounds)", {}),
    ("round-4 scorer (bm25_rec=0, cutk=0, bounds8=0)", {"bm25_rec": "0", "bm25_cutk": "0", "bm25_bounds8": "0"}),
    ("records, 16384 slots", {"bm25_slots": "16384"}),
    ("records, list-merge cut", {"bm25_cutk": "0"}),
    ("records, binary bounds", {"bm25_bounds8": "0"}),
    ("bm25p + radix cut + 8-ary bounds (bm25_rec=0)", {"bm25_rec": "0"}),
    # ablations of the record scorer (results are WRONG by construction: timing only)
    ("ablation: windows generated, nothing loaded or scored (dbg=8)", {"bm25_dbg": "8"}),
    ("ablation: no shared-document filter (dbg=1)", {"bm25_dbg": "1"}),
    ("ablation: nothing leaves a window (dbg=4)", {"bm25_dbg": "4"}),
    ("ablation: no filter, no output (dbg=5)", {"bm25_dbg": "5"}),
    ("no second look at the flagged records (dbg=2; exact)", {"bm25_dbg": "2"}),
    ("general record scorer (bm25_lean=0)", {"bm25_lean": "0"}),
    ("ablation: filter runs, its flags are ignored (dbg=16)", {"bm25_dbg": "16"}),
    ("experiment: an empty launch in front of the bounds launch (dbg=32; exact)", {"bm25_dbg": "32"}),
    ("ranking select kernel (bm25_select2=0)", {"bm25_select2": "0"}),
    ("1.5 emit items per resident wavefront", {"bm25_items_per_wave": "1.5"}),
    ("four emit items per resident wavefront", {"bm25_items_per_wave": "4"}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--vocab", type=int, default=200_000)
    ap.add_argument("--batches", default="16,64,256,1024")
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--variants", default="")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    capi.set_device(0)
    ps, df_all, total, n_post = build_postings(a.docs, a.vocab)
    rng = np.random.default_rng(5)
    mids = np.argsort(-df_all)[50:2000]
    stream = torch.cuda.current_stream().cuda_stream
    sel = [int(v) for v in a.variants.split(",")] if a.variants else list(range(len(VARIANTS)))
    for B in [int(b) for b in a.batches.split(",")]:
        oi = torch.empty((B, a.k), device=dev, dtype=torch.int64)
        od = torch.empty((B, a.k), device=dev, dtype=torch.float32)
        sets = []
        for _ in range(4):
            qs = [rng.choice(mids, 3, replace=False) for _ in range(B)]
            dfs_ = [df_all[q] for q in qs]
            sets.append((qs, dfs_, ps.prepare_batch(qs, dfs_, total)))
        byts = np.mean([sum(int(d.sum()) * 8 + min(int(d.sum()), a.docs) for d in dfs) for _, dfs, _ in sets])
        ref = None
        for qs, dfs, prep in sets:  # scratch arenas grow to this batch size before any variant is timed
            ps.bm25_search_batch_device(qs, dfs, a.docs, total, a.k, oi.data_ptr(), od.data_ptr(), stream, prepared=prep)
        torch.cuda.synchronize()
        for vi in sel:
            name, opts = VARIANTS[vi]
            for k_, v_ in opts.items():
                capi.set_option(k_, v_)
            try:
                for qs, dfs, prep in sets[:2]:
                    ps.bm25_search_batch_device(qs, dfs, a.docs, total, a.k, oi.data_ptr(), od.data_ptr(), stream, prepared=prep)
                torch.cuda.synchronize()
                qs, dfs, prep = sets[0]
                ps.bm25_search_batch_device(qs, dfs, a.docs, total, a.k, oi.data_ptr(), od.data_ptr(), stream, prepared=prep)
                torch.cuda.synchronize()
                got = (oi.cpu().numpy().copy(), od.cpu().numpy().view(np.uint32).copy())
                if ref is None:
                    ref = got
                    eq = "reference"
                else:
                    eq = "== default" if (got[0] == ref[0]).all() and (got[1] == ref[1]).all() else \
                        "DIFFERS from default in %d of %d rows" % (int((got[0] != ref[0]).any(axis=1).sum()), B)
                capi.profile_reset()
                capi.profile_enable(True)
                steps = int(os.environ.get("STEPS", "0")) or (40 if B <= 256 else 24)
                t = time.perf_counter()
                for i in range(steps):
                    qs, dfs, prep = sets[i % len(sets)]
                    ps.bm25_search_batch_device(qs, dfs, a.docs, total, a.k, oi.data_ptr(), od.data_ptr(), stream, prepared=prep)
                host = (time.perf_counter() - t) / steps  # what the calls themselves took (the device runs behind)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t) / steps
                capi.profile_enable(False)
                cnt, ms = capi.profile_get("bm25_score")
                capi.profile_reset()
                q_, f_ = capi.bm25_stats()
                print("batch %5d  %-62s %.3f ms/batch  %.2f us/query  kernels %.3f ms  host %.3f ms  %.0f GB/s = %.4f of HBM  [%s; fallbacks so far %d]"
                      % (B, name, dt * 1e3, dt / B * 1e6, ms / steps, host * 1e3, byts / dt / 1e9, byts / dt / 8e12, eq, f_), flush=True)
            except Exception as e:
                print("batch %5d  %-62s FAILED: %r" % (B, name, e), flush=True)
            for k_ in opts:
                capi.set_option(k_, None)


if __name__ == "__main__":
    main()
