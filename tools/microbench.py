#!/usr/bin/env python3
"""Secondary measurements of the other BASELINE.json configs (not the headline bench.py line): device-resident,
HIP-event-free wall timing with stream sync, printed as plain text for profiles/.

  C1  FLAT brute force 10k x 128 L2 top-10 (resident FLAT index; the A2 host-pointer call is PCIe-bound by design)
  C2' FLAT exhaustive 1M x 768 L2 (the exact scan the IVF recall is measured against)
  C3  partition scan ("MSTG-style"): IVFFLAT cosine, 10M x 768 (or --rows), nlist 4096, batch 64, nprobe 32
  C5  BM25 over 10M documents (Zipf(1.1) vocabulary of 200k terms, Poisson(30) lengths), 3-term queries
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import myscaledb_amd.capi as capi  # noqa: E402
from bench import build_postings, make_data, make_queries  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


def flat_case(name, n, d, nqs, k, dev):
    model, x = make_data(n, d, 1234, dev)
    ix = capi.Index(capi.INDEX_FLAT, capi.METRIC_L2, d)
    ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    ix.build()
    stream = torch.cuda.current_stream().cuda_stream
    for nq in nqs:
        q = make_queries(model, nq, 4321, dev)
        oi = torch.empty((nq, k), device=dev, dtype=torch.int64)
        od = torch.empty((nq, k), device=dev, dtype=torch.float32)
        f0 = capi.prefilter_stats()
        dt = timed(lambda: ix.search_device(q.data_ptr(), nq, k, 0, oi.data_ptr(), od.data_ptr(), stream), 20)
        f1 = capi.prefilter_stats()
        # passes over the rows: one per 128-query tile through the candidate pass (nq >= 16 on a table of >= 128
        # tile x slice work items), else one per 8-query tile of the canonical scan
        cand = nq >= 16 and -(-nq // 128) * -(-n // 128) >= 128
        passes = -(-nq // 128) if cand else (-(-nq // 8) if nq > 4 else 1)
        print("%s FLAT %dx%d nq=%d k=%d : %.3f ms/call  %.0f QPS  streamed %.2f GB -> %.0f GB/s  (per-query model %.0f GB/s)"
              "  candidate pass (queries, fallbacks) = (%d, %d)"
              % (name, n, d, nq, k, dt * 1e3, nq / dt, passes * n * d * 4 / 1e9, passes * n * d * 4 / dt / 1e9,
                 nq * n * d * 4 / dt / 1e9, f1[0] - f0[0], f1[1] - f0[1]), flush=True)
    ix.close()
    del x


def ivf_cosine_case(n, d, nlist, batch, nprobe, k, dev, metric=capi.METRIC_COSINE, name="C3", blobs=1024):
    model, x = make_data(n, d, 1234, dev, blobs=blobs)
    ix = capi.Index(capi.INDEX_IVFFLAT, metric, d, "ncentroids=%d,kmeans_iters=8,train_sample=%d" % (nlist, nlist * 48))
    t0 = time.time()
    ix.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    step = 2_000_000
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        ix.add(x[lo:hi].data_ptr(), n=hi - lo, mem=capi.MEM_DEVICE)
    ix.build()
    torch.cuda.synchronize()
    build_s = time.time() - t0
    del x
    q = make_queries(model, batch * 8, 4321, dev)
    stream = torch.cuda.current_stream().cuda_stream
    oi = torch.empty((batch, k), device=dev, dtype=torch.int64)
    od = torch.empty((batch, k), device=dev, dtype=torch.float32)
    it = [0]

    def run():
        b = it[0] % 8
        it[0] += 1
        ix.search_device(q[b * batch:(b + 1) * batch].data_ptr(), batch, k, nprobe, oi.data_ptr(), od.data_ptr(), stream)
    dt = timed(run, 20)
    rows, streamed, uniq = ix.scanned_rows(q[:batch].cpu().numpy(), nprobe)
    rb = 4 * d + 4
    f0 = capi.prefilter_stats()
    run()
    torch.cuda.synchronize()
    f1 = capi.prefilter_stats()
    print("%s IVFFLAT %s %dx%d nlist=%d batch=%d nprobe=%d k=%d : build %.1f s, %.3f ms/batch  %.0f QPS ; rows/query %.0f ; "
          "union %.2f GB -> %.0f GB/s, per-query model %.2f GB ; candidate pass (queries, fallbacks) per batch = (%d, %d)"
          % (name, {capi.METRIC_COSINE: "cosine", capi.METRIC_IP: "IP"}.get(metric, "L2"), n, d, nlist, batch, nprobe, k, build_s, dt * 1e3,
             batch / dt, rows / batch, uniq * rb / 1e9, uniq * rb / dt / 1e9, rows * rb / 1e9, f1[0] - f0[0],
             f1[1] - f0[1]), flush=True)
    ix.close()


def bm25_case(n_docs, vocab, k):
    rng = np.random.default_rng(5)
    ps, df_all, total, n_post = build_postings(n_docs, vocab)
    mids = np.argsort(-df_all)[50:2000]
    times, bytes_ = [], []
    for i in range(30):
        q = rng.choice(mids, 3, replace=False)
        df = df_all[q]
        t = time.perf_counter()
        rows, scores = ps.bm25_search(q, df, n_docs, total, k)
        times.append(time.perf_counter() - t)
        bytes_.append(int(df.sum()) * 8 + min(int(df.sum()), n_docs))
    times, bytes_ = np.array(times[5:]), np.array(bytes_[5:])
    print("C5 BM25 %d docs, %d postings, 3-term queries, k=%d : p50 %.3f ms/query (host call incl. D2H), "
          "postings+fieldnorm bytes/query %.1f MB -> %.0f GB/s"
          % (n_docs, n_post, k, np.median(times) * 1e3, bytes_.mean() / 1e6, (bytes_ / times).mean() / 1e9), flush=True)


def hybrid_case(n, d, nlist, nprobe, vocab, dev):
    """BASELINE config 5: per query, vector top-100 (IVFFLAT cosine, the partition scan) + BM25 top-100 over the same n
    rows + reciprocal-rank fusion on the host (libmsvs_host.so) -> top-10; host-pointer calls, wall time per query."""
    import myscaledb_amd.host as host
    model, x = make_data(n, d, 1234, dev)
    ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_COSINE, d, "ncentroids=%d,kmeans_iters=8,train_sample=%d" % (nlist, nlist * 48))
    ix.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    for lo in range(0, n, 2_000_000):
        hi = min(n, lo + 2_000_000)
        ix.add(x[lo:hi].data_ptr(), n=hi - lo, mem=capi.MEM_DEVICE)
    ix.build()
    del x
    ps, df_all, total, _ = build_postings(n, vocab)
    rng = np.random.default_rng(6)
    mids = np.argsort(-df_all)[50:2000]
    qs = make_queries(model, 40, 4321, dev).cpu().numpy()
    tv, tb, tf_, tt = [], [], [], []
    z = lambda m: np.zeros(m, np.uint64)
    for i in range(40):
        qt = rng.choice(mids, 3, replace=False)
        t0 = time.perf_counter()
        vi, vd = ix.search(qs[i:i + 1], 100, "nprobe=%d" % nprobe)
        t1 = time.perf_counter()
        rows, scores = ps.bm25_search(qt, df_all[qt], n, total, 100)
        t2 = time.perf_counter()
        host.hybrid_search("rrf", (vd[0], z(100), vi[0].astype(np.uint64)), (scores, z(len(rows)), rows), 10, fusion_k=60)
        t3 = time.perf_counter()
        tv.append(t1 - t0), tb.append(t2 - t1), tf_.append(t3 - t2), tt.append(t3 - t0)
    med = lambda a: float(np.median(a[8:])) * 1e3
    print("C5h hybrid %d rows x %d (IVFFLAT cosine nlist=%d nprobe=%d top-100 + BM25 top-100 + RRF top-10), host calls per "
          "query: p50 total %.3f ms = vector %.3f + bm25 %.3f + fusion %.3f" % (n, d, nlist, nprobe, med(tt), med(tv), med(tb), med(tf_)),
          flush=True)
    ix.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--skip", default="")
    ap.add_argument("--big-rows", type=int, default=0,
                    help="C4s: IVFFLAT L2 on this many rows x 768 (nlist = rows / 2048, nprobe 64, 4096 queries per batch); "
                         "needs ~3 x rows x 3 KB of HBM during the build")
    ap.add_argument("--c4-rows", type=int, default=0,
                    help="C4 shard shape: IVFFLAT inner product on this many rows x 1536 (nlist = rows / 1536, nprobe 64), batches "
                         "of 64 / 1024 / 4096 -- one GPU's share of BASELINE config 4 is 12.5M rows")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    capi.set_device(0)
    if "c1" not in a.skip:
        flat_case("C1", 10_000, 128, [1, 8, 64, 1000], 10, dev)
    if "c2" not in a.skip:
        flat_case("C2'", 1_000_000, 768, [1, 8, 64, 1024], 10, dev)
    if "c3" not in a.skip:
        ivf_cosine_case(a.rows, 768, 4096, 64, 32, 10, dev)
    if a.big_rows:
        nl = max(1024, a.big_rows // 2048)
        ivf_cosine_case(a.big_rows, 768, nl, 4096, 64, 10, dev, metric=capi.METRIC_L2, name="C4s", blobs=nl)
    if a.c4_rows:
        nl = max(1024, a.c4_rows // 1536)
        for b in (4096, 1024, 64):
            ivf_cosine_case(a.c4_rows, 1536, nl, b, 64, 10, dev, metric=capi.METRIC_IP, name="C4-shard", blobs=nl)
    if "c5h" not in a.skip and "c5" not in a.skip:
        hybrid_case(a.rows, 768, 4096, 32, 200_000, dev)
    if "c5" not in a.skip:
        bm25_case(a.docs, 200_000, 100)


if __name__ == "__main__":
    main()
