"""Randomised parity: seeded random shapes, data models, filters and entry points of the vector search (IVFFLAT, FLAT) and the
BM25 scorer through the C-ABI against the CPU oracle, bit for bit -- the fixed cases of test_gpu_parity.py aim at the branches the
kernels have, this module walks the space between them (dimensions that are no multiple of anything, k and nprobe at their limits,
lists shorter than k, duplicated rows, queries that ARE rows, filters that let nothing or everything through, batch sizes on
either side of every path switch).

MSVS_FUZZ_ITERS (default 16 per family: ~1 minute) sets how many configurations each test draws; MSVS_FUZZ_SEED moves the
sequence.  A failing configuration prints its seed and parameters: `MSVS_FUZZ_SEED=<seed> MSVS_FUZZ_ITERS=1` reproduces it."""
import os

import numpy as np
import pytest

import myscaledb_amd.capi as capi
from oracle import oracle as o

pytestmark = pytest.mark.gpu
ITERS = int(os.environ.get("MSVS_FUZZ_ITERS", "16"))
SEED = int(os.environ.get("MSVS_FUZZ_SEED", "20260930"))
OM = {capi.METRIC_L2: o.METRIC_L2, capi.METRIC_IP: o.METRIC_IP, capi.METRIC_COSINE: o.METRIC_COSINE}
MNAME = {capi.METRIC_L2: "L2", capi.METRIC_IP: "IP", capi.METRIC_COSINE: "COSINE"}


def loguni(rng, lo, hi):
    return int(round(float(np.exp(rng.uniform(np.log(lo), np.log(hi))))))


def make_rows(rng, n, d, nlist, model):
    """-> rows, query source.  Models: blobs (clustered, sigma drawn), iid, dup (a tenth of the rows copied many times: ties at every
    rank), tiny (blobs whose spread is 1e-3 of the centre distance: the fp16 shadow cannot separate rows), scaled (rows x 1e4 or 1e-4)."""
    centres = rng.standard_normal((max(nlist, 1), d), dtype=np.float32) * 2
    if model == "iid":
        x = rng.standard_normal((n, d), dtype=np.float32)
    else:
        sigma = {"tiny": 1e-3}.get(model, float(rng.choice([0.1, 0.3, 1.0])))
        x = (centres[rng.integers(0, len(centres), n)] + sigma * rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    if model == "dup":
        src = rng.integers(0, n, max(1, n // 10))
        x[rng.integers(0, n, n // 2)] = x[src[rng.integers(0, len(src), n // 2)]]
    if model == "scaled":
        x = (x * np.float32(rng.choice([1e4, 1e-4]))).astype(np.float32)
    return x, centres


def make_queries(rng, x, nq, d):
    kind = rng.choice(["near", "rows", "far"], p=[0.6, 0.3, 0.1])
    scale = np.float32(np.abs(x).mean() if x.size else 1.0)
    if kind == "rows":  # the query IS a row (distance 0, ties with its duplicates)
        return x[rng.integers(0, len(x), nq)].copy()
    if kind == "far":
        return (rng.standard_normal((nq, d), dtype=np.float32) * scale * np.float32(10)).astype(np.float32)
    return (x[rng.integers(0, len(x), nq)] + np.float32(0.3) * scale * rng.standard_normal((nq, d), dtype=np.float32)).astype(np.float32)


def make_alive(rng, n):
    mode = rng.choice(["none", "few", "half", "most", "nothing"], p=[0.45, 0.15, 0.2, 0.15, 0.05])
    if mode == "none":
        return None, mode
    p = {"few": 0.01, "half": 0.5, "most": 0.95, "nothing": 0.0}[mode]
    return rng.random(n) < p, mode


def oracle_ivf(ix, q, nprobe, k, metric, alive):
    cent, off, vecs, lids = ix.export()
    if metric == capi.METRIC_COSINE:
        oi, od, _ = o.ivf_search(cent, off, vecs, lids, o.normalize_rows(q), nprobe, k, o.METRIC_IP, alive=alive)
        return oi, (np.float32(1) - od).astype(np.float32)
    oi, od, _ = o.ivf_search(cent, off, vecs, lids, q, nprobe, k, OM[metric], alive=alive)
    return oi, od


def device_search(ix, q, k, nprobe, alive):
    import torch
    dev = torch.device("cuda", 0)
    dq = torch.from_numpy(np.ascontiguousarray(q)).to(dev)
    di = torch.empty((len(q), k), device=dev, dtype=torch.int64)
    dd = torch.empty((len(q), k), device=dev, dtype=torch.float32)
    bits, nbits = 0, 0
    if alive is not None:
        db = torch.from_numpy(capi.pack_bits(alive).view(np.int64)).to(dev)
        bits, nbits = db.data_ptr(), len(alive)
    torch.cuda.synchronize()
    ix.search_device(dq.data_ptr(), len(q), k, nprobe, di.data_ptr(), dd.data_ptr(), torch.cuda.current_stream().cuda_stream, d_alive=bits,
                     nbits=nbits)
    torch.cuda.synchronize()
    return di.cpu().numpy(), dd.cpu().numpy()


def check(tag, got, exp):
    (gi, gd), (ei, ed) = got, exp
    assert gi.shape == ei.shape, tag
    bad = np.argwhere(gi != ei)
    assert bad.size == 0, "%s: ids differ at %s (got %s, oracle %s)" % (tag, bad[:3].tolist(), gi[tuple(bad[0])], ei[tuple(bad[0])])
    assert (gd.view(np.uint32) == ed.view(np.uint32)).all(), "%s: distance bits differ" % tag


def test_fuzz_ivfflat_against_the_oracle():
    for it in range(ITERS):
        seed = SEED + it
        rng = np.random.default_rng(seed)
        metric = int(rng.choice([capi.METRIC_L2, capi.METRIC_IP, capi.METRIC_COSINE]))
        d = int(rng.choice([3, 8, 17, 32, 48, 64, 100, 128, 200, 256, 384, 768, 1000, 1536],
                           p=[.06, .06, .08, .1, .08, .12, .08, .12, .06, .06, .06, .06, .03, .03]))
        n = loguni(rng, 300, 60000 if d <= 256 else 20000)
        nlist = max(1, min(loguni(rng, 1, 1200), n // 4 if rng.random() < 0.85 else n))  # (sometimes more lists than a tenth of the rows: empty lists)
        model = str(rng.choice(["blobs", "iid", "dup", "tiny", "scaled"], p=[.45, .2, .15, .1, .1]))
        x, _ = make_rows(rng, n, d, nlist, model)
        ix = capi.Index(capi.INDEX_IVFFLAT, metric, d, "ncentroids=%d,kmeans_iters=3" % nlist)
        ix.train(x)
        cut = int(rng.integers(0, n + 1))  # two adds of any split (an empty one included)
        labels = None
        if rng.random() < 0.3:  # caller-supplied labels, not the row number
            labels = rng.permutation(n).astype(np.int64)
        if cut:
            ix.add(x[:cut], None if labels is None else labels[:cut])
        if cut < n:
            ix.add(x[cut:], None if labels is None else labels[cut:])
        ix.build()
        try:
            for rep in range(3):
                nq = int(rng.choice([1, 2, 3, 4, 5, 8, 15, 16, 17, 31, 33, 64, 100, 255, 256, 257, 700]))
                k = min(loguni(rng, 1, 128) if rng.random() < 0.85 else int(rng.choice([129, 200, 256])), 256)
                nprobe = min(loguni(rng, 1, max(1, nlist)), 256) if rng.random() < 0.9 else min(nlist + 3, 256)
                q = make_queries(rng, x, nq, d)
                alive, amode = make_alive(rng, n)
                tag = "seed %d rep %d: IVFFLAT %s n %d d %d nlist %d model %s labels %s | nq %d k %d nprobe %d filter %s" % (
                    seed, rep, MNAME[metric], n, d, nlist, model, labels is not None, nq, k, nprobe, amode)
                exp = oracle_ivf(ix, q, nprobe, k, metric, alive)
                check(tag + " [host]", ix.search(q, k, "nprobe=%d" % nprobe, alive=alive), exp)
                if rng.random() < 0.5 and alive is not None:  # the same filter as a resident delete bitmap
                    ix.set_delete_bitmap(alive)
                    check(tag + " [delete bitmap]", ix.search(q, k, "nprobe=%d" % nprobe), exp)
                    ix.set_delete_bitmap(None)
                if rng.random() < 0.4 and alive is not None:  # ... and as a filter object (bit test or compacted view by its count)
                    flt = capi.Filter.from_bool(alive)
                    check(tag + " [filter object]", ix.search_filter(q, k, "nprobe=%d" % nprobe, flt), exp)
                    flt.close()
                if rng.random() < 0.5:  # the device entry: device pointers, stream-ordered
                    check(tag + " [device]", device_search(ix, q, k, nprobe, alive), exp)
                if rng.random() < 0.15:  # k beyond one top-k pass: exact rounds of 256 (host entry only)
                    kk = int(rng.choice([257, 400, 1000]))
                    check(tag + " [k %d]" % kk, ix.search(q[:5], kk, "nprobe=%d" % nprobe, alive=alive), oracle_ivf(ix, q[:5], nprobe, kk, metric, alive))
            if rng.random() < 0.25:  # the index through its files and back: same structure, same answers
                store = {}
                ix.serialize_io(store)
                iy = capi.Index.load_io(store, capi.INDEX_IVFFLAT, metric, d)
                try:
                    check(tag + " [reloaded]", iy.search(q, k, "nprobe=%d" % nprobe, alive=alive), exp)
                finally:
                    iy.close()
        finally:
            ix.close()


def test_fuzz_flat_index_and_brute_force_against_the_oracle():
    for it in range(ITERS):
        seed = SEED + 100000 + it
        rng = np.random.default_rng(seed)
        metric = int(rng.choice([capi.METRIC_L2, capi.METRIC_IP, capi.METRIC_COSINE]))
        d = int(rng.choice([3, 16, 33, 64, 128, 300, 768, 1536]))
        n = loguni(rng, 50, 40000 if d <= 128 else 12000)
        model = str(rng.choice(["blobs", "iid", "dup", "tiny", "scaled"]))
        x, _ = make_rows(rng, n, d, 16, model)
        ix = capi.Index(capi.INDEX_FLAT, metric, d, "")
        ix.add(x)
        ix.build()
        try:
            for rep in range(3):
                nq = int(rng.choice([1, 2, 4, 7, 15, 16, 17, 40, 200, 300]))
                k = min(loguni(rng, 1, 100), 256)
                q = make_queries(rng, x, nq, d)
                alive, amode = make_alive(rng, n)
                tag = "seed %d rep %d: FLAT %s n %d d %d model %s | nq %d k %d filter %s" % (seed, rep, MNAME[metric], n, d, model, nq, k, amode)
                if metric == capi.METRIC_COSINE:
                    xi = ix.export()[2]  # the rows as the index keeps them (normalised)
                    ei, ed = o.knn(o.normalize_rows(q), xi, k, o.METRIC_IP, alive=alive)
                    ed = (np.float32(1) - ed).astype(np.float32)
                else:
                    ei, ed = o.knn(q, x, k, OM[metric], alive=alive)
                check(tag + " [index]", ix.search(q, k, "", alive=alive), (ei, ed))
                if metric != capi.METRIC_COSINE:  # seam A2: the block scan of the same rows
                    check(tag + " [knn]", capi.knn(q, x, k, metric, alive=alive), (ei, ed))
        finally:
            ix.close()


def test_fuzz_bm25_against_the_oracle():
    for it in range(max(1, ITERS // 2)):
        seed = SEED + 200000 + it
        rng = np.random.default_rng(seed)
        n_docs = loguni(rng, 200, 700_000)
        vocab = loguni(rng, 5, 3000)
        lists, total = [], 0
        for t in range(vocab):
            shape = rng.choice(["rare", "mid", "dense", "burst", "empty"], p=[.55, .25, .08, .07, .05])
            if total > 4_000_000 and shape in ("dense", "mid"):  # (keeps a configuration to a few seconds)
                shape = "rare"
            if shape == "empty":
                lists.append(np.zeros(0, np.uint32))
                continue
            cnt = {"rare": loguni(rng, 1, 50), "mid": loguni(rng, 50, max(51, n_docs // 50)), "dense": max(1, n_docs // int(rng.integers(2, 8))),
                   "burst": loguni(rng, 10, max(11, n_docs // 20))}[str(shape)]
            cnt = min(cnt, n_docs)
            if shape == "burst":
                lo = int(rng.integers(0, n_docs - cnt + 1))
                docs = np.arange(lo, lo + cnt, dtype=np.uint32)
            else:
                docs = np.sort(rng.choice(n_docs, cnt, replace=False)).astype(np.uint32)
            lists.append(docs)
            total += len(docs)
        post_off = np.zeros(vocab + 1, np.int64)
        np.cumsum([len(v) for v in lists], out=post_off[1:])
        doc = np.concatenate(lists) if post_off[-1] else np.zeros(0, np.uint32)
        tf = rng.integers(1, 9, len(doc)).astype(np.uint32)
        lens = np.maximum(1, rng.poisson(int(rng.choice([3, 20, 60])), n_docs))
        fn = np.array([o.fieldnorm_id(int(v)) for v in range(int(lens.max()) + 1)], np.uint8)[lens]
        total_tokens = int(lens.sum())
        ps = capi.Postings(post_off, doc, tf, fn)
        df_all = np.diff(post_off)
        try:
            for rep in range(2):
                nq = int(rng.choice([1, 3, 17, 64, 130]))
                queries = []
                for _ in range(nq):
                    nt = int(rng.choice([1, 2, 3, 4, 5, 9], p=[.2, .25, .25, .15, .1, .05]))
                    qt = [int(t) for t in rng.integers(0, vocab, nt)]  # (repeats allowed: a term twice counts twice)
                    queries.append(qt)
                dfs = [[int(df_all[t]) for t in qt] for qt in queries]
                k = loguni(rng, 1, 120)
                operator_or = bool(rng.random() < 0.8)
                alive, amode = make_alive(rng, n_docs)
                tag = "seed %d rep %d: BM25 docs %d vocab %d postings %d | nq %d k %d %s filter %s" % (
                    seed, rep, n_docs, vocab, len(doc), nq, k, "OR" if operator_or else "AND", amode)
                got = ps.bm25_search_batch(queries, dfs, n_docs, total_tokens, k, alive=alive, operator_or=operator_or)
                for qt, dfq, (gr, gs) in zip(queries, dfs, got):
                    er, es = o.bm25_search_ex(post_off, doc, tf, fn, qt, dfq, n_docs, total_tokens, k, alive=alive, operator_or=operator_or)
                    assert gr.tolist() == er.tolist(), "%s: rows differ for terms %s" % (tag, qt)
                    assert (gs.view(np.uint32) == es.view(np.uint32)).all(), "%s: score bits differ for terms %s" % (tag, qt)
        finally:
            ps.close()


def test_fuzz_binary_vectors_against_the_oracle():
    for it in range(max(1, ITERS // 2)):
        seed = SEED + 300000 + it
        rng = np.random.default_rng(seed)
        metric = int(rng.choice([capi.METRIC_HAMMING, capi.METRIC_JACCARD]))
        nbytes = int(rng.choice([1, 3, 8, 16, 17, 48, 96, 128, 200]))
        n = loguni(rng, 20, 50000)
        y = rng.integers(0, 256, (n, nbytes), dtype=np.uint8)
        if rng.random() < 0.3:
            y[rng.integers(0, n, n // 3)] = y[int(rng.integers(0, n))]  # many identical rows (ties by id); Jaccard of empty sets below
        if rng.random() < 0.3:
            y[rng.integers(0, n, max(1, n // 20))] = 0
        nq = int(rng.choice([1, 2, 9, 64, 130]))
        x = y[rng.integers(0, n, nq)].copy()
        flip = rng.random(x.shape) < 0.05
        x[flip] ^= rng.integers(1, 256, int(flip.sum()), dtype=np.uint8)
        k = loguni(rng, 1, 200)
        alive, amode = make_alive(rng, n)
        tag = "seed %d: binary %s n %d nbytes %d | nq %d k %d filter %s" % (seed, "HAMMING" if metric == capi.METRIC_HAMMING else "JACCARD", n, nbytes, nq, k, amode)
        om = o.METRIC_HAMMING if metric == capi.METRIC_HAMMING else o.METRIC_JACCARD
        exp = o.knn_bin(x, y, k, om, alive=alive)
        check(tag + " [knn_bin]", capi.knn_bin(x, y, k, metric, alive=alive), exp)
        bx = capi.BinIndex(nbytes, metric)
        try:
            cut = int(rng.integers(0, n + 1))
            if cut:
                bx.add(y[:cut])
            if cut < n:
                bx.add(y[cut:])
            check(tag + " [index]", bx.search(x, k, alive=alive), exp)
        finally:
            bx.close()


def test_fuzz_concurrent_mixed_searches_on_one_index():
    """Host threads searching ONE index at once with different shapes (single queries next to batches, with and without filters):
    per-thread streams and scratch, the combiner, the stale-by-one hints an index keeps between searches (plan feedback, stage-2
    grid hint) must never leak into another caller's answer."""
    import threading

    for it in range(max(1, ITERS // 8)):
        seed = SEED + 400000 + it
        rng = np.random.default_rng(seed)
        metric = int(rng.choice([capi.METRIC_L2, capi.METRIC_IP, capi.METRIC_COSINE]))
        d = int(rng.choice([32, 96, 128, 768]))
        n, nlist = 40000 if d <= 128 else 15000, int(rng.choice([32, 200, 600]))
        x, _ = make_rows(rng, n, d, nlist, str(rng.choice(["blobs", "iid", "dup"])))
        ix = capi.Index(capi.INDEX_IVFFLAT, metric, d, "ncentroids=%d,kmeans_iters=3" % nlist)
        ix.train(x)
        ix.add(x)
        ix.build()
        jobs = []
        for j in range(24):
            nq = int(rng.choice([1, 1, 1, 2, 4, 16, 40, 300]))
            k = int(rng.choice([1, 10, 10, 40, 100]))
            nprobe = int(rng.choice([1, 4, 16, min(32, nlist)]))
            q = make_queries(rng, x, nq, d)
            alive = (rng.random(n) < 0.3) if rng.random() < 0.25 else None
            jobs.append((q, k, nprobe, alive, oracle_ivf(ix, q, nprobe, k, metric, alive)))
        errors = []

        def worker(t):
            order = np.random.default_rng(seed * 100 + t).permutation(len(jobs))
            try:
                for rep in range(3):
                    for j in order:
                        q, k, nprobe, alive, exp = jobs[j]
                        check("seed %d thread %d job %d (nq %d k %d nprobe %d filter %s)" % (seed, t, j, len(q), k, nprobe, alive is not None),
                              ix.search(q, k, "nprobe=%d" % nprobe, alive=alive), exp)
            except BaseException as e:  # noqa: BLE001 -- reported by the main thread
                errors.append(e)

        threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        ix.close()
        assert not errors, errors[0]
