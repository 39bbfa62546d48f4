"""N > 1 path on CPU: two processes, gloo backend.  The per-rank partial results come from the oracle restricted to
the rank's lists (list_id % 2 == rank) -- the GPU scan itself is covered by the -m gpu tests -- and go through the
product's exchange (all-gather) + host merge; the result must equal the unsharded search (the reference's own
"distributed == single MergeTree" assertion, tests/integration/test_mqvs_distributed_hybrid_search/test.py:109-121)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from golden_util import TextIndex, load_goldens, tokenize
from oracle import oracle as o


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_case():
    rng = np.random.default_rng(77)
    n, d, nlist, nq, k, nprobe = 6000, 24, 16, 9, 10, 6
    x = rng.standard_normal((n, d), dtype=np.float32)
    x[100:140] = x[7]  # duplicates: ties must merge deterministically
    q = np.concatenate([rng.standard_normal((nq - 1, d), dtype=np.float32), x[7:8]])
    ids = np.arange(n, dtype=np.int64) * 2 + 1
    cent = o.kmeans(x, nlist, 4)
    off, vecs, lids = o.build_ivf(x, ids, cent)
    return cent, off, vecs, lids, q, k, nprobe


def _shard(off, vecs, lids, rank, world):
    """keep only lists with list_id % world == rank (the library's sharding rule)"""
    keep = np.zeros(len(lids), bool)
    new_off = np.zeros_like(off)
    for l in range(len(off) - 1):
        if l % world == rank:
            keep[off[l]:off[l + 1]] = True
        new_off[l + 1] = new_off[l] + (off[l + 1] - off[l] if l % world == rank else 0)
    return new_off, vecs[keep], lids[keep]


def _worker(rank, world, port, metric, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from myscaledb_amd import sharded
        cent, off, vecs, lids, q, k, nprobe = _make_case()
        soff, svecs, slids = _shard(off, vecs, lids, rank, world)
        li, ld, _ = o.ivf_search(cent, soff, svecs, slids, q, nprobe, k, metric)
        mi, md = sharded.exchange_and_merge(torch.from_numpy(li), torch.from_numpy(ld), metric)
        # BM25 statistics exchange: every rank owns half of the documents
        docs = [d["texts"] for d in load_goldens()["00041_two_parts"]["docs"]]
        mine = docs[rank * 10:(rank + 1) * 10]
        idx = TextIndex(mine, o.fieldnorm_id)
        terms = tokenize("Ancient")
        n, tok, df = sharded.all_reduce_bm25_stats(idx.num_docs, idx.total_tokens, [idx.doc_freq(t) for t in terms])
        qt = [idx.vocab[t] for t in terms if t in idx.vocab]
        dfq = [f for t, f in zip(terms, df) if t in idx.vocab]
        rows, scores = o.bm25_search(idx.post_off, idx.doc_ids, idx.tfs, idx.fieldnorm_ids, qt, dfq, n, tok, 5)
        out.put((rank, mi.numpy(), md.numpy(), (n, tok, df), (rows + rank * 10).tolist(), scores.tolist()))
    finally:
        dist.destroy_process_group()


def _run(metric):
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, metric, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_sharded_l2_equals_unsharded():
    res = _run(o.METRIC_L2)
    cent, off, vecs, lids, q, k, nprobe = _make_case()
    fi, fd, _ = o.ivf_search(cent, off, vecs, lids, q, nprobe, k, o.METRIC_L2)
    for _, mi, md, _, _, _ in res:
        assert (mi == fi).all() and (md.view(np.uint32) == fd.view(np.uint32)).all()
    # BM25: summed statistics == whole-table statistics; global-stat scores reproduce the 00041 golden
    g = load_goldens()["00041_two_parts"]
    whole = TextIndex([d["texts"] for d in g["docs"]], o.fieldnorm_id)
    assert res[0][3] == res[1][3] == (whole.num_docs, whole.total_tokens, [whole.doc_freq("ancient")])
    hits = sorted([(s, r) for _, _, _, _, rows, sc in res for r, s in zip(rows, sc)], key=lambda t: -t[0])
    assert [h[1] for h in hits] == g["text_search_2parts"][0]
    assert [np.float32(h[0]) for h in hits] == [np.float32(s) for s in g["text_search_2parts"][1]]


def test_sharded_ip_equals_unsharded():
    res = _run(o.METRIC_IP)
    cent, off, vecs, lids, q, k, nprobe = _make_case()
    fi, fd, _ = o.ivf_search(cent, off, vecs, lids, q, nprobe, k, o.METRIC_IP)
    for _, mi, md, _, _, _ in res:
        assert (mi == fi).all() and (md.view(np.uint32) == fd.view(np.uint32)).all()


# ---------------------------------------------------------------------------------------- the product's sharded search

def _gpu_worker(rank, world, port, metric_name, typ, out):
    """Two processes on ONE GPU, each holding its shard of a libmsvs index, through msvs_shard_search_device with the
    all-gather carried by gloo (the RCCL transport needs one GPU per rank; everything else is the product path:
    coarse quantiser sharded by query, probe exchange, local scan, packed exchange, strided merge)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import myscaledb_amd.capi as capi
        from myscaledb_amd import sharded
        capi.set_device(0)
        metric = {"L2": capi.METRIC_L2, "IP": capi.METRIC_IP, "cosine": capi.METRIC_COSINE}[metric_name]
        rng = np.random.default_rng(2024)
        n, d, nlist, k, nprobe = 20000, 40, 24, 10, 7
        x = rng.standard_normal((n, d), dtype=np.float32)
        x[500:540] = x[3]  # duplicates across lists' members: ties must merge deterministically
        cent = o.kmeans(x, nlist, 3)
        ix = capi.Index(typ, metric, d, "ncentroids=%d,shard_rank=%d,shard_world=%d" % (nlist, rank, world))
        if typ == capi.INDEX_IVFFLAT:
            ix.set_centroids(cent)
        ix.add(x)
        ix.build()
        comm = sharded.gloo_comm()
        res = {}
        for nq in (5, 64, 700):  # per-query kernels, small batch, candidate pass; 5 and 700 are not multiples of the world
            q = rng.standard_normal((nq, d), dtype=np.float32)
            q[0] = x[3]
            dq = torch.from_numpy(q).cuda()
            oi = torch.empty((nq, k), dtype=torch.int64, device="cuda")
            od = torch.empty((nq, k), dtype=torch.float32, device="cuda")
            ix.shard_search_device(comm, dq.data_ptr(), nq, k, nprobe, oi.data_ptr(), od.data_ptr())
            torch.cuda.synchronize()
            res[nq] = (q, oi.cpu().numpy(), od.cpu().numpy())
        # the statistics exchange of a sharded BM25 search through the C-ABI (msvs_host_all_reduce_bm25_stats ->
        # msvs_comm_all_reduce_u64 on the same communicator)
        res["stats"] = sharded.all_reduce_bm25_stats(1000 + rank, 30000 + 7 * rank, [5 + rank, 0, 2 ** 40 + rank], comm=comm)
        out.put((rank, res))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _rccl_worker(rank, world, port, out):
    """One process per GPU, RCCL communicator owned by libmsvs (the production transport): sharded search == unsharded,
    statistics all-reduce."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        import myscaledb_amd.capi as capi
        from myscaledb_amd import sharded
        capi.set_device(rank)
        comm = sharded.rccl_comm()
        rng = np.random.default_rng(2025)
        n, d, nlist, k, nprobe = 30000, 64, 32, 10, 6
        x = rng.standard_normal((n, d), dtype=np.float32)
        cent = o.kmeans(x, nlist, 3)
        ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, "ncentroids=%d,shard_rank=%d,shard_world=%d" % (nlist, rank, world))
        ix.set_centroids(cent)
        ix.add(x)
        ix.build()
        res = {}
        stream = torch.cuda.current_stream().cuda_stream
        for nq in (3, 600):
            q = rng.standard_normal((nq, d), dtype=np.float32)
            dq = torch.from_numpy(q).cuda()
            oi = torch.empty((nq, k), dtype=torch.int64, device="cuda")
            od = torch.empty((nq, k), dtype=torch.float32, device="cuda")
            ix.shard_search_device(comm, dq.data_ptr(), nq, k, nprobe, oi.data_ptr(), od.data_ptr(), stream)
            torch.cuda.synchronize()
            res[nq] = (q, oi.cpu().numpy(), od.cpu().numpy())
        res["stats"] = sharded.all_reduce_bm25_stats(10 + rank, 100 * (rank + 1), [rank, 7], comm=comm)
        out.put((rank, res))
        dist.barrier()
        comm.close()
    finally:
        dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.gpu
def test_rccl_two_ranks_sharded_search_and_statistics():
    """The production transport with more than one rank: runs whenever the box shows at least two GPUs (the round-end driver's
    multi-GPU node), skipped on a one-GPU box."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import myscaledb_amd.capi as capi
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    rng = np.random.default_rng(2025)
    n, d, nlist, k, nprobe = 30000, 64, 32, 10, 6
    x = rng.standard_normal((n, d), dtype=np.float32)
    cent = o.kmeans(x, nlist, 3)
    capi.set_device(0)
    ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, "ncentroids=%d" % nlist)
    ix.set_centroids(cent)
    ix.add(x)
    ix.build()
    for nq in (3, 600):
        q = res[0][1][nq][0]
        fi, fd = ix.search(q, k, "nprobe=%d" % nprobe)
        for r in range(world):
            assert (res[r][1][nq][1] == fi).all() and (res[r][1][nq][2].view(np.uint32) == fd.view(np.uint32)).all()
    assert res[0][1]["stats"] == res[1][1]["stats"] == (21, 300, [1, 14])


@pytest.mark.gpu
@pytest.mark.parametrize("metric_name,typ", [("L2", 1), ("cosine", 1), ("IP", 0)])
def test_product_sharded_search_two_processes_equals_unsharded(metric_name, typ):
    import myscaledb_amd.capi as capi
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, metric_name, typ, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # the unsharded index in this process, same data / centroids
    metric = {"L2": capi.METRIC_L2, "IP": capi.METRIC_IP, "cosine": capi.METRIC_COSINE}[metric_name]
    rng = np.random.default_rng(2024)
    n, d, nlist, k, nprobe = 20000, 40, 24, 10, 7
    x = rng.standard_normal((n, d), dtype=np.float32)
    x[500:540] = x[3]
    cent = o.kmeans(x, nlist, 3)
    ix = capi.Index(typ, metric, d, "ncentroids=%d" % nlist)
    if typ == capi.INDEX_IVFFLAT:
        ix.set_centroids(cent)
    ix.add(x)
    ix.build()
    for nq in (5, 64, 700):
        q = res[0][1][nq][0]
        assert (q == res[1][1][nq][0]).all()
        fi, fd = ix.search(q, k, "nprobe=%d" % nprobe if typ == capi.INDEX_IVFFLAT else "")
        for r in range(world):
            assert (res[r][1][nq][1] == fi).all()
            assert (res[r][1][nq][2].view(np.uint32) == fd.view(np.uint32)).all()
    assert res[0][1]["stats"] == res[1][1]["stats"] == (2001, 60007, [11, 0, 2 ** 41 + 1])


def _pruned_case(geometry="blobs"):
    rng = np.random.default_rng(77)
    n, d, nlist, nq = 60000, 64, 256, 800  # (the centroid-shadow coarse pass -- the source of the probe words -- serves nlist >= 256)
    centres = 4.0 * rng.standard_normal((nlist, d), dtype=np.float32)
    x = (centres[rng.integers(0, nlist, n)] + rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    q = (centres[rng.integers(0, nlist, nq)] + rng.standard_normal((nq, d), dtype=np.float32)).astype(np.float32)
    if geometry == "outlier_neighbour":
        # adversarial for the radius bounds: one far row per list sets its radius and IS the nearest row of a query next to it; ties
        # across the k-th rank; lists of exactly k rows do not exist here, zero rows do (cosine keeps them unnormalised)
        u = rng.standard_normal((nlist, d)).astype(np.float32)
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        far = (centres + 18.0 * u).astype(np.float32)
        x = np.concatenate([x, far, np.repeat(x[7:8], 6, axis=0), np.zeros((5, d), np.float32)]).astype(np.float32)
        q[:nlist] = far + 0.05 * rng.standard_normal((nlist, d)).astype(np.float32)
        q[nlist] = x[7]
        q[nlist + 1:nlist + 9] = centres[:8]
    return x, q, centres, nlist, 10, 8


def _pruned_worker(rank, world, port, metric_name, out, geometry="blobs"):
    """The sharded search with the probe pruning ON on every rank: the coarse pass of a query runs on ONE rank, its distance word
    per probe travels with the probe lists (ProbeWords), and every rank prunes the pairs of its own lists."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import myscaledb_amd.capi as capi
        from myscaledb_amd import sharded
        capi.set_device(0)
        metric = {"L2": capi.METRIC_L2, "IP": capi.METRIC_IP, "cosine": capi.METRIC_COSINE}[metric_name]
        x, q, centres, nlist, k, nprobe = _pruned_case(geometry)
        ix = capi.Index(capi.INDEX_IVFFLAT, metric, x.shape[1], "ncentroids=%d,shard_rank=%d,shard_world=%d" % (nlist, rank, world))
        ix.set_centroids(centres)
        ix.add(x)
        ix.build()
        comm = sharded.gloo_comm()
        capi.set_option("h16_prune", "2")
        capi.set_option("rerank_stats", "1")
        capi.set_option("ivf_pass", "2")
        nq = q.shape[0]
        dq = torch.from_numpy(q).cuda()
        oi = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        od = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        s0 = capi.debug_prune_stats()
        ix.shard_search_device(comm, dq.data_ptr(), nq, k, nprobe, oi.data_ptr(), od.data_ptr())
        torch.cuda.synchronize()
        s1 = capi.debug_prune_stats()
        # two batches in flight (msvs_shard_search_device_async): four batches of different sizes back to back, each with its own
        # result buffers, one drain at the end -- the exchange + merge of batch i run beside the scan of batch i + 1
        cuts = [(0, 300), (300, 301), (301, 800), (0, 800)]
        outs = [(torch.empty((hi - lo, k), dtype=torch.int64, device="cuda"), torch.empty((hi - lo, k), dtype=torch.float32, device="cuda"))
                for lo, hi in cuts]
        for (lo, hi), (bi, bd) in zip(cuts, outs):
            ix.shard_search_device_async(comm, dq[lo:hi].data_ptr(), hi - lo, k, nprobe, bi.data_ptr(), bd.data_ptr())
        comm.drain()
        torch.cuda.synchronize()
        piped = [(bi.cpu().numpy(), bd.cpu().numpy()) for bi, bd in outs]
        out.put((rank, oi.cpu().numpy(), od.cpu().numpy(), (s1[0] - s0[0], s1[1] - s0[1]), piped))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("metric_name,geometry", [("L2", "blobs"), ("cosine", "blobs"), ("IP", "blobs"),
                                                  ("L2", "outlier_neighbour"), ("cosine", "outlier_neighbour")])
def test_product_sharded_search_prunes_on_every_rank_and_equals_unsharded(metric_name, geometry):
    import myscaledb_amd.capi as capi
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pruned_worker, args=(r, world, port, metric_name, out, geometry)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    metric = {"L2": capi.METRIC_L2, "IP": capi.METRIC_IP, "cosine": capi.METRIC_COSINE}[metric_name]
    x, q, centres, nlist, k, nprobe = _pruned_case(geometry)
    ix = capi.Index(capi.INDEX_IVFFLAT, metric, x.shape[1], "ncentroids=%d" % nlist)
    ix.set_centroids(centres)
    ix.add(x)
    ix.build()
    capi.set_option("h16_prune", "0")
    try:
        fi, fd = ix.search(q, k, "nprobe=%d" % nprobe)
    finally:
        capi.set_option("h16_prune", None)
    for r in range(world):
        assert (res[r][1] == fi).all() and (res[r][2].view(np.uint32) == fd.view(np.uint32)).all()
        dropped, looked = res[r][3]
        assert looked == q.shape[0] * nprobe, "rank %d: the pruning did not look at the batch" % r
        if metric_name == "L2":
            assert dropped > 0.1 * looked, (r, dropped, looked)  # well separated blobs (the bound comes from the rank's OWN sample rows: looser than unsharded)
        for (lo, hi), (bi, bd) in zip([(0, 300), (300, 301), (301, 800), (0, 800)], res[r][4]):
            assert (bi == fi[lo:hi]).all() and (bd.view(np.uint32) == fd[lo:hi].view(np.uint32)).all(), (r, lo, hi)


def _routed_worker(rank, world, port, metric_name, geometry, out):
    """The routed sharded search: every rank brings its OWN queries (different counts, one rank with an odd count, an empty batch),
    gets its own results back, and reports how many (query, rank) pairs it served."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import myscaledb_amd.capi as capi
        from myscaledb_amd import sharded
        capi.set_device(0)
        metric = {"L2": capi.METRIC_L2, "IP": capi.METRIC_IP, "cosine": capi.METRIC_COSINE}[metric_name]
        x, q, centres, nlist, k, nprobe = _pruned_case(geometry)
        ix = capi.Index(capi.INDEX_IVFFLAT, metric, x.shape[1], "ncentroids=%d,shard_rank=%d,shard_world=%d" % (nlist, rank, world))
        ix.set_centroids(centres)
        ix.add(x)
        ix.build()
        comm = sharded.gloo_comm()
        capi.set_option("h16_prune", "2")
        capi.set_option("ivf_pass", "2")
        res = []
        # (step, this rank's slice of q): rank 0 takes the even queries, rank 1 the odd ones; then unequal shares; then one rank idle
        for lo, hi, mine in ((0, 800, lambda i: i % world == rank), (0, 301, lambda i: (i < 77) == (rank == 0)), (300, 340, lambda i: rank == 1)):
            sel = np.array([i for i in range(lo, hi) if mine(i)], np.int64)
            nq = len(sel)
            dq = torch.from_numpy(q[sel]).cuda() if nq else None
            oi = torch.empty((max(nq, 1), k), dtype=torch.int64, device="cuda")
            od = torch.empty((max(nq, 1), k), dtype=torch.float32, device="cuda")
            served = ix.shard_search_routed_device(comm, dq.data_ptr() if nq else 0, nq, k, nprobe, oi.data_ptr(), od.data_ptr())
            torch.cuda.synchronize()
            res.append((sel, oi.cpu().numpy()[:nq], od.cpu().numpy()[:nq], served))
        out.put((rank, res))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("metric_name,geometry", [("L2", "blobs"), ("cosine", "blobs"), ("IP", "blobs"), ("L2", "outlier_neighbour")])
def test_routed_sharded_search_two_processes_equals_unsharded(metric_name, geometry):
    """msvs_shard_search_routed_device over two processes (gloo transport: the point-to-point exchange emulated by its all-gather):
    every rank's own queries come back with the unsharded index's ids and distance bits; with the pre-pruning at work (L2, cosine: well
    separated blobs) a query visits fewer than all ranks; the inner-product index routes every query everywhere and is still right."""
    import myscaledb_amd.capi as capi
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_routed_worker, args=(r, world, port, metric_name, geometry, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    metric = {"L2": capi.METRIC_L2, "IP": capi.METRIC_IP, "cosine": capi.METRIC_COSINE}[metric_name]
    x, q, centres, nlist, k, nprobe = _pruned_case(geometry)
    ix = capi.Index(capi.INDEX_IVFFLAT, metric, x.shape[1], "ncentroids=%d" % nlist)
    ix.set_centroids(centres)
    ix.add(x)
    ix.build()
    capi.set_option("h16_prune", "0")
    try:
        fi, fd = ix.search(q, k, "nprobe=%d" % nprobe)
    finally:
        capi.set_option("h16_prune", None)
    for step in range(3):
        total_queries = sum(len(res[r][1][step][0]) for r in range(world))
        served = sum(res[r][1][step][3] for r in range(world))
        for r in range(world):
            sel, gi, gd, _ = res[r][1][step]
            assert (gi == fi[sel]).all(), (r, step, np.argwhere(gi != fi[sel])[:4])
            assert (gd.view(np.uint32) == fd[sel].view(np.uint32)).all(), (r, step)
        assert total_queries <= served <= world * total_queries, (step, served, total_queries)
        if metric_name == "L2" and geometry == "blobs" and step == 0:
            assert served < 1.7 * total_queries, "well separated blobs: most queries need one rank (%d pairs for %d queries)" % (served, total_queries)


@pytest.mark.gpu
def test_rccl_transport_single_rank_roundtrip():
    """The RCCL code path itself (dlopen'd librccl: ncclGetUniqueId, ncclCommInitRank, in-place ncclAllGather on the
    search stream) with the one rank a 1-GPU box can host: the sharded search through it == the plain search."""
    import myscaledb_amd.capi as capi
    capi.set_device(0)
    comm = capi.Comm(1, 0, id=capi.comm_unique_id())
    rng = np.random.default_rng(5)
    n, d, k = 8000, 32, 10
    x = rng.standard_normal((n, d), dtype=np.float32)
    for typ, params in ((capi.INDEX_IVFFLAT, "ncentroids=16"), (capi.INDEX_FLAT, "")):
        ix = capi.Index(typ, capi.METRIC_L2, d, params)
        if typ == capi.INDEX_IVFFLAT:
            ix.train(x)
        ix.add(x)
        ix.build()
        for nq in (3, 200):
            q = rng.standard_normal((nq, d), dtype=np.float32)
            dq = torch.from_numpy(q).cuda()
            oi = torch.empty((nq, k), dtype=torch.int64, device="cuda")
            od = torch.empty((nq, k), dtype=torch.float32, device="cuda")
            ix.shard_search_device(comm, dq.data_ptr(), nq, k, 5, oi.data_ptr(), od.data_ptr())
            torch.cuda.synchronize()
            fi, fd = ix.search(q, k, "nprobe=5" if typ == capi.INDEX_IVFFLAT else "")
            assert (oi.cpu().numpy() == fi).all() and (od.cpu().numpy() == fd).all()
            if typ != capi.INDEX_IVFFLAT:
                continue
            # the ROUTED form through the real ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd path: with route_self_rccl the
            # rank's own piece travels to itself through the group instead of a device copy (both exchanges of the step)
            capi.set_option("route_self_rccl", "1")
            try:
                oi.fill_(-7)
                served = ix.shard_search_routed_device(comm, dq.data_ptr(), nq, k, 5, oi.data_ptr(), od.data_ptr())
                torch.cuda.synchronize()
                assert served == nq and (oi.cpu().numpy() == fi).all() and (od.cpu().numpy() == fd).all()
                alive = rng.random(n) < 0.5
                bits = torch.from_numpy(capi.pack_bits(alive).view(np.int64)).cuda()
                ix.shard_search_routed_device(comm, dq.data_ptr(), nq, k, 5, oi.data_ptr(), od.data_ptr(), 0, bits.data_ptr(), n)
                torch.cuda.synchronize()
                ai, ad = ix.search(q, k, "nprobe=5", alive=alive)
                assert (oi.cpu().numpy() == ai).all() and (od.cpu().numpy() == ad).all()
                # two steps in flight
                o2 = torch.empty((nq, k), dtype=torch.int64, device="cuda")
                d2 = torch.empty((nq, k), dtype=torch.float32, device="cuda")
                assert ix.shard_search_routed_device_async(comm, dq.data_ptr(), nq, k, 5, oi.data_ptr(), od.data_ptr()) is None
                assert ix.shard_search_routed_device_async(comm, dq.data_ptr(), nq, k, 5, o2.data_ptr(), d2.data_ptr()) is not None
                comm.drain()
                torch.cuda.synchronize()
                for a, b in ((oi, od), (o2, d2)):
                    assert (a.cpu().numpy() == fi).all() and (b.cpu().numpy() == fd).all()
            finally:
                capi.set_option("route_self_rccl", None)
    comm.close()


@pytest.mark.gpu
def test_torch_distributed_fallback_transport_single_rank():
    """sharded.torch_comm (bench.py's fallback when libmsvs cannot create its own communicator): the all-gathers of the
    sharded search through torch.distributed's NCCL = RCCL process group on the caller's stream, with the one rank a 1-GPU
    box can host: == the plain search."""
    import subprocess
    import sys

    code = """
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
import myscaledb_amd.capi as capi
from myscaledb_amd import sharded
capi.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
comm = sharded.torch_comm()
rng = np.random.default_rng(5)
n, d, k = 8000, 32, 10
x = rng.standard_normal((n, d), dtype=np.float32)
ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, "ncentroids=16")
ix.train(x); ix.add(x); ix.build()
stream = torch.cuda.current_stream().cuda_stream
for nq in (3, 200):
    q = rng.standard_normal((nq, d), dtype=np.float32)
    dq = torch.from_numpy(q).cuda()
    oi = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    od = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    ix.shard_search_device(comm, dq.data_ptr(), nq, k, 5, oi.data_ptr(), od.data_ptr(), stream)
    torch.cuda.synchronize()
    fi, fd = ix.search(q, k, "nprobe=5")
    assert (oi.cpu().numpy() == fi).all() and (od.cpu().numpy() == fd).all()
comm.close()
dist.destroy_process_group()
print("torch_comm ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "torch_comm ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_bench_n_gt_1_code_path_runs_on_one_gpu():
    """bench.py's N > 1 branch (sharded build, msvs_shard_search_device, max-over-ranks timing, rank-0 JSON line) with two
    ranks sharing cuda:0 and a gloo transport (--test-single-device): the driver's multi-GPU run must not be the first
    time this code executes."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    # plain `python bench.py --gpus 2`: the script spawns its ranks itself (what a driver without torchrun would run)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--rows", "200000", "--batch", "512", "--nlist", "256", "--c4-rows", "60000", "--test-single-device"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [ln for ln in r.stdout.split("\n") if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["parallelism"].startswith("lists % 2")
    assert len(line) < 8000  # (the compact driver line; the full object is bench_detail.json)
    mg = out["legs"]["multi_gpu"]  # the routed form is the headline of N > 1 (its own batch per rank), the replicated one timed beside it
    assert mg["mode"] == "routed" and out["scaling"] == "weak" and len(mg["routed"]["routed_pairs_per_step_by_rank"]) == 2
    assert all(512 <= p_ <= 2 * 512 for p_ in mg["routed"]["routed_pairs_per_step_by_rank"]), mg
    assert mg["replicated"]["qps"] > 0 and mg["routed"]["stage_ms_rank0"]["shard_exchange"] > 0
    c4 = out["legs"]["c4_sharded"]
    assert "error" not in c4 and c4["batches"]["4096"]["qps"] > 0 and 0 < c4["rows_on_rank0"] < 120000, c4
    with open(os.path.join(root, "bench_detail.json")) as f:
        full = json.load(f)
    assert full["value"] == out["value"] and full["multi_gpu"]["routed"]["stage_ms_rank0"] == mg["routed"]["stage_ms_rank0"]
