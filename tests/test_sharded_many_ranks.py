"""The multi-GPU forms with MORE than two ranks: W = 4 and W = 8 processes sharing the one GPU of the box (gloo carries the
all-gather; the RCCL transport needs one GPU per rank), every product entry -- msvs_shard_search_device[_async],
msvs_shard_search_routed[_filtered]_device, msvs_shard_search_routed_device_async -- compared with the ORACLE on the unsharded
structure (ids and distance bits), uneven / empty batches, per-rank filters + resident delete bitmaps, a failing rank, a
mismatching k, a shard object replaced on one rank.  Reference analogue: distributed == single MergeTree
(tests/integration/test_mqvs_distributed_hybrid_search/test.py:109-121), MergeTreeBaseSearchManager.cpp:207-299,
VIWithDataPart.cpp:903-908."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as o

K, NPROBE, NLIST, D = 10, 8, 256, 48


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case():
    """Blobs (so the pre-pruning has something to drop), duplicates across the k-th rank, a few far rows that set list radii."""
    rng = np.random.default_rng(606)
    n, nq = 40000, 1200
    centres = 4.0 * rng.standard_normal((NLIST, D), dtype=np.float32)
    x = (centres[rng.integers(0, NLIST, n)] + rng.standard_normal((n, D), dtype=np.float32)).astype(np.float32)
    x[900:940] = x[5]
    u = rng.standard_normal((64, D)).astype(np.float32)
    x[2000:2064] = centres[:64] + 12.0 * u / np.linalg.norm(u, axis=1, keepdims=True)
    q = (centres[rng.integers(0, NLIST, nq)] + rng.standard_normal((nq, D), dtype=np.float32)).astype(np.float32)
    q[0] = x[5]
    q[1:33] = x[2000:2032] + 0.05 * rng.standard_normal((32, D)).astype(np.float32)
    alive = rng.random(n) < 0.4      # the per-search filter (the same predicate evaluated on every shard)
    deleted = rng.random(n) < 0.1    # lightweight deletes, resident per shard
    return x, q, centres, alive, deleted


def _slices(world, nq):
    """(name, rows of q that are rank r's own batch) for three steps: round robin; very uneven with an idle rank; one rank only."""
    def step0(r):
        return np.arange(r, 1000, world)  # (W = 4: 250 queries per rank, coarse pass through the centroid shadow; W = 8: 125, the canonical one)

    def step1(r):
        if r == 1:
            return np.arange(0)  # an empty batch still serves the others
        lo = 100 + 90 * r
        return np.arange(lo, lo + (77 if r == 0 else 5 + 3 * r))

    def step2(r):
        return np.arange(1100, 1133) if r == world - 1 else np.arange(0)
    return [step0, step1, step2]


def _worker(rank, world, port, metric_name, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import myscaledb_amd.capi as capi
        from myscaledb_amd import sharded
        capi.set_device(0)
        metric = {"L2": capi.METRIC_L2, "IP": capi.METRIC_IP, "cosine": capi.METRIC_COSINE}[metric_name]
        x, q, centres, alive, deleted = _case()
        params = "ncentroids=%d,shard_rank=%d,shard_world=%d" % (NLIST, rank, world)

        def make():
            ix_ = capi.Index(capi.INDEX_IVFFLAT, metric, D, params)
            ix_.set_centroids(centres)
            ix_.add(x)
            ix_.build()
            return ix_
        ix = make()
        comm = sharded.gloo_comm()
        capi.set_option("h16_prune", "2")
        capi.set_option("ivf_pass", "2")
        dq = torch.from_numpy(q).cuda()
        bits = torch.from_numpy(capi.pack_bits(alive).view(np.int64)).cuda()
        res = {}

        def run(fn, nq):
            oi = torch.full((max(nq, 1), K), -7, dtype=torch.int64, device="cuda")
            od = torch.empty((max(nq, 1), K), dtype=torch.float32, device="cuda")
            extra = fn(oi, od)
            torch.cuda.synchronize()
            return oi.cpu().numpy()[:nq], od.cpu().numpy()[:nq], extra

        # ---- replicated form: every rank works through the same 200 queries (not a multiple of 8 x anything: 203)
        res["replicated"] = run(lambda oi, od: ix.shard_search_device(comm, dq.data_ptr(), 203, K, NPROBE, oi.data_ptr(), od.data_ptr()), 203)[:2]
        res["replicated_filtered"] = run(lambda oi, od: ix.shard_search_device(comm, dq.data_ptr(), 203, K, NPROBE, oi.data_ptr(), od.data_ptr(), 0,
                                                                               bits.data_ptr(), len(alive)), 203)[:2]
        outs = [(torch.empty((hi - lo, K), dtype=torch.int64, device="cuda"), torch.empty((hi - lo, K), dtype=torch.float32, device="cuda"))
                for lo, hi in ((0, 90), (90, 91), (91, 203))]
        for (lo, hi), (bi, bd) in zip(((0, 90), (90, 91), (91, 203)), outs):
            ix.shard_search_device_async(comm, dq[lo:hi].data_ptr(), hi - lo, K, NPROBE, bi.data_ptr(), bd.data_ptr())
        comm.drain()
        torch.cuda.synchronize()
        res["replicated_async"] = (np.concatenate([a.cpu().numpy() for a, _ in outs]), np.concatenate([b.cpu().numpy() for _, b in outs]))

        # ---- routed form: own batches
        routed = []
        for sl in _slices(world, q.shape[0]):
            sel = sl(rank)
            mine = torch.from_numpy(q[sel]).cuda() if len(sel) else None
            gi, gd, served = run(lambda oi, od: ix.shard_search_routed_device(comm, mine.data_ptr() if mine is not None else 0, len(sel), K, NPROBE,
                                                                             oi.data_ptr(), od.data_ptr()), len(sel))
            routed.append((sel, gi, gd, served))
        res["routed"] = routed

        # ---- routed, two steps in flight: four batches back to back, results through the events / the drain
        sls = _slices(world, q.shape[0])
        batches = [sls[0](rank), sls[1](rank), sls[2](rank), sls[0](rank)[::-1].copy()]
        held = []
        for sel in batches:
            mine = torch.from_numpy(q[sel]).cuda() if len(sel) else None
            oi = torch.full((max(len(sel), 1), K), -7, dtype=torch.int64, device="cuda")
            od = torch.empty((max(len(sel), 1), K), dtype=torch.float32, device="cuda")
            served = ctypes.c_uint64(0)
            ev = ix.shard_search_routed_device_async(comm, mine.data_ptr() if mine is not None else 0, len(sel), K, NPROBE, oi.data_ptr(), od.data_ptr(),
                                                     served=served)
            assert (ev is None) == (len(held) == 0)
            held.append((sel, mine, oi, od, served))
        comm.drain()
        torch.cuda.synchronize()
        res["routed_async"] = [(sel, oi.cpu().numpy()[:len(sel)], od.cpu().numpy()[:len(sel)], served.value) for sel, _, oi, od, served in held]

        # ---- routed under filters: the per-search bitmap on every rank AND a resident delete bitmap
        ix.set_delete_bitmap(~deleted)
        filt = []
        for sl in _slices(world, q.shape[0])[:2]:
            sel = sl(rank)
            mine = torch.from_numpy(q[sel]).cuda() if len(sel) else None
            gi, gd, served = run(lambda oi, od: ix.shard_search_routed_device(comm, mine.data_ptr() if mine is not None else 0, len(sel), K, NPROBE,
                                                                             oi.data_ptr(), od.data_ptr(), 0, bits.data_ptr(), len(alive)), len(sel))
            filt.append((sel, gi, gd, served))
        res["routed_filtered"] = filt
        # the delete bitmap alone (no per-search filter): the ranks still must not pre-prune (the hint of the last step, then the matrix)
        sel = _slices(world, q.shape[0])[0](rank)
        mine = torch.from_numpy(q[sel]).cuda()
        res["routed_deleted"] = (sel,) + run(lambda oi, od: ix.shard_search_routed_device(comm, mine.data_ptr(), len(sel), K, NPROBE, oi.data_ptr(),
                                                                                         od.data_ptr()), len(sel))
        # ... on ONE rank only (the others learn it from the count matrix and redo their front phase without the pre-pruning)
        if rank != world - 1:
            ix.set_delete_bitmap(None)
        dist.barrier()
        res["routed_deleted_one_rank"] = (sel,) + run(lambda oi, od: ix.shard_search_routed_device(comm, mine.data_ptr(), len(sel), K, NPROBE,
                                                                                                  oi.data_ptr(), od.data_ptr()), len(sel))
        ix.set_delete_bitmap(None)

        # ---- one rank fails in its front phase: EVERY rank gets an error, nobody blocks, the next step works
        errs = []
        for bad_k, bad_rank in ((capi.MAX_K + 1, 1), (5, world - 2)):
            try:
                run(lambda oi, od: ix.shard_search_routed_device(comm, mine.data_ptr(), len(sel), bad_k if rank == bad_rank else K, NPROBE,
                                                                 oi.data_ptr(), od.data_ptr()), len(sel))
                errs.append(None)
            except capi.MsvsError as e:
                errs.append(e.code)
        res["errors"] = errs
        # ---- a shard object replaced on one rank (a reloaded part): noticed through its instance word, statistics gathered again
        if rank == world - 1:
            ix.close()
            ix = make()
        res["routed_after_reload"] = (sel,) + run(lambda oi, od: ix.shard_search_routed_device(comm, mine.data_ptr(), len(sel), K, NPROBE, oi.data_ptr(),
                                                                                              od.data_ptr()), len(sel))
        out.put((rank, res))
        dist.barrier()
        comm.close()
    finally:
        dist.destroy_process_group()


_STRUCT = {}


def _structure(metric_name):
    """The UNSHARDED index's exported structure (built once per metric in the parent): what the oracle searches."""
    if metric_name not in _STRUCT:
        import myscaledb_amd.capi as capi
        metric = {"L2": capi.METRIC_L2, "IP": capi.METRIC_IP, "cosine": capi.METRIC_COSINE}[metric_name]
        x, _, centres, _, _ = _case()
        ix = capi.Index(capi.INDEX_IVFFLAT, metric, D, "ncentroids=%d" % NLIST)
        ix.set_centroids(centres)
        ix.add(x)
        ix.build()
        _STRUCT[metric_name] = ix.export()
        ix.close()
    return _STRUCT[metric_name]


def _expect(metric_name, alive=None):
    cent, off, vecs, lids = _structure(metric_name)
    q = _case()[1]
    if metric_name == "cosine":
        oi, od, _ = o.ivf_search(cent, off, vecs, lids, o.normalize_rows(q), NPROBE, K, o.METRIC_IP, alive=alive)
        return oi, (np.float32(1) - od).astype(np.float32)
    oi, od, _ = o.ivf_search(cent, off, vecs, lids, q, NPROBE, K, {"L2": o.METRIC_L2, "IP": o.METRIC_IP}[metric_name], alive=alive)
    return oi, od


def _same(gi, gd, ei, ed, what):
    assert gi.shape == ei.shape, what
    assert (gi == ei).all(), (what, np.argwhere(gi != ei)[:4])
    assert (gd.view(np.uint32) == ed.view(np.uint32)).all(), what


@pytest.mark.gpu
@pytest.mark.parametrize("world,metric_name", [(4, "L2"), (4, "cosine"), (4, "IP"), (8, "L2")])
def test_sharded_forms_with_many_ranks_match_the_oracle(world, metric_name):
    import myscaledb_amd.capi as capi
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, metric_name, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=900) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    x, q, centres, alive, deleted = _case()
    ei, ed = _expect(metric_name)
    fi, fd = _expect(metric_name, alive=alive)
    ai, ad = _expect(metric_name, alive=alive & ~deleted)
    di, dd = _expect(metric_name, alive=~deleted)
    for r in range(world):
        got = res[r][1]
        _same(*got["replicated"], ei[:203], ed[:203], ("replicated", r))
        _same(*got["replicated_filtered"], fi[:203], fd[:203], ("replicated_filtered", r))
        _same(*got["replicated_async"], ei[:203], ed[:203], ("replicated_async", r))
        for name, (xi, xd) in (("routed", (ei, ed)), ("routed_async", (ei, ed)), ("routed_filtered", (ai, ad))):
            for step, (sel, gi, gd, _) in enumerate(got[name]):
                _same(gi, gd, xi[sel], xd[sel], (name, r, step))
        for name, (xi, xd) in (("routed_deleted", (di, dd)), ("routed_after_reload", (ei, ed))):
            sel, gi, gd, _ = got[name]
            _same(gi, gd, xi[sel], xd[sel], (name, r))
        assert got["errors"] == [capi.ERR_UNSUPPORTED_K, capi.ERR_INVALID_ARGUMENT], (r, got["errors"])
    # deletes on the LAST rank only: its lists lose their deleted rows, the other ranks' lists keep theirs
    _, off, _, lids = _structure(metric_name)
    owner_last = np.zeros(x.shape[0], bool)
    for l in range(NLIST):
        if l % world == world - 1:
            owner_last[lids[off[l]:off[l + 1]]] = True
    oi, od = _expect(metric_name, alive=~(deleted & owner_last))
    for r in range(world):
        sel, gi, gd, _ = res[r][1]["routed_deleted_one_rank"]
        _same(gi, gd, oi[sel], od[sel], ("routed_deleted_one_rank", r))
    # the routed pairs: every query visits at least one rank, at most all; with well separated blobs (L2) the pre-pruning keeps most at one
    for name in ("routed", "routed_async"):
        for step in range(3):
            total = sum(len(res[r][1][name][step][0]) for r in range(world))
            served = sum(res[r][1][name][step][3] for r in range(world))
            assert total <= served <= world * total, (name, step, served, total)
            if metric_name == "L2" and step == 0:
                assert served < 0.5 * world * total, "pre-pruning did not route: %d pairs for %d queries over %d ranks" % (served, total, world)
    # under a filter nobody may pre-prune: a query goes to every rank that owns one of its probes
    for step in range(2):
        total = sum(len(res[r][1]["routed_filtered"][step][0]) for r in range(world))
        served_f = sum(res[r][1]["routed_filtered"][step][3] for r in range(world))
        served_u = sum(res[r][1]["routed"][step][3] for r in range(world))
        assert served_f >= served_u and served_f >= total


# ---------------------------------------------------------------------------------------- a routed batch beyond one sub-batch

def _big_case():
    rng = np.random.default_rng(99)
    n, d, nlist = 20000, 16, 256
    centres = 4.0 * rng.standard_normal((nlist, d), dtype=np.float32)
    x = (centres[rng.integers(0, nlist, n)] + rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    nq = 33000  # index_search_device cuts a batch into sub-batches of max(256, 2^21 / nprobe) = 32768 queries at nprobe 64
    q = (centres[rng.integers(0, nlist, nq)] + rng.standard_normal((nq, d), dtype=np.float32)).astype(np.float32)
    return x, q, centres


def _big_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import myscaledb_amd.capi as capi
        from myscaledb_amd import sharded
        capi.set_device(0)
        x, q, centres = _big_case()
        ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, x.shape[1], "ncentroids=%d,shard_rank=%d,shard_world=%d" % (centres.shape[0], rank, world))
        ix.set_centroids(centres)
        ix.add(x)
        ix.build()
        comm = sharded.gloo_comm()
        sel = np.arange(q.shape[0] - 10) if rank == 0 else np.arange(q.shape[0] - 10, q.shape[0])  # 32990 queries on rank 0, 10 on rank 1
        dq = torch.from_numpy(q[sel]).cuda()
        oi = torch.full((len(sel), K), -7, dtype=torch.int64, device="cuda")
        od = torch.empty((len(sel), K), dtype=torch.float32, device="cuda")
        served = ix.shard_search_routed_device(comm, dq.data_ptr(), len(sel), K, 64, oi.data_ptr(), od.data_ptr())
        torch.cuda.synchronize()
        out.put((rank, sel, oi.cpu().numpy(), od.cpu().numpy(), served))
        dist.barrier()
        comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_routed_batch_beyond_one_sub_batch_matches_the_oracle():
    """a per-rank batch above the device-level search's sub-batch size (32768 queries at nprobe 64) -- the sub-batch
    loop must carry the routed front phase's index-wide ProbeWords fields and offset its per-pair outputs (the survivors' array the
    route mask reads): every query's result == the oracle's on the unsharded structure."""
    import myscaledb_amd.capi as capi
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_big_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=900) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    x, q, centres = _big_case()
    ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, x.shape[1], "ncentroids=%d" % centres.shape[0])
    ix.set_centroids(centres)
    ix.add(x)
    ix.build()
    cent, off, vecs, lids = ix.export()
    ix.close()
    ei, ed, _ = o.ivf_search(cent, off, vecs, lids, q, 64, K, o.METRIC_L2, threads=8)
    total = 0
    for rank, sel, gi, gd, served in res:
        _same(gi, gd, ei[sel], ed[sel], ("big", rank))
        total += served
    assert q.shape[0] <= total <= world * q.shape[0]


# ---------------------------------------------------------------------------------------- a shard whose rows are smaller than other shards' centroids

def _small_case():
    """Rank 0's lists (l % 4 == 0) hold rows around small centres, the other ranks' lists around centres four times as far out: every
    shard holds EVERY centroid, and the centroid table shares the rows' fp16 scale."""
    rng = np.random.default_rng(4242)
    n, d, nlist = 60000, 48, 1024  # (1024 lists x >= 1024 queries: the batched coarse stage, with or without a centroid shadow)
    centres = rng.standard_normal((nlist, d), dtype=np.float32)
    centres[0::4] *= 1.0
    centres[1::4] *= 4.0
    centres[2::4] *= 4.0
    centres[3::4] *= 4.0
    x = (centres[rng.integers(0, nlist, n)] + 0.1 * rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    q = (centres[rng.integers(0, nlist, 4400)] + 0.1 * rng.standard_normal((4400, d), dtype=np.float32)).astype(np.float32)
    return x, q, centres


def _small_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import myscaledb_amd.capi as capi
        from myscaledb_amd import sharded
        capi.set_device(0)
        x, q, centres = _small_case()
        ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, x.shape[1], "ncentroids=%d,shard_rank=%d,shard_world=%d" % (centres.shape[0], rank, world))
        ix.set_centroids(centres)
        ix.add(x)
        ix.build()
        comm = sharded.gloo_comm()
        sel = np.arange(rank * 1100, (rank + 1) * 1100)  # 1100 queries per rank: the batched coarse stage
        dq = torch.from_numpy(q[sel]).cuda()
        oi = torch.full((len(sel), K), -7, dtype=torch.int64, device="cuda")
        od = torch.empty((len(sel), K), dtype=torch.float32, device="cuda")
        served = 0
        for _ in range(2):  # (the first step gathers the lists' radii; the second routes by them from the start)
            served = ix.shard_search_routed_device(comm, dq.data_ptr(), len(sel), K, NPROBE, oi.data_ptr(), od.data_ptr())
            torch.cuda.synchronize()
        out.put((rank, sel, oi.cpu().numpy(), od.cpu().numpy(), served))
        dist.barrier()
        comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_routed_pruning_on_a_shard_whose_rows_are_smaller_than_the_other_shards_centroids():
    """Round 6, found by running bench.py --gpus 8 on one GPU: a shard keeps every centroid but only its own lists' rows, and the centroid
    shadow shares the rows' fp16 scale -- a shard whose largest row component was below the largest centroid component had no centroid
    shadow, hence no approximate centroid distances, hence no pre-pruning of ITS queries: they visited every rank that owned a probe
    (3.6 ranks per query at W = 8 on the bench data instead of 1.0).  The scale now covers the centroids.  Results == the oracle's, and
    the routed pairs stay near one per query on well separated blobs."""
    import myscaledb_amd.capi as capi
    world = 4
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_small_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=900) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    x, q, centres = _small_case()
    ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, x.shape[1], "ncentroids=%d" % centres.shape[0])
    ix.set_centroids(centres)
    ix.add(x)
    ix.build()
    cent, off, vecs, lids = ix.export()
    ix.close()
    ei, ed, _ = o.ivf_search(cent, off, vecs, lids, q, NPROBE, K, o.METRIC_L2)
    total = served = 0
    for rank, sel, gi, gd, sv in res:
        _same(gi, gd, ei[sel], ed[sel], ("small rows", rank))
        total += len(sel)
        served += sv
    assert total <= served <= 1.3 * total, "pre-pruning did not route: %d pairs for %d queries over %d ranks" % (served, total, world)
