"""Randomised parity of the multi-GPU forms: W processes sharing the box's GPU over the gloo transport (W drawn from 2, 3, 5, 6, 7 --
the fixed cases of test_sharded_many_ranks.py run 4 and 8), a random index (metric, d, rows, lists: not a multiple of W, fewer lists
than ranks included), per-rank batches of random sizes (empty ones included), k and nprobe at random, with / without a per-search
filter and resident delete bitmaps -- every entry (replicated, routed, routed with filters, routed with two steps in flight) against
the ORACLE on the unsharded structure, ids and distance bits.  MSVS_FUZZ_SHARD_ITERS (default 3) configurations, MSVS_FUZZ_SEED moves
the sequence; a failure names its seed.  Reference analogue: distributed == single MergeTree
(tests/integration/test_mqvs_distributed_hybrid_search/test.py:109-121)."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as o

ITERS = int(os.environ.get("MSVS_FUZZ_SHARD_ITERS", "3"))
SEED = int(os.environ.get("MSVS_FUZZ_SEED", "20260930"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _loguni(rng, lo, hi):
    return int(round(float(np.exp(rng.uniform(np.log(lo), np.log(hi))))))


def _case(seed):
    """Everything a configuration is, from its seed alone (every rank and the parent redo it)."""
    rng = np.random.default_rng(seed)
    world = int(rng.choice([2, 3, 5, 6, 7]))
    metric_name = str(rng.choice(["L2", "IP", "cosine"]))
    d = int(rng.choice([5, 16, 48, 100, 128, 256]))
    n = _loguni(rng, 400, 30000)
    nlist = max(1, min(_loguni(rng, 1, 400), n // 4))
    k = _loguni(rng, 1, 40)
    nprobe = min(_loguni(rng, 1, nlist), 64)
    sigma = float(rng.choice([0.1, 1.0]))
    centres = (4.0 * rng.standard_normal((nlist, d), dtype=np.float32)).astype(np.float32)
    x = (centres[rng.integers(0, nlist, n)] + np.float32(sigma) * rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    if rng.random() < 0.3:
        x[rng.integers(0, n, n // 3)] = x[int(rng.integers(0, n))]  # ties across ranks
    nq = 900
    q = (centres[rng.integers(0, nlist, nq)] + np.float32(sigma) * rng.standard_normal((nq, d), dtype=np.float32)).astype(np.float32)
    q[:8] = x[:8]
    alive = rng.random(n) < float(rng.choice([0.02, 0.5, 0.9]))
    deleted = rng.random(n) < 0.15
    # three routed steps: per rank a random slice of q (sizes from 0 to a few hundred; one step where most ranks are idle)
    steps = []
    for s in range(3):
        sizes = [0 if rng.random() < (0.6 if s == 2 else 0.15) else int(rng.choice([1, 3, 17, 64, 260])) for _ in range(world)]
        if sum(sizes) == 0:
            sizes[int(rng.integers(0, world))] = 5
        sels = [np.sort(rng.choice(nq, sz, replace=False)) if sz else np.arange(0) for sz in sizes]
        steps.append(sels)
    nrep = int(rng.choice([1, 9, 130, 300]))
    return dict(world=world, metric_name=metric_name, d=d, n=n, nlist=nlist, k=k, nprobe=nprobe, centres=centres, x=x, q=q, alive=alive,
                deleted=deleted, steps=steps, nrep=nrep)


def _worker(rank, world, port, seed, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import myscaledb_amd.capi as capi
        from myscaledb_amd import sharded
        capi.set_device(0)
        c = _case(seed)
        metric = {"L2": capi.METRIC_L2, "IP": capi.METRIC_IP, "cosine": capi.METRIC_COSINE}[c["metric_name"]]
        K, NPROBE, q = c["k"], c["nprobe"], c["q"]
        ix = capi.Index(capi.INDEX_IVFFLAT, metric, c["d"], "ncentroids=%d,shard_rank=%d,shard_world=%d" % (c["nlist"], rank, world))
        ix.set_centroids(c["centres"])
        ix.add(c["x"])
        ix.build()
        comm = sharded.gloo_comm()
        dq = torch.from_numpy(q).cuda()
        bits = torch.from_numpy(capi.pack_bits(c["alive"]).view(np.int64)).cuda()
        res = {}

        def run(fn, nq):
            oi = torch.full((max(nq, 1), K), -7, dtype=torch.int64, device="cuda")
            od = torch.empty((max(nq, 1), K), dtype=torch.float32, device="cuda")
            extra = fn(oi, od)
            torch.cuda.synchronize()
            return oi.cpu().numpy()[:nq], od.cpu().numpy()[:nq], extra

        nrep = c["nrep"]
        res["replicated"] = run(lambda oi, od: ix.shard_search_device(comm, dq.data_ptr(), nrep, K, NPROBE, oi.data_ptr(), od.data_ptr()), nrep)[:2]
        res["replicated_filtered"] = run(lambda oi, od: ix.shard_search_device(comm, dq.data_ptr(), nrep, K, NPROBE, oi.data_ptr(), od.data_ptr(), 0,
                                                                               bits.data_ptr(), len(c["alive"])), nrep)[:2]
        routed = []
        for sels in c["steps"]:
            sel = sels[rank]
            mine = torch.from_numpy(q[sel]).cuda() if len(sel) else None
            gi, gd, served = run(lambda oi, od: ix.shard_search_routed_device(comm, mine.data_ptr() if mine is not None else 0, len(sel), K, NPROBE,
                                                                             oi.data_ptr(), od.data_ptr()), len(sel))
            routed.append((gi, gd, served))
        res["routed"] = routed
        held = []
        for sels in c["steps"] + c["steps"][:1]:
            sel = sels[rank]
            mine = torch.from_numpy(q[sel]).cuda() if len(sel) else None
            oi = torch.full((max(len(sel), 1), K), -7, dtype=torch.int64, device="cuda")
            od = torch.empty((max(len(sel), 1), K), dtype=torch.float32, device="cuda")
            served = ctypes.c_uint64(0)
            ix.shard_search_routed_device_async(comm, mine.data_ptr() if mine is not None else 0, len(sel), K, NPROBE, oi.data_ptr(), od.data_ptr(),
                                                served=served)
            held.append((sel, mine, oi, od, served))
        comm.drain()
        torch.cuda.synchronize()
        res["routed_async"] = [(oi.cpu().numpy()[:len(sel)], od.cpu().numpy()[:len(sel)], served.value) for sel, _, oi, od, served in held]
        ix.set_delete_bitmap(~c["deleted"])
        filt = []
        for sels in c["steps"][:2]:
            sel = sels[rank]
            mine = torch.from_numpy(q[sel]).cuda() if len(sel) else None
            gi, gd, served = run(lambda oi, od: ix.shard_search_routed_device(comm, mine.data_ptr() if mine is not None else 0, len(sel), K, NPROBE,
                                                                             oi.data_ptr(), od.data_ptr(), 0, bits.data_ptr(), len(c["alive"])), len(sel))
            filt.append((gi, gd, served))
        res["routed_filtered"] = filt
        sel = c["steps"][0][rank]
        mine = torch.from_numpy(q[sel]).cuda() if len(sel) else None
        res["routed_deleted"] = run(lambda oi, od: ix.shard_search_routed_device(comm, mine.data_ptr() if mine is not None else 0, len(sel), K, NPROBE,
                                                                                oi.data_ptr(), od.data_ptr()), len(sel))
        ix.set_delete_bitmap(None)
        out.put((rank, res))
        dist.barrier()
        comm.close()
    finally:
        dist.destroy_process_group()


def _same(gi, gd, ei, ed, what):
    assert gi.shape == ei.shape, what
    assert (gi == ei).all(), (what, np.argwhere(gi != ei)[:4].tolist())
    assert (gd.view(np.uint32) == ed.view(np.uint32)).all(), what


@pytest.mark.gpu
def test_fuzz_sharded_forms_against_the_oracle():
    import myscaledb_amd.capi as capi
    for it in range(ITERS):
        seed = SEED + 500000 + it
        c = _case(seed)
        world, metric_name, K, NPROBE = c["world"], c["metric_name"], c["k"], c["nprobe"]
        tag = "seed %d: W %d %s n %d d %d nlist %d k %d nprobe %d" % (seed, world, metric_name, c["n"], c["d"], c["nlist"], K, NPROBE)
        ctx = mp.get_context("spawn")
        out = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, seed, out)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted([out.get(timeout=900) for _ in procs], key=lambda t: t[0])
        for p in procs:
            p.join(120)
            assert p.exitcode == 0, tag
        metric = {"L2": capi.METRIC_L2, "IP": capi.METRIC_IP, "cosine": capi.METRIC_COSINE}[metric_name]
        ix = capi.Index(capi.INDEX_IVFFLAT, metric, c["d"], "ncentroids=%d" % c["nlist"])
        ix.set_centroids(c["centres"])
        ix.add(c["x"])
        ix.build()
        cent, off, vecs, lids = ix.export()
        ix.close()

        def expect(alive=None):
            if metric_name == "cosine":
                oi, od, _ = o.ivf_search(cent, off, vecs, lids, o.normalize_rows(c["q"]), NPROBE, K, o.METRIC_IP, alive=alive)
                return oi, (np.float32(1) - od).astype(np.float32)
            oi, od, _ = o.ivf_search(cent, off, vecs, lids, c["q"], NPROBE, K, {"L2": o.METRIC_L2, "IP": o.METRIC_IP}[metric_name], alive=alive)
            return oi, od
        ei, ed = expect()
        fi, fd = expect(c["alive"])
        ai, ad = expect(c["alive"] & ~c["deleted"])
        di, dd = expect(~c["deleted"])
        nrep = c["nrep"]
        for r in range(world):
            got = res[r][1]
            _same(*got["replicated"], ei[:nrep], ed[:nrep], (tag, "replicated", r))
            _same(*got["replicated_filtered"], fi[:nrep], fd[:nrep], (tag, "replicated_filtered", r))
            for s, (gi, gd, _) in enumerate(got["routed"]):
                sel = c["steps"][s][r]
                _same(gi, gd, ei[sel], ed[sel], (tag, "routed", r, s))
            for s, (gi, gd, _) in enumerate(got["routed_async"]):
                sel = (c["steps"] + c["steps"][:1])[s][r]
                _same(gi, gd, ei[sel], ed[sel], (tag, "routed_async", r, s))
            for s, (gi, gd, _) in enumerate(got["routed_filtered"]):
                sel = c["steps"][s][r]
                _same(gi, gd, ai[sel], ad[sel], (tag, "routed_filtered", r, s))
            gi, gd, _ = got["routed_deleted"]
            sel = c["steps"][0][r]
            _same(gi, gd, di[sel], dd[sel], (tag, "routed_deleted", r))
        for s in range(3):
            total = sum(len(c["steps"][s][r]) for r in range(world))
            served = sum(res[r][1]["routed"][s][2] for r in range(world))
            assert total <= served <= world * total, (tag, s, served, total)
