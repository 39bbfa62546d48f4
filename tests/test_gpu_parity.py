"""GPU parity tests proper: everything goes through the C-ABI of libmsvs.so (ctypes) and is compared with the
CPU oracle on the same seeded inputs -- ids bit-exact, distances bit-exact (the oracle and the kernels share one
canonical arithmetic), which is stricter than the north-star 1e-4 relative tolerance."""
import os
import tempfile

import numpy as np
import pytest

import myscaledb_amd.capi as capi
import myscaledb_amd.host as mhost
from golden_util import TextIndex, eval_filter, f32_of, load_goldens, materialize, tokenize
from oracle import oracle as o

pytestmark = pytest.mark.gpu
G = load_goldens()
OM = {capi.METRIC_L2: o.METRIC_L2, capi.METRIC_IP: o.METRIC_IP, capi.METRIC_COSINE: o.METRIC_COSINE}


def same(a_ids, a_dis, b_ids, b_dis):
    assert a_ids.shape == b_ids.shape
    assert (a_ids == b_ids).all(), np.argwhere(a_ids != b_ids)[:5]
    assert (a_dis.view(np.uint32) == b_dis.view(np.uint32)).all()


# ---------------------------------------------------------------------------------------- seam A2

@pytest.mark.parametrize("nx,ny,d,k", [(1, 100, 3, 10), (3, 1000, 128, 10), (8, 5000, 768, 10), (17, 3000, 100, 30),
                                       (2, 777, 5, 7), (4, 2000, 64, 100), (1, 4000, 48, 200), (5, 1536, 1536, 10),
                                       (9, 10000, 128, 10), (1, 50000, 128, 64), (33, 999, 36, 1)])
@pytest.mark.parametrize("metric", [capi.METRIC_L2, capi.METRIC_IP])
def test_knn_matches_oracle(nx, ny, d, k, metric):
    rng = np.random.default_rng(nx * 1000 + ny + d + k)
    x = rng.standard_normal((nx, d), dtype=np.float32)
    y = rng.standard_normal((ny, d), dtype=np.float32)
    ids, dis = capi.knn(x, y, k, metric)
    oi, od = o.knn(x, y, k, OM[metric])
    same(ids, dis, oi, od)


def test_knn_fewer_rows_than_k_pads_with_minus_one():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 16), dtype=np.float32)
    y = rng.standard_normal((5, 16), dtype=np.float32)
    for metric in (capi.METRIC_L2, capi.METRIC_IP):
        ids, dis = capi.knn(x, y, 8, metric)
        oi, od = o.knn(x, y, 8, OM[metric])
        same(ids, dis, oi, od)
        assert (ids[:, 5:] == -1).all()


def test_knn_ties_break_by_ascending_id_and_nan_inf_rows_are_skipped():
    rng = np.random.default_rng(2)
    y = rng.standard_normal((600, 32), dtype=np.float32)
    y[100:300] = y[7]  # 201 identical rows
    y[400, 3] = np.nan
    y[401, 5] = np.inf
    y[402] = np.finfo(np.float32).max  # FLT_MAX padding of empty rows (MergeTreeVSManager.cpp:1380)
    x = y[[7, 8]].copy()
    for metric in (capi.METRIC_L2, capi.METRIC_IP):
        ids, dis = capi.knn(x, y, 50, metric)
        oi, od = o.knn(x, y, 50, OM[metric])
        same(ids, dis, oi, od)
    ids, _ = capi.knn(x, y, 50, capi.METRIC_L2)
    assert ids[0, :5].tolist() == [7, 100, 101, 102, 103]
    assert not np.isin(ids, [400, 401, 402]).any()


def test_knn_cosine_is_not_implemented_like_the_reference():
    x = np.zeros((1, 4), np.float32)
    with pytest.raises(capi.MsvsError) as e:
        capi.knn(x, x, 1, capi.METRIC_COSINE)
    assert e.value.code == capi.ERR_NOT_IMPLEMENTED


def test_k_too_large_is_an_error_not_a_fallback():
    x = np.zeros((1, 4), np.float32)
    with pytest.raises(capi.MsvsError) as e:
        capi.knn(x, np.zeros((1000, 4), np.float32), 4097, capi.METRIC_L2)
    assert e.value.code == capi.ERR_UNSUPPORTED_K


@pytest.mark.parametrize("metric", [capi.METRIC_L2, capi.METRIC_IP])
def test_large_k_runs_exact_rounds(metric):
    """k beyond one top-k pass (256): the reference's k + deleted-rows over-fetch and LIMIT 1000 style queries."""
    rng = np.random.default_rng(51)
    y = rng.standard_normal((3000, 24), dtype=np.float32)
    y[500:560] = y[3]  # ties across a round boundary
    x = np.concatenate([rng.standard_normal((2, 24), dtype=np.float32), y[3:4]])
    for k in (257, 700, 3000):
        ids, dis = capi.knn(x, y, k, metric)
        oi, od = o.knn(x, y, k, OM[metric])
        same(ids, dis, oi, od)
    alive = rng.random(3000) < 0.4
    ids, dis = capi.knn(x, y, 1500, metric, alive=alive)  # fewer alive rows than k: -1 padded tail
    oi, od = o.knn(x, y, 1500, OM[metric], alive=alive)
    same(ids, dis, oi, od)
    ix = build_ivf(y, metric, 16)
    i1, d1 = ix.search(x, 600, "nprobe=16", alive=alive)
    oi, od, _ = oracle_on_exported(ix, x, 16, 600, metric, alive=alive)
    same(i1, d1, oi, od)


def test_filtered_knn_matches_oracle():
    rng = np.random.default_rng(52)
    y = rng.standard_normal((5000, 96), dtype=np.float32)
    x = rng.standard_normal((9, 96), dtype=np.float32)
    alive = rng.random(5000) < 0.25
    for metric in (capi.METRIC_L2, capi.METRIC_IP):
        ids, dis = capi.knn(x, y, 20, metric, alive=alive)
        oi, od = o.knn(x, y, 20, OM[metric], alive=alive)
        same(ids, dis, oi, od)


def test_normalize_matches_reference_normalize():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((300, 77), dtype=np.float32) * 5
    x[5] = 0  # below FLT_EPSILON: left untouched
    x[6] = 1e-5
    a = capi.normalize(x)
    b = o.normalize_rows(x)
    assert (a.view(np.uint32) == b.view(np.uint32)).all()


@pytest.mark.parametrize("name", ["00001_flat_l2", "00002_batch_l2", "00002_batch_ip"])
def test_golden_via_knn(name):
    c = G[name]
    parts = c.get("parts") or [c["base"]]
    k = c["k"]
    q = np.array(c["queries"], np.float32)
    m = capi.METRICS[c["metric"]]
    res = []
    for segs in parts:
        ids, vecs, _ = materialize(segs)
        i, d = capi.knn(q, vecs, k, m)
        res.append((ids[i], d))
    for qi in range(len(q)):
        ids = np.concatenate([r[0][qi] for r in res])
        dis = np.concatenate([r[1][qi] for r in res])
        order = np.argsort(-dis if c["metric"] == "IP" else dis, kind="stable")[:k]
        assert ids[order].tolist() == c["ids"][qi]
        assert dis[order].tolist() == f32_of(c["dists"][qi]).tolist()


# ---------------------------------------------------------------------------------------- seam A1: FLAT

@pytest.mark.parametrize("name", ["00001_flat_l2", "00003_prewhere", "00008_empty_vectors", "00014_cosine_d4_index",
                                  "00028_768_l2", "00028_768_cosine", "00028_768_cosine_where", "00028_768_cosine_lwd"])
def test_golden_via_flat_index(name):
    c = G[name]
    ids, vecs, empty = materialize(c["base"])
    metric = capi.METRICS[c["metric"]]
    ix = capi.Index(capi.INDEX_FLAT, metric, vecs.shape[1])
    ix.add(vecs[~empty], ids[~empty])  # empty rows are never fed to the index build (VIPartReader.h:240-244)
    ix.build()
    alive = None
    if c.get("filter") or c.get("deleted"):
        alive = np.zeros(int(ids.max()) + 1, bool)
        alive[ids] = eval_filter(c.get("filter"), ids) & ~np.isin(ids, c.get("deleted", []))
    i, d = ix.search(np.array(c["queries"], np.float32), c["k"], alive=alive)
    exp = f32_of(c["dists"][0])
    assert i[0].tolist() == c["ids"][0]
    if name.startswith("00028"):
        assert np.abs(d[0] / exp - 1).max() < 2e-6  # accumulation order not pinned at d=768, tolerance 1e-4
    else:
        assert d[0].tolist() == exp.tolist()


@pytest.mark.parametrize("metric", [capi.METRIC_L2, capi.METRIC_IP, capi.METRIC_COSINE])
@pytest.mark.parametrize("n,d,nq,k", [(5000, 128, 7, 10), (3000, 768, 64, 10), (1200, 40, 3, 100)])
def test_flat_index_filter_and_labels(metric, n, d, nq, k):
    rng = np.random.default_rng(n + d)
    x = rng.standard_normal((n, d), dtype=np.float32)
    labels = rng.permutation(n * 3)[:n].astype(np.int64)
    q = rng.standard_normal((nq, d), dtype=np.float32)
    alive = rng.random(n * 3) < 0.3
    ix = capi.Index(capi.INDEX_FLAT, metric, d)
    ix.add(x[: n // 2], labels[: n // 2])  # chunked feed
    ix.add(x[n // 2:], labels[n // 2:])
    ix.build()
    assert ix.ready and ix.num_data == n
    ids, dis = ix.search(q, k, alive=alive)
    if metric == capi.METRIC_COSINE:
        xn, qn = o.normalize_rows(x), o.normalize_rows(q)
        oi, od = o.knn(qn, xn, k, o.METRIC_IP, labels=labels, alive=alive[labels])
        od = (np.float32(1) - od).astype(np.float32)
    else:
        oi, od = o.knn(q, x, k, OM[metric], labels=labels, alive=alive[labels])
    same(ids, dis, oi, od)


@pytest.mark.parametrize("metric", [capi.METRIC_L2, capi.METRIC_IP, capi.METRIC_COSINE])
@pytest.mark.parametrize("n,d,nq,k,order", [(70000, 96, 300, 10, "random"),    # several tiles: two row blocks per wavefront
                                           (20000, 768, 48, 10, "random"),    # one tile: the table streams once
                                           (33001, 100, 130, 40, "random"),   # rows not a multiple of 32, 64 candidates re-ranked
                                           (1500, 20, 17, 1, "random"),       # a table smaller than the sample
                                           (60000, 64, 100, 10, "clustered")])  # rows SORTED by cluster: a strided sample still sees them all
def test_flat_shadow_pass_matches_oracle(metric, n, d, nq, k, order, opt):
    """A batch against a FLAT index goes through the index's fp16 shadow (h16_flat_kernel: sample of 64 blocks spread over the
    table -> cut -> exhaustive scan of (segment, tile) items -> candidates -> canonical re-rank + certificate): the canonical
    exhaustive answer bit for bit, with labels, filters (dense and nearly empty), tables sorted by cluster, every segment size
    and tile shape, overflowing candidate buffers and failing certificates (canonical fallback); == the split-bf16 pass it replaces."""
    rng = np.random.default_rng(n + d + nq + 3)
    if order == "clustered":
        centres = 4.0 * rng.standard_normal((40, d), dtype=np.float32)
        z = np.sort(rng.integers(0, 40, n))
        x = (centres[z] + rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
        q = (centres[rng.integers(0, 40, nq)] + rng.standard_normal((nq, d), dtype=np.float32)).astype(np.float32)
    else:
        x = rng.standard_normal((n, d), dtype=np.float32)
        q = (x[rng.integers(0, n, nq)] + 0.3 * rng.standard_normal((nq, d), dtype=np.float32)).astype(np.float32)
    x[100:105] = x[7]  # ties
    labels = rng.permutation(n * 2)[:n].astype(np.int64)
    ix = capi.Index(capi.INDEX_FLAT, metric, d)
    ix.add(x, labels)
    ix.build()
    opt("flat_mfma", "2")  # also for the shapes below the automatic threshold
    xs, qs, om = (o.normalize_rows(x), o.normalize_rows(q), o.METRIC_IP) if metric == capi.METRIC_COSINE else (x, q, OM[metric])

    def expect(alive=None):
        oi, od = o.knn(qs, xs, k, om, labels=labels, alive=None if alive is None else alive[labels])
        return oi, ((np.float32(1) - od).astype(np.float32) if metric == capi.METRIC_COSINE else od)

    capi.profile_reset()
    capi.profile_enable(True)
    q0, f0 = capi.prefilter_stats()
    ids, dis = ix.search(q, k)
    capi.profile_enable(False)
    same(ids, dis, *expect())
    q1, f1 = capi.prefilter_stats()
    assert q1 - q0 == nq and capi.profile_get("flat_shadow_scan")[0] >= 1, "the shadow pass is the one that ran"
    capi.profile_reset()
    if order == "random":
        assert f1 - f0 <= nq // 10
    dense, sparse = rng.random(n * 2) < 0.4, rng.random(n * 2) < 0.003
    for alive in (dense, sparse):
        ids, dis = ix.search(q, k, alive=alive)
        same(ids, dis, *expect(alive))
    for segb, ncb, shape in ((8, 1, 1), (17, 2, 3), (256, 3, 6), (1, 0, 1)):
        opt("flat_segb", str(segb))
        opt("flat_ncb", str(ncb))
        opt("flat_h16", str(shape))
        ids, dis = ix.search(q, k)
        same(ids, dis, *expect())
    opt("flat_segb", None)
    opt("flat_ncb", None)
    opt("flat_h16", None)
    opt("h16_grid", "3")  # three workgroups walk every item
    ids, dis = ix.search(q, k, alive=dense)
    same(ids, dis, *expect(dense))
    opt("h16_grid", None)
    opt("cand_cap", "64")  # overflowing candidate buffers: no certificate, canonical fallback
    ids, dis = ix.search(q, k)
    same(ids, dis, *expect())
    opt("cand_cap", None)
    opt("ivf_eps_scale", "1e12")  # no certificates at all
    ids, dis = ix.search(q[:40], k)
    ei, ed = expect()
    same(ids, dis, ei[:40], ed[:40])
    opt("ivf_eps_scale", None)
    opt("flat_h16", "0")  # the split-bf16 pass over the f32 rows
    ids, dis = ix.search(q, k)
    same(ids, dis, *expect())


@pytest.mark.parametrize("metric", [capi.METRIC_L2, capi.METRIC_IP, capi.METRIC_COSINE])
@pytest.mark.parametrize("n,d,k", [(50000, 768, 10), (33001, 100, 40), (200_000, 64, 10)])
def test_flat_shadow_few_queries_match_oracle(metric, n, d, k, opt):
    """Round 5: ONE query per call (and 2 .. 15) over a FLAT index takes the fp16 shadow pass too (`flat_few`; by default from
    ~128 M elements on, here forced): the canonical exhaustive answer bit for bit -- through the host-pointer entry the host calls
    (one query per VectorIndex::search, VIWithDataPart.cpp:922-926), with labels, filters, failing certificates, and equal to
    what the same query returns inside a batch and from the canonical f32 scan."""
    rng = np.random.default_rng(n + d + k)
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = (x[rng.integers(0, n, 15)] + 0.3 * rng.standard_normal((15, d), dtype=np.float32)).astype(np.float32)
    x[100:105] = x[7]  # ties
    q[3] = x[7]
    labels = rng.permutation(n * 2)[:n].astype(np.int64)
    ix = capi.Index(capi.INDEX_FLAT, metric, d)
    ix.add(x, labels)
    ix.build()
    opt("flat_few", "2")
    xs, qs, om = (o.normalize_rows(x), o.normalize_rows(q), o.METRIC_IP) if metric == capi.METRIC_COSINE else (x, q, OM[metric])

    def expect(alive=None):
        oi, od = o.knn(qs, xs, k, om, labels=labels, alive=None if alive is None else alive[labels])
        return oi, ((np.float32(1) - od).astype(np.float32) if metric == capi.METRIC_COSINE else od)

    ei, ed = expect()
    capi.profile_reset()
    capi.profile_enable(True)
    for nq in (1, 2, 5, 15):
        ids, dis = ix.search(q[:nq], k)
        same(ids, dis, ei[:nq], ed[:nq])
    for j in (3, 14):  # alone == inside the batch
        ids, dis = ix.search(q[j:j + 1], k)
        same(ids, dis, ei[j:j + 1], ed[j:j + 1])
    capi.profile_enable(False)
    assert capi.profile_get("flat_shadow_scan")[0] >= 6, "the shadow pass is the one that ran"
    capi.profile_reset()
    dense, sparse = rng.random(n * 2) < 0.4, rng.random(n * 2) < 0.003
    for alive in (dense, sparse):
        fi, fd = expect(alive)
        for nq in (1, 7):
            ids, dis = ix.search(q[:nq], k, alive=alive)
            same(ids, dis, fi[:nq], fd[:nq])
    opt("ivf_eps_scale", "1e12")  # no certificates at all: the canonical fallback answers
    ids, dis = ix.search(q[:2], k)
    same(ids, dis, ei[:2], ed[:2])
    opt("ivf_eps_scale", None)
    for knob in ("flat_host_signal", "flat_sample_few"):  # copies + stream synchronisation / the two-launch sample + cut
        opt(knob, "0")
        for nq in (1, 6):
            ids, dis = ix.search(q[:nq], k)
            same(ids, dis, ei[:nq], ed[:nq])
        opt(knob, None)
    opt("flat_few", "0")  # the canonical f32 scan
    ids, dis = ix.search(q[:1], k)
    same(ids, dis, ei[:1], ed[:1])


@pytest.mark.parametrize("metric", [capi.METRIC_L2, capi.METRIC_IP, capi.METRIC_COSINE])
@pytest.mark.parametrize("n,d,nq,k", [(40000, 96, 200, 10), (20000, 768, 48, 10), (33000, 100, 130, 40), (5000, 20, 17, 1)])
def test_flat_index_batches_through_the_candidate_pass(metric, n, d, nq, k, opt):
    """A batch against the whole table: matrix-core candidate pass + canonical re-rank, exact and certified; the
    answer (ids, distance bits) must be the canonical exhaustive one, with filters, labels and forced fallbacks."""
    rng = np.random.default_rng(n + d + nq)
    x = rng.standard_normal((n, d), dtype=np.float32)
    labels = rng.permutation(n * 2)[:n].astype(np.int64)
    q = (x[rng.integers(0, n, nq)] + 0.3 * rng.standard_normal((nq, d), dtype=np.float32)).astype(np.float32)
    ix = capi.Index(capi.INDEX_FLAT, metric, d)
    ix.add(x, labels)
    ix.build()
    opt("flat_mfma", "2")  # also for the shapes below the automatic threshold
    xs, qs, om = (o.normalize_rows(x), o.normalize_rows(q), o.METRIC_IP) if metric == capi.METRIC_COSINE else (x, q, OM[metric])

    def expect(alive=None):
        oi, od = o.knn(qs, xs, k, om, labels=labels, alive=None if alive is None else alive[labels])
        return oi, ((np.float32(1) - od).astype(np.float32) if metric == capi.METRIC_COSINE else od)

    q0, _ = capi.prefilter_stats()
    ids, dis = ix.search(q, k)
    same(ids, dis, *expect())
    assert capi.prefilter_stats()[0] - q0 == nq  # the pass is the one that ran
    alive = rng.random(n * 2) < 0.4
    ids, dis = ix.search(q, k, alive=alive)
    same(ids, dis, *expect(alive))
    opt("ivf_eps_scale", "1e12")  # no certificates: canonical fallback for every query
    ids, dis = ix.search(q, k)
    same(ids, dis, *expect())
    opt("fb_cap", "5")  # ... in rounds of 5 queries (the fallback buffers hold `cap` queries, not nq)
    ids, dis = ix.search(q, k)
    same(ids, dis, *expect())
    opt("fb_cap", None)
    opt("ivf_eps_scale", None)
    opt("ivf_nqg", "2")  # 256-query tiles
    ids, dis = ix.search(q, k)
    same(ids, dis, *expect())
    opt("ivf_nqg", None)
    # small candidate buffers = what a long table looks like: sample first, its m-th candidate cuts the rest
    opt("cand_cap", "1024")
    ids, dis = ix.search(q, k)
    same(ids, dis, *expect())
    ids, dis = ix.search(q, k, alive=alive)
    same(ids, dis, *expect(alive))
    opt("cand_cap", None)
    opt("flat_mfma", "0")
    ids, dis = ix.search(q, k)
    same(ids, dis, *expect())


# ---------------------------------------------------------------------------------------- seam A1: IVFFLAT

def build_ivf(x, metric, nlist, ids=None, params=""):
    ix = capi.Index(capi.INDEX_IVFFLAT, metric, x.shape[1], "ncentroids=%d,kmeans_iters=5%s" % (nlist, params))
    ix.train(x)
    half = x.shape[0] // 2
    ix.add(x[:half], None if ids is None else ids[:half])
    ix.add(x[half:], None if ids is None else ids[half:])
    ix.build()
    return ix


def oracle_on_exported(ix, q, nprobe, k, metric, alive=None):
    cent, off, vecs, lids = ix.export()
    if metric == capi.METRIC_COSINE:
        oi, od, pr = o.ivf_search(cent, off, vecs, lids, o.normalize_rows(q), nprobe, k, o.METRIC_IP, alive=alive)
        return oi, (np.float32(1) - od).astype(np.float32), pr
    return o.ivf_search(cent, off, vecs, lids, q, nprobe, k, OM[metric], alive=alive)


@pytest.mark.parametrize("metric", [capi.METRIC_L2, capi.METRIC_IP, capi.METRIC_COSINE])
@pytest.mark.parametrize("n,d,nlist,nq,nprobe,k", [(20000, 128, 64, 16, 8, 10), (30000, 768, 128, 5, 32, 10),
                                                   (8000, 100, 32, 1, 4, 100), (6000, 36, 16, 70, 16, 30),
                                                   (10000, 64, 64, 12, 16, 10), (9000, 32, 40, 300, 8, 5)])
def test_ivfflat_matches_oracle_on_exported_structure(metric, n, d, nlist, nq, nprobe, k):
    rng = np.random.default_rng(n + d + nlist)
    centers = rng.standard_normal((nlist, d), dtype=np.float32) * 2
    x = (centers[rng.integers(0, nlist, n)] + rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    q = (centers[rng.integers(0, nlist, nq)] + rng.standard_normal((nq, d), dtype=np.float32)).astype(np.float32)
    ix = build_ivf(x, metric, nlist)
    assert ix.num_data == n and ix.num_lists == nlist
    ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
    oi, od, _ = oracle_on_exported(ix, q, nprobe, k, metric)
    same(ids, dis, oi, od)
    # with a filter bitmap (PREWHERE / lightweight delete)
    alive = rng.random(n) < 0.5
    ids, dis = ix.search(q, k, "nprobe=%d" % nprobe, alive=alive)
    oi, od, _ = oracle_on_exported(ix, q, nprobe, k, metric, alive=alive)
    same(ids, dis, oi, od)


@pytest.mark.parametrize("metric", [capi.METRIC_L2, capi.METRIC_IP, capi.METRIC_COSINE])
@pytest.mark.parametrize("n,d,nlist,nq,nprobe,k", [(40000, 768, 16, 200, 4, 10), (30000, 100, 8, 256, 8, 40),
                                                   (20000, 20, 8, 129, 3, 1), (3000, 64, 4, 100, 4, 12),
                                                   (6000, 1536, 4, 64, 2, 10), (50000, 200, 40, 700, 5, 10)])
@pytest.mark.parametrize("h16", [1, 0])
def test_ivfflat_matrix_core_candidate_pass_matches_oracle(metric, n, d, nlist, nq, nprobe, k, h16, opt):
    """>= 16 queries per list: MFMA candidate pass + canonical re-rank + certificate.  h16 = 1: over the fp16 shadow
    (h16_scan_kernels.hpp, the default); h16 = 0: split bf16 over the f32 rows (mfma_scan_kernels.hpp).  The result
    must be the canonical one bit for bit, with and without certificates."""
    rng = np.random.default_rng(n + d + nlist + 5)
    centers = rng.standard_normal((nlist, d), dtype=np.float32) * 2
    x = (centers[rng.integers(0, nlist, n)] + rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    q = (centers[rng.integers(0, nlist, nq)] + rng.standard_normal((nq, d), dtype=np.float32)).astype(np.float32)
    ix = build_ivf(x, metric, nlist)
    opt("ivf_h16", str(h16))
    q0, f0 = capi.prefilter_stats()
    ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
    oi, od, _ = oracle_on_exported(ix, q, nprobe, k, metric)
    same(ids, dis, oi, od)
    alive = rng.random(n) < 0.5
    ids, dis = ix.search(q, k, "nprobe=%d" % nprobe, alive=alive)
    oa, oda, _ = oracle_on_exported(ix, q, nprobe, k, metric, alive=alive)
    same(ids, dis, oa, oda)
    q1, f1 = capi.prefilter_stats()
    assert q1 - q0 == 2 * nq  # the matrix-core pass is the one that ran
    assert f1 - f0 <= nq // 4  # and it certified (almost) every query
    # no certificate for anybody: every query takes the canonical fallback, same answer
    opt("ivf_eps_scale", "1e12")
    ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
    same(ids, dis, oi, od)
    q2, f2 = capi.prefilter_stats()
    assert f2 - f1 == nq
    opt("fb_cap", "7")  # the same in rounds of 7 queries
    ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
    same(ids, dis, oi, od)
    opt("fb_cap", None)
    opt("ivf_eps_scale", None)
    if not h16:
        # 256-query tiles (one 8-wavefront workgroup per CU): same answer
        opt("ivf_nqg", "2")
        ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
        same(ids, dis, oi, od)
        ids, dis = ix.search(q, k, "nprobe=%d" % nprobe, alive=alive)
        same(ids, dis, oa, oda)
        opt("ivf_nqg", None)
    else:
        # every tile size that fits LDS: same answer
        for ncb in (1, 2, 3, 4):
            opt("h16_ncb", str(ncb))
            ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
            same(ids, dis, oi, od)
        opt("h16_ncb", None)
    # candidate buffers far too small: overflowing queries lose their certificate and take the fallback, same answer
    opt("cand_cap", "64")
    ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
    same(ids, dis, oi, od)
    opt("cand_cap", None)
    q2 = capi.prefilter_stats()[0]
    # the pass switched off: the list-batched canonical scan, same answer
    opt("ivf_pass", "0")
    ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
    same(ids, dis, oi, od)
    assert capi.prefilter_stats()[0] == q2


def h16_keys(nq, cap=4096):
    """Candidate keys (approximate distance word << 32 | stored row position) and counts of this thread's last shadow pass."""
    import ctypes as C

    keys = np.zeros((nq, cap), np.uint64)
    cnt = np.zeros(nq, np.uint32)
    rc = capi.lib().msvs_debug_h16_keys(keys.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_size_t(cap), cnt.ctypes.data_as(C.POINTER(C.c_uint32)),
                                        C.c_size_t(nq))
    assert rc == 0, capi.lib().msvs_last_error()
    return keys, cnt


@pytest.mark.parametrize("metric", [capi.METRIC_L2, capi.METRIC_IP, capi.METRIC_COSINE])
@pytest.mark.parametrize("n,d,nlist,nq,nprobe,k", [(40000, 768, 16, 900, 4, 10),    # 12 chunks: the ring chains through the blocks
                                                   (30000, 500, 8, 300, 8, 10),     # 8 chunks
                                                   (30000, 384, 8, 2100, 8, 40),    # > 96 queries per list: several tiles, short last ones
                                                   (20000, 20, 8, 129, 3, 1),       # one chunk
                                                   (3000, 330, 4, 100, 4, 12),      # 6 chunks: no chaining, lists of ~750 rows
                                                   (12000, 1536, 4, 300, 2, 10),    # 24 chunks: one column block fits
                                                   (9000, 1400, 3, 150, 3, 5),      # 22 chunks
                                                   (2000, 512, 40, 700, 5, 10)])    # lists of ~50 rows: one or two blocks beyond the sample
def test_shadow_scan_items_of_every_size_match_oracle(metric, n, d, nlist, nq, nprobe, k, opt):
    """h16_scan_kernel's work items hold 1 .. 32 NCB probing queries and load / multiply only the column blocks they have queries
    for; whatever the tile the planner or the knob picks (ncb 1 .. 4: the same lists cut into more or fewer items) and whatever
    the grid (2 workgroups walk every item in turn; 1024: more workgroups than items), with no cut at all (every probed row
    is appended), tiny candidate buffers (overflow -> fallback): the canonical answer bit for bit, and the SAME candidate sets
    (every block of every list scanned exactly once per probing query, with and without row segments)."""
    rng = np.random.default_rng(n + d + nlist + 11)
    centers = rng.standard_normal((nlist, d), dtype=np.float32) * 2
    x = (centers[rng.integers(0, nlist, n)] + rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    q = (centers[rng.integers(0, nlist, nq)] + rng.standard_normal((nq, d), dtype=np.float32)).astype(np.float32)
    ix = build_ivf(x, metric, nlist)
    oi, od, _ = oracle_on_exported(ix, q, nprobe, k, metric)
    opt("ivf_pass", "2")
    opt("h16_prune", "2")  # (whether the pruning runs depends on the tile size, and with it the rank of the cut: pinned, so that the
    #                         candidate SETS can be compared across the configurations below)
    ref_keys = None
    # (h16_segs: lists cut into row segments sized on the device -- 0 never, 1 when the launch could run short of items, 2 always)
    for grid, ncb, segs in ((0, 0, 1), (0, 1, 2), (2, 2, 0), (1024, 3, 2), (37, 4, 2)):
        opt("h16_grid", str(grid))
        opt("h16_ncb", str(ncb))
        opt("h16_segs", str(segs))
        q0, f0 = capi.prefilter_stats()
        ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
        same(ids, dis, oi, od)
        q1, f1 = capi.prefilter_stats()
        assert q1 - q0 == nq and f1 - f0 <= nq // 4  # the candidate pass ran and certified (almost) everybody
        keys, cnt = h16_keys(nq)
        sets = [sorted(keys[i][: cnt[i]].tolist()) if cnt[i] <= keys.shape[1] else None for i in range(0, nq, max(1, nq // 16))]
        if ref_keys is None:
            ref_keys, ref_cnt = sets, cnt
        else:
            assert (cnt == ref_cnt).all() and sets == ref_keys
    opt("h16_grid", "1024")
    opt("h16_nocut", "1")  # every probed row a candidate
    ids, dis = ix.search(q[:64], k, "nprobe=%d" % nprobe)
    same(ids, dis, oi[:64], od[:64])
    opt("h16_nocut", None)
    opt("cand_cap", "64")
    ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
    same(ids, dis, oi, od)
    opt("cand_cap", None)
    opt("ivf_eps_scale", "1e12")
    ids, dis = ix.search(q[:100], k, "nprobe=%d" % nprobe)
    same(ids, dis, oi[:100], od[:100])
    opt("ivf_eps_scale", None)
    # a filter: the register kernel has no bit test, the LDS-tile kernel serves it -- same answer
    alive = rng.random(n) < 0.5
    ids, dis = ix.search(q, k, "nprobe=%d" % nprobe, alive=alive)
    oa, oda, _ = oracle_on_exported(ix, q, nprobe, k, metric, alive=alive)
    same(ids, dis, oa, oda)


@pytest.mark.parametrize("h16", ["1", "0"])
@pytest.mark.parametrize("metric", [capi.METRIC_L2, capi.METRIC_IP, capi.METRIC_COSINE])
def test_coarse_quantiser_through_the_candidate_pass(metric, h16, opt):
    """nlist >= 256 and >= 512 queries: the top-nprobe centroids come from the matrix-core pass (over the centroid
    shadow, h16 = 1, or the split-bf16 table pass) + canonical re-rank; the probe SETS must equal the exact scan's,
    hence so must the final answer (with and without certificates).  nlist = 330: the last 32-centroid block is partial."""
    opt("coarse_h16", h16)
    rng = np.random.default_rng(2025)
    n, d, nlist, nq, nprobe, k = 40000, 64, 330, 600, 12, 10
    centers = rng.standard_normal((nlist, d), dtype=np.float32) * 2
    x = (centers[rng.integers(0, nlist, n)] + rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    q = (centers[rng.integers(0, nlist, nq)] + rng.standard_normal((nq, d), dtype=np.float32)).astype(np.float32)
    ix = build_ivf(x, metric, nlist)
    oi, od, _ = oracle_on_exported(ix, q, nprobe, k, metric)
    opt("coarse_mfma", "2")  # whenever eligible (by default only from ~128 tile x slice items on)
    capi.profile_reset()
    capi.profile_enable(True)
    c0 = capi.coarse_stats()
    ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
    capi.profile_enable(False)
    assert capi.profile_get("coarse_pass")[0] == 1  # the pass is the one that ran
    capi.profile_reset()
    same(ids, dis, oi, od)
    c1 = capi.coarse_stats()
    assert c1[0] - c0[0] == nq and c1[1] - c0[1] <= nq // 20  # msvs_coarse_stats: queries through the pass, few fallbacks
    opt("ivf_eps_scale", "1e12")  # every certificate fails: canonical fallbacks everywhere
    ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
    same(ids, dis, oi, od)
    assert capi.coarse_stats()[1] - c1[1] == nq
    opt("fb_cap", "3")
    ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
    same(ids, dis, oi, od)
    opt("fb_cap", None)
    opt("ivf_eps_scale", None)
    opt("coarse_mfma", "0")
    ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
    same(ids, dis, oi, od)


@pytest.mark.parametrize("knobs", [{}, {"wave_select": "0"}, {"wave_select": "3"}, {"plan_lds": "0"}, {"fb_segs": "4"}, {"fb_segs": "16"},
                                   {"cand_cap": "3000"}, {"rerank_early": "0"}, {"rerank_groups": "32"}])
def test_selection_plan_and_fallback_variants_agree_with_the_oracle(knobs, opt):
    """The radix / bitwise (3) wave selection (candidate buffers and centroid words held in registers), the LDS-aggregated
    plan, the 16-segment fallback and the early exit of the re-rank (64 candidates: the second half is skipped when beyond the
    exact k-th of the first) against their insertion / per-pair-atomic / 4-segment / evaluate-everything forms: the same exact answer.  Massive
    duplicates (every approximate value shared by several rows) exercise the tie branch of the selection; cand_cap 3000
    is above the register form's capacity (the insertion kernel serves it)."""
    for name, v in knobs.items():
        opt(name, v)
    rng = np.random.default_rng(515)
    n, d, nlist, nq, nprobe, k = 50000, 96, 300, 700, 16, 10
    centers = rng.standard_normal((nlist, d), dtype=np.float32) * 2
    x = (centers[rng.integers(0, nlist, n)] + rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    x[n // 2:] = x[: n - n // 2]  # every row twice: equal approximate AND canonical distances, ids break the ties
    q = (centers[rng.integers(0, nlist, nq)] + rng.standard_normal((nq, d), dtype=np.float32)).astype(np.float32)
    for metric in (capi.METRIC_L2, capi.METRIC_IP):
        ix = build_ivf(x, metric, nlist)
        oi, od, _ = oracle_on_exported(ix, q, nprobe, k, metric)
        opt("coarse_mfma", "2")
        q0, f0 = capi.prefilter_stats()
        ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
        same(ids, dis, oi, od)
        assert capi.prefilter_stats()[0] > q0  # the candidate pass ran
        opt("ivf_eps_scale", "1e12")  # every certificate fails: the fallback in its segments
        ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
        same(ids, dis, oi, od)
        opt("ivf_eps_scale", None)
        # k > 12: 64 candidates per query
        oi, od, _ = oracle_on_exported(ix, q[:520], nprobe, 40, metric)
        ids, dis = ix.search(q[:520], 40, "nprobe=%d" % nprobe)
        same(ids, dis, oi, od)


@pytest.mark.parametrize("knobs", [{}, {"coarse_few": "0"}, {"plan_fused": "0"}, {"host_pinned": "0"}, {"lat_select": "3"}, {"coarse_dense": "0"},
                                   {"host_signal_batch": "0"},
                                   {"coarse_few": "0", "plan_fused": "0", "host_pinned": "0"}])
@pytest.mark.parametrize("metric", [capi.METRIC_L2, capi.METRIC_IP, capi.METRIC_COSINE])
def test_small_batches_one_launch_coarse_and_plan_match_oracle(metric, knobs, opt):
    """Batches of 3 .. 130 queries through msvs_index_search (the combined batches of concurrent single-query callers): the
    one-launch coarse quantiser (coarse_dense_kernel: every key of a table of <= 2048 centroids; coarse_few_kernel: a top-nprobe
    list per block -- the 2100-list index and coarse_dense = 0; one and four queries per block), the one-launch pair grouping (the fused plan
    kernel: up to 8192 pairs, beyond it the three-launch form) and the pinned query / result block, each against its multi-launch /
    staged form: the oracle's ids and distances whichever runs.  Duplicate centroid distances (two copies of every row) put ties
    into the probe selection; 1100 lists exercise the second round of the plan's scan (> 1024 lists) and more than 16 lists per
    query group of the coarse selection."""
    for name, v in knobs.items():
        opt(name, v)
    opt("lat_path", "0")  # (the batches of 1 - 2 queries below go through the general path too)
    rng = np.random.default_rng(4711)
    for n, d, nlist, nprobe, k in ((60000, 96, 300, 16, 10), (40000, 768, 64, 32, 10), (70000, 40, 1100, 64, 40), (5000, 20, 8, 8, 1), (60000, 24, 2100, 20, 10)):
        centers = rng.standard_normal((nlist, d), dtype=np.float32) * 2
        x = (centers[rng.integers(0, nlist, n)] + rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
        x[n // 2:] = x[: n - n // 2]
        ix = build_ivf(x, metric, nlist)
        for nq in (1, 3, 4, 5, 16, 17, 33, 64, 130, 300):
            q = (centers[rng.integers(0, nlist, nq)] + rng.standard_normal((nq, d), dtype=np.float32)).astype(np.float32)
            oi, od, _ = oracle_on_exported(ix, q, nprobe, k, metric)
            # the multi-launch / staged forms first: the queries that lose their certificate there ...
            for name in ("coarse_few", "plan_fused", "host_pinned"):
                opt(name, "0")
            f0 = capi.prefilter_stats()[1]
            ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
            same(ids, dis, oi, od)
            f_multi = capi.prefilter_stats()[1] - f0
            for name in ("coarse_few", "plan_fused", "host_pinned"):
                opt(name, knobs.get(name))
            for _ in range(2):  # (the second call reuses the arenas and the arrival counters of the first)
                f0 = capi.prefilter_stats()[1]
                ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
                same(ids, dis, oi, od)
                # ... are the ones that lose it here: the same probes, distances to the centroids and bounds reach the list scan (a
                # wrong probe distance would still end in the right rows -- through the canonical fallback of every query)
                assert capi.prefilter_stats()[1] - f0 == f_multi
        ix.close()


@pytest.mark.parametrize("metric", [capi.METRIC_L2, capi.METRIC_IP, capi.METRIC_COSINE])
def test_round6_batch_forms_match_the_oracle_and_their_round5_forms(metric, opt):
    """Round 6's launch-level changes of a batched search (>= 256 queries: the coarse pass through the centroid shadow), each against
    the form it replaces AND the oracle: coarse_tail_kernel (selection + band in one launch, a wavefront per query; queries whose band
    cannot be formed take the block-per-query re-rank through RerankParams::qmap) vs coarse_select_kernel + ivf_rerank_kernel;
    coarse_gemm_kernel vs coarse_h16_kernel; the second chance inside the re-rank launch vs a launch of its own; grouped appends vs
    one atomic per record; the plan feedback on / off.  Ties in the centroid distances (every centroid twice), lists that are not a
    multiple of 32, more than 1024 lists.  The queries that reach the canonical fallback are the same whichever form runs (the same
    probes and bounds reach the list scan); with the error bound blown up every query's band fails and all of them take the queue."""
    rng = np.random.default_rng(606)
    for n, d, nlist, nprobe, k, nq in ((60000, 96, 300, 16, 10, 700), (50000, 64, 1100, 32, 10, 400), (30000, 768, 256, 24, 10, 300)):
        half = nlist // 2
        centers = rng.standard_normal((half, d), dtype=np.float32) * 2
        centers = np.concatenate([centers, centers[: nlist - half]])  # duplicate centroids: ties at every rank of the probe selection
        x = (centers[rng.integers(0, nlist, n)] + rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
        ix = capi.Index(capi.INDEX_IVFFLAT, metric, d, "ncentroids=%d" % nlist)
        ix.set_centroids(centers)
        ix.add(x)
        ix.build()
        q = (centers[rng.integers(0, nlist, nq)] + rng.standard_normal((nq, d), dtype=np.float32)).astype(np.float32)
        oi, od, _ = oracle_on_exported(ix, q, nprobe, k, metric)
        counts = {}
        for name, knobs in (("round6", {}), ("select+rerank", {"coarse_tail": "0"}), ("wave tiles", {"coarse_h16": "3"}),
                            ("second chance launch", {"rerank_fused": "0"}), ("atomic per record", {"h16_group_appends": "0"}),
                            ("no feedback", {"h16_feedback": "0"}), ("round5", {"coarse_tail": "0", "coarse_h16": "3", "rerank_fused": "0",
                                                                                 "h16_group_appends": "0", "h16_feedback": "0", "pinned_fetch": "0"})):
            for kn, v in knobs.items():
                opt(kn, v)
            for rep in range(2):  # (the second search of a shape runs with the first one's plan feedback)
                c0, f0 = capi.coarse_stats(), capi.prefilter_stats()
                ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
                same(ids, dis, oi, od)
                c1, f1 = capi.coarse_stats(), capi.prefilter_stats()
                counts.setdefault(rep, {})[name] = (c1[0] - c0[0], c1[1] - c0[1], f1[1] - f0[1])
            for kn in knobs:
                opt(kn, None)
        for rep in (0, 1):
            assert len(set(counts[rep].values())) == 1, counts[rep]
            if metric != capi.METRIC_COSINE:  # (a cosine index ranks its centroids canonically: no shadow coarse pass to compare)
                assert counts[rep]["round6"][0] == nq  # the shadow coarse pass is the one that ran
        # nobody's band can be formed: every query is computed exactly -- by its own wavefront inside the tail launch while the index has
        # not met such a query for `coarse_slow_window` searches (window 0: always), through the queue to the block-per-query re-rank
        # and on to the canonical fallback otherwise (coarse_slow_inline = 0: always); the default takes the first form once, then the second
        opt("ivf_eps_scale", "1e12")
        for knobs in ({"coarse_slow_window": "0"}, {"coarse_slow_inline": "0"}, {}, {}):
            for kn, v in knobs.items():
                opt(kn, v)
            c0 = capi.coarse_stats()
            ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
            same(ids, dis, oi, od)
            c1 = capi.coarse_stats()
            if metric != capi.METRIC_COSINE:
                assert c1[1] - c0[1] == nq, knobs
            for kn in knobs:
                opt(kn, None)
        opt("ivf_eps_scale", None)
        ix.close()


@pytest.mark.parametrize("metric", [capi.METRIC_L2, capi.METRIC_COSINE])
def test_shadow_pass_serves_k_up_to_128_with_256_candidates(metric, opt):
    """40 < k <= 128 (a hybrid search takes the vector top-100): the fp16-shadow pass with 256 candidates per query, the
    insertion select with four registers per lane, the re-rank's four-register top-k and its early exit after the first
    ceil(k / 16) rounds; forced certificate failures (canonical fallback with k = 100); h16_k128 = 0 = the canonical scan."""
    rng = np.random.default_rng(4100)
    n, d, nlist, nq, nprobe = 60000, 64, 64, 320, 8
    centers = rng.standard_normal((nlist, d), dtype=np.float32) * 2
    x = (centers[rng.integers(0, nlist, n)] + rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    q = (centers[rng.integers(0, nlist, nq)] + rng.standard_normal((nq, d), dtype=np.float32)).astype(np.float32)
    ix = build_ivf(x, metric, nlist)
    for k in (41, 100, 128):
        oi, od, _ = oracle_on_exported(ix, q, nprobe, k, metric)
        q0, f0 = capi.prefilter_stats()
        ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
        same(ids, dis, oi, od)
        q1, f1 = capi.prefilter_stats()
        assert q1 - q0 == nq and f1 - f0 <= nq // 10  # the candidate pass ran, few fallbacks
        if k == 100:
            opt("ivf_eps_scale", "1e12")
            ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
            same(ids, dis, oi, od)
            assert capi.prefilter_stats()[1] - f1 == nq
            opt("ivf_eps_scale", None)
            opt("rerank_early", "0")
            ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
            same(ids, dis, oi, od)
            opt("rerank_early", None)
            opt("h16_k128", "0")
            q2 = capi.prefilter_stats()[0]
            ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
            same(ids, dis, oi, od)
            assert capi.prefilter_stats()[0] == q2  # canonical path
            opt("h16_k128", None)
    # k = 129: beyond the pass
    q2 = capi.prefilter_stats()[0]
    oi, od, _ = oracle_on_exported(ix, q[:40], nprobe, 129, metric)
    ids, dis = ix.search(q[:40], 129, "nprobe=%d" % nprobe)
    same(ids, dis, oi, od)
    assert capi.prefilter_stats()[0] == q2


@pytest.mark.parametrize("metric", [capi.METRIC_L2, capi.METRIC_IP, capi.METRIC_COSINE])
@pytest.mark.parametrize("data", ["blobs", "iid", "tied_centroids"])
def test_coarse_band_decides_most_probes_without_reading_the_centroids(metric, data, opt):
    """The coarse quantiser of a batch (centroid shadow -> 64 candidates -> exact top-nprobe): candidates whose approximate
    value puts them certainly inside / outside the top-nprobe are not evaluated canonically (coarse_band, default), only the
    band around the boundary is.  The probe SET must be the oracle's whatever the data does to the band: well separated
    centroids (band of a few rows), iid centroids (band = most candidates), groups of IDENTICAL centroids (the boundary falls
    inside a tie: decided by id).  Results == the oracle's with the band on and off."""
    rng = np.random.default_rng({"blobs": 5, "iid": 6, "tied_centroids": 7}[data] + metric)
    n, d, nlist, nq, nprobe, k = 40000, 96, 256, 512, 20, 10
    if data == "blobs":
        centres = 3.0 * rng.standard_normal((nlist, d), dtype=np.float32)
        x = (centres[rng.integers(0, nlist, n)] + rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
        q = (centres[rng.integers(0, nlist, nq)] + rng.standard_normal((nq, d), dtype=np.float32)).astype(np.float32)
        ix = build_ivf(x, metric, nlist)
    else:
        x = rng.standard_normal((n, d), dtype=np.float32)
        q = rng.standard_normal((nq, d), dtype=np.float32)
        if data == "iid":
            ix = build_ivf(x, metric, nlist)
        else:
            base = rng.standard_normal((16, d), dtype=np.float32)  # 16 distinct centroids, each 16 times
            cent = np.repeat(base, nlist // 16, axis=0)[rng.permutation(nlist)]
            ix = capi.Index(capi.INDEX_IVFFLAT, metric, d, "ncentroids=%d" % nlist)
            ix.set_centroids(cent if metric != capi.METRIC_COSINE else o.normalize_rows(cent))
            ix.add(x)
            ix.build()
    oi, od, _ = oracle_on_exported(ix, q, nprobe, k, metric)
    for band in ("1", "0"):
        opt("coarse_band", band)
        c0 = capi.coarse_stats()
        ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
        same(ids, dis, oi, od)
        c1 = capi.coarse_stats()
        assert c1[0] - c0[0] == nq, "the batch did not go through the centroid-shadow pass"
    opt("coarse_band", None)


@pytest.mark.parametrize("metric", [capi.METRIC_L2, capi.METRIC_COSINE, capi.METRIC_IP])
@pytest.mark.parametrize("data", ["blobs", "iid", "outlier_rows"])
def test_probe_pruning_drops_pairs_and_keeps_the_oracle_result(data, metric, opt):
    """The shadow list scan of a batch drops (query, list) pairs that provably cannot hold one of the query's k nearest rows
    (triangle inequality with the list radius against the k-th best sample row: H16Prune; L2 and cosine indexes; inner-product
    indexes through <q, x> <= <q, c> + |q| r).  Whatever it drops, the result is the
    oracle's, which scans every probed list: well separated blobs (most pairs go), iid rows (nothing can be proved: nothing goes),
    lists with a far outlier row each (huge radii: nothing goes, and the outliers are still found by the queries placed on
    them); with and without a filter; == the same search with the pruning off."""
    rng = np.random.default_rng({"blobs": 11, "iid": 12, "outlier_rows": 13}[data])
    n, d, nlist, nq, nprobe, k = 60000, 64, 256, 600, 16, 10
    if data == "iid":
        x = rng.standard_normal((n, d), dtype=np.float32)
        q = rng.standard_normal((nq, d), dtype=np.float32)
    else:
        centres = 4.0 * rng.standard_normal((nlist, d), dtype=np.float32)
        z = rng.integers(0, nlist, n)
        x = (centres[z] + rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
        q = (centres[rng.integers(0, nlist, nq)] + rng.standard_normal((nq, d), dtype=np.float32)).astype(np.float32)
        if data == "outlier_rows":
            far = rng.choice(n, 300, replace=False)
            x[far] += 40.0 * rng.standard_normal((300, d), dtype=np.float32)  # rows far from every centroid
            q[:100] = x[far[:100]] + 0.01 * rng.standard_normal((100, d), dtype=np.float32)  # ... and queries sitting on them
    ix = build_ivf(x, metric, nlist)
    alive = rng.random(n) < 0.5
    opt("rerank_stats", "1")
    for al in (None, alive):
        oi, od, _ = oracle_on_exported(ix, q, nprobe, k, metric, alive=al)
        opt("h16_prune", "2")  # always (the default only prunes when a list is probed by more queries than one tile holds)
        s0 = capi.debug_prune_stats()
        p0 = capi.prefilter_stats()
        ids, dis = ix.search(q, k, "nprobe=%d" % nprobe, alive=al)
        same(ids, dis, oi, od)
        s1 = capi.debug_prune_stats()
        assert capi.prefilter_stats()[0] - p0[0] == nq, "the batch did not go through the shadow pass"
        assert s1[1] - s0[1] == nq * nprobe, "the pruning did not look at the batch"
        dropped = (s1[0] - s0[0]) / float(nq * nprobe)
        if data == "blobs":
            # (5 k-means iterations leave merged and freshly split blobs: wide lists; inner product: |q| r is a loose bound next to <q, c>)
            assert dropped > {capi.METRIC_L2: 0.3, capi.METRIC_COSINE: 0.02, capi.METRIC_IP: 0.0}[metric], dropped
        if data == "iid":
            assert dropped < 0.05, dropped
        opt("h16_prune", "0")
        ids, dis = ix.search(q, k, "nprobe=%d" % nprobe, alive=al)
        same(ids, dis, oi, od)
    # a hybrid search's top-100 (k > 64: the bound takes a selection of its own) and a filter that leaves most queries fewer than
    # k live sample rows (no bound for those: nothing dropped, no cut "none" either)
    opt("h16_prune", "2")
    oi, od, _ = oracle_on_exported(ix, q[:200], nprobe, 100, metric)
    ids, dis = ix.search(q[:200], 100, "nprobe=%d" % nprobe)
    same(ids, dis, oi, od)
    sparse = rng.random(n) < 0.02
    oi, od, _ = oracle_on_exported(ix, q, nprobe, k, metric, alive=sparse)
    ids, dis = ix.search(q, k, "nprobe=%d" % nprobe, alive=sparse)
    same(ids, dis, oi, od)
    # the pre-pruning (list radius alone, before the sample launch; L2 and -- through ||q - c||^2 = |q|^2 + |c|^2 - 2 <q, c> -- cosine,
    # unfiltered) on and off: the same answer; a small batch takes its centroid distances from the canonical coarse scan
    if metric in (capi.METRIC_L2, capi.METRIC_COSINE):
        oi, od, _ = oracle_on_exported(ix, q, nprobe, k, metric)
        for pre in ("0", "1"):
            opt("h16_preprune", pre)
            s0 = capi.debug_prune_stats()
            ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
            same(ids, dis, oi, od)
            s1 = capi.debug_prune_stats()
            assert s1[1] - s0[1] == nq * nprobe
            ids, dis = ix.search(q[:40], k, "nprobe=%d" % nprobe)
            same(ids, dis, oi[:40], od[:40])
            s2 = capi.debug_prune_stats()
            if data == "blobs" and pre == "1":
                assert s2[0] - s1[0] > 40 * nprobe // 4, "the small batch's pre-pruning dropped nothing: %d of %d" % (s2[0] - s1[0], 40 * nprobe)
        opt("ivf_eps_scale", "1e12")  # every query takes the canonical fallback -- over the pre-pruned probe lists
        ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
        same(ids, dis, oi, od)
        opt("ivf_eps_scale", None)
        opt("h16_preprune", None)
    opt("h16_prune", None)
    opt("rerank_stats", None)


def test_second_chance_rerank_of_the_whole_candidate_buffer(opt):
    """Distances that concentrate (gaussian blobs of sigma 0.3 in a few hundred dimensions: the 10th and the 32nd neighbour of a
    query are a couple of error bounds apart) fail the first certificate -- k-th exact distance against the kc-th approximate
    one -- for a good share of the queries.  The second chance re-ranks every row of a failed query's candidate buffer and
    certifies against the cut: almost nobody is left for the canonical scan, and the answer is the canonical one either way."""
    rng = np.random.default_rng(303)
    n, d, blobs, nq, nprobe, k = 100000, 768, 96, 400, 4, 10
    centres = rng.standard_normal((blobs, d), dtype=np.float32)
    x = (centres[rng.integers(0, blobs, n)] + 0.3 * rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    q = (centres[rng.integers(0, blobs, nq)] + 0.3 * rng.standard_normal((nq, d), dtype=np.float32)).astype(np.float32)
    ix = build_ivf(x, capi.METRIC_L2, blobs)
    oi, od, _ = oracle_on_exported(ix, q, nprobe, k, capi.METRIC_L2)
    opt("ivf_pass", "2")
    opt("h16_rho", "0")  # the worst-case error model: with the measured one (round 4) most of these queries pass the first stage
    opt("rerank_second", "0")
    q0, f0 = capi.prefilter_stats()
    ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
    same(ids, dis, oi, od)
    q1, f1 = capi.prefilter_stats()
    assert q1 - q0 == nq
    first_stage_failures = f1 - f0
    opt("rerank_second", None)
    ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
    same(ids, dis, oi, od)
    q2, f2 = capi.prefilter_stats()
    assert q2 - q1 == nq
    assert first_stage_failures >= nq // 50, "the data no longer provokes first-stage failures: the test tests nothing"
    assert f2 - f1 <= first_stage_failures // 5
    # ... the same without the first stage's k-th distance as a skip bound (every row of the buffer evaluated), and for inner product
    opt("rerank_hint", "0")
    ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
    same(ids, dis, oi, od)
    q3, f3 = capi.prefilter_stats()
    assert f3 - f2 == f2 - f1  # the bound only saves row reads: the same queries get their certificate
    opt("rerank_hint", None)
    ixp = build_ivf(x, capi.METRIC_IP, blobs)
    opi, opd, _ = oracle_on_exported(ixp, q[:200], nprobe, k, capi.METRIC_IP)
    ids, dis = ixp.search(q[:200], k, "nprobe=%d" % nprobe)
    same(ids, dis, opi, opd)
    # overflowing buffers cannot be certified by the second chance either: canonical scan, same answer
    opt("cand_cap", "64")
    ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
    same(ids, dis, oi, od)
    opt("cand_cap", None)
    # k = 100 through the 256-candidate form
    oi, od, _ = oracle_on_exported(ix, q[:200], nprobe, 100, capi.METRIC_L2)
    ids, dis = ix.search(q[:200], 100, "nprobe=%d" % nprobe)
    same(ids, dis, oi, od)


def test_matrix_core_pass_with_massive_ties_and_unusable_norms():
    rng = np.random.default_rng(77)
    n, d, nlist, nq = 6000, 48, 4, 128
    base = rng.standard_normal((40, d), dtype=np.float32)
    x = base[rng.integers(0, 40, n)].copy()  # 40 distinct vectors: every top-10 is one big tie broken by id
    q = (base[rng.integers(0, 40, nq)] + 0.01 * rng.standard_normal((nq, d), dtype=np.float32)).astype(np.float32)
    for metric in (capi.METRIC_L2, capi.METRIC_IP):
        ix = build_ivf(x, metric, nlist)
        q0, f0 = capi.prefilter_stats()
        ids, dis = ix.search(q, 10, "nprobe=2")
        oi, od, _ = oracle_on_exported(ix, q, 2, 10, metric)
        same(ids, dis, oi, od)
        q1, f1 = capi.prefilter_stats()
        assert q1 - q0 == nq and f1 - f0 > 0  # ties cannot be certified: they went through the fallback
    # a NaN / huge row makes the error bound meaningless: the pass must switch itself off
    for bad in (np.nan, 3e19):
        y = rng.standard_normal((n, d), dtype=np.float32)
        y[123, 5] = bad
        ix = build_ivf(y, capi.METRIC_L2, nlist)
        q0, _ = capi.prefilter_stats()
        qq = rng.standard_normal((nq, d), dtype=np.float32)
        ids, dis = ix.search(qq, 10, "nprobe=2")
        oi, od, _ = oracle_on_exported(ix, qq, 2, 10, capi.METRIC_L2)
        same(ids, dis, oi, od)
        assert capi.prefilter_stats()[0] == q0


def test_ivfflat_structure_invariants_and_full_probe_equals_flat():
    rng = np.random.default_rng(11)
    n, d, nlist = 12000, 64, 48
    x = rng.standard_normal((n, d), dtype=np.float32)
    ids = (np.arange(n, dtype=np.int64) * 3 + 1)
    ix = build_ivf(x, capi.METRIC_L2, nlist, ids=ids)
    cent, off, vecs, lids = ix.export()
    assert off[0] == 0 and off[-1] == n and (np.diff(off) >= 0).all()
    assert sorted(lids.tolist()) == ids.tolist()  # a permutation of the fed ids
    for l in range(nlist):
        seg = lids[off[l]:off[l + 1]]
        assert (np.diff(seg) > 0).all()  # ascending ids inside a list
    back = {int(i): r for r, i in enumerate(lids)}
    pick = rng.integers(0, n, 200)
    assert all((vecs[back[int(ids[p])]] == x[p]).all() for p in pick)  # rows moved intact
    # every row sits in (one of) its nearest list(s): compare against the oracle's canonical assignment
    a = o.assign(x[pick], cent)
    got = np.searchsorted(off, [back[int(ids[p])] for p in pick], side="right") - 1
    assert (a == got).mean() > 0.97  # MFMA assignment may differ only on near-ties
    # probing every list == exhaustive search
    q = rng.standard_normal((9, d), dtype=np.float32)
    i1, d1 = ix.search(q, 10, "nprobe=%d" % nlist)
    oi, od = o.knn(q, x, 10, o.METRIC_L2, labels=ids)
    same(i1, d1, oi, od)


def test_ivfflat_recall_on_clustered_data():
    rng = np.random.default_rng(12)
    n, d, nlist = 50000, 64, 64
    centers = rng.standard_normal((nlist, d), dtype=np.float32) * 3
    x = (centers[rng.integers(0, nlist, n)] + 0.5 * rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    q = (centers[rng.integers(0, nlist, 50)] + 0.5 * rng.standard_normal((50, d), dtype=np.float32)).astype(np.float32)
    ix = build_ivf(x, capi.METRIC_L2, nlist)
    ids, _ = ix.search(q, 10, "nprobe=8")
    gt, _ = o.knn(q, x, 10, o.METRIC_L2, threads=8)
    recall = np.mean([len(set(a) & set(b)) / 10 for a, b in zip(ids.tolist(), gt.tolist())])
    assert recall >= 0.95


def test_index_errors_and_serialization_roundtrip():
    rng = np.random.default_rng(13)
    x = rng.standard_normal((4000, 32), dtype=np.float32)
    ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, 32, "ncentroids=16")
    with pytest.raises(capi.MsvsError) as e:
        ix.add(x)  # not trained
    assert e.value.code == capi.ERR_NOT_READY
    ix.train(x)
    ix.add(x)
    with pytest.raises(capi.MsvsError) as e:
        ix.search(x[:1], 5, "nprobe=4")  # not built
    assert e.value.code == capi.ERR_NOT_READY
    ix.build()
    with pytest.raises(capi.MsvsError) as e:
        ix.search(x[:1], 5, "alpha=3")  # unknown search parameter -> BAD_ARGUMENTS (00040 golden, serverError)
    assert e.value.code == capi.ERR_INVALID_ARGUMENT
    a = ix.search(x[:20], 5, "nprobe=4")
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "v1")  # the index name: files v1-data_bin.vidx3, v1-id_list.vidx3 (VICommon.h:55)
        ix.serialize(p)
        assert sorted(os.listdir(td)) == ["v1-data_bin.vidx3", "v1-id_list.vidx3"]
        ix2 = capi.Index.load(p, capi.INDEX_IVFFLAT, capi.METRIC_L2, 32)
        with open(os.path.join(td, "v1-data_bin.vidx3"), "r+b") as f:  # a truncated file is an IO error, not a crash
            f.truncate(os.path.getsize(f.name) - 100)
        with pytest.raises(capi.MsvsError) as e:
            capi.Index.load(p, capi.INDEX_IVFFLAT, capi.METRIC_L2, 32)
        assert e.value.code == capi.ERR_IO
    b = ix2.search(x[:20], 5, "nprobe=4")
    same(a[0], a[1], b[0], b[1])
    # the same through stream callbacks (what the host's VectorIndexWriter / VectorIndexReader over IDisk plug into)
    store = {}
    ix.serialize_io(store)
    assert sorted(store) == ["data_bin", "id_list"]
    mem, disk, build = ix.resource_usage()
    assert disk == sum(len(v) for v in store.values()) and mem >= 4000 * 32 * 4 and build >= mem // 2
    assert capi.index_version().startswith("msvs-")
    ix3 = capi.Index.load_io(store, capi.INDEX_IVFFLAT, capi.METRIC_L2, 32)
    c = ix3.search(x[:20], 5, "nprobe=4")
    same(a[0], a[1], c[0], c[1])
    # corrupt structure: offsets that are not a partition of the rows, ids outside the u32 range, a wrong header
    import struct
    hdr = struct.calcsize("<8sIiiIIIQQQ")
    for name, off, val in (("data_bin", hdr + 16 * 32 * 4 + 8, struct.pack("<q", 5000)),
                           ("id_list", 8, struct.pack("<q", 1 << 40)), ("data_bin", 12, struct.pack("<i", 9))):
        bad = {k: bytearray(v) for k, v in store.items()}
        bad[name][off:off + len(val)] = val
        with pytest.raises(capi.MsvsError) as e:
            capi.Index.load_io(bad, capi.INDEX_IVFFLAT, capi.METRIC_L2, 32)
        assert e.value.code == capi.ERR_IO
    with pytest.raises(capi.MsvsError):
        capi.Index(7, capi.METRIC_L2, 8)


def test_sharded_lists_merge_equals_unsharded():
    """lists sharded list_id % W across W index objects + canonical merge == the unsharded index."""
    rng = np.random.default_rng(14)
    n, d, nlist, W = 16000, 48, 32, 4
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((11, d), dtype=np.float32)
    full = build_ivf(x, capi.METRIC_IP, nlist)
    cent = full.export()[0]
    parts = []
    for r in range(W):
        ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_IP, d, "shard_rank=%d,shard_world=%d" % (r, W))
        ix.set_centroids(cent)
        ix.add(x)
        ix.build()
        parts.append(ix.search(q, 10, "nprobe=8"))
    assert sum(1 for _ in parts) == W
    mi, md = capi.merge_topk(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]), capi.METRIC_IP)
    fi, fd = full.search(q, 10, "nprobe=8")
    same(mi, md, fi, fd)
    # a batch large enough for the matrix-core candidate pass on every shard (most lists of a shard are empty)
    qb = rng.standard_normal((300, d), dtype=np.float32)
    q0 = capi.prefilter_stats()[0]
    shard_ix = []
    for r in range(W):
        ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_IP, d, "shard_rank=%d,shard_world=%d" % (r, W))
        ix.set_centroids(cent)
        ix.add(x)
        ix.build()
        shard_ix.append(ix)
    parts = [ix.search(qb, 10, "nprobe=8") for ix in shard_ix]
    assert capi.prefilter_stats()[0] - q0 == W * 300
    mi, md = capi.merge_topk(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]), capi.METRIC_IP)
    fi, fd = full.search(qb, 10, "nprobe=8")
    same(mi, md, fi, fd)
    oi, od, _ = oracle_on_exported(full, qb, 8, 10, capi.METRIC_IP)
    same(fi, fd, oi, od)


@pytest.mark.parametrize("metric", [capi.METRIC_L2, capi.METRIC_IP, capi.METRIC_COSINE])
@pytest.mark.parametrize("d", [768, 100, 30])
def test_few_query_path_equals_general_path_and_oracle(metric, d, opt):
    """latency_kernels.hpp (nq <= 4, k and nprobe <= 64: two self-merging launches, pinned-memory I/O) against the
    general path (lat_path = 0) and the oracle: host-pointer entry with padding / cosine preparation on the CPU, the
    device entry, filters, lists shorter than a segment, nprobe = nlist, repeated calls (counters reset themselves)."""
    import torch

    rng = np.random.default_rng(500 + d + metric)
    n, nlist = 30000, 96
    centers = rng.standard_normal((nlist, d), dtype=np.float32) * 2
    sizes = rng.integers(0, 2 * n // nlist, nlist)
    sizes[3] = 0
    sizes[7] = 1
    x = np.concatenate([centers[i] + rng.standard_normal((int(sizes[i]), d), dtype=np.float32) for i in range(nlist)]).astype(np.float32)
    q = (centers[rng.integers(0, nlist, 4)] + rng.standard_normal((4, d), dtype=np.float32)).astype(np.float32)
    ix = build_ivf(x, metric, nlist)
    alive = rng.random(len(x)) < 0.3
    for rep in range(2):
        for nq in (1, 2, 4):
            for k, nprobe in ((10, 8), (1, 1), (64, 64), (10, 200)):
                for al in (None, alive):
                    opt("lat_path", "2")
                    ids, dis = ix.search(q[:nq], k, "nprobe=%d" % nprobe, alive=al)
                    opt("lat_path", "0")
                    gi, gd = ix.search(q[:nq], k, "nprobe=%d" % nprobe, alive=al)
                    same(ids, dis, gi, gd)
                    if rep == 0:
                        oi, od, _ = oracle_on_exported(ix, q[:nq], nprobe, k, metric, alive=al)
                        same(ids, dis, oi, od)
    # the stream-ordered device entry (takes the two-launch path when the queries are scan-ready)
    qd = torch.from_numpy(q).cuda()
    oi_t = torch.empty((4, 10), device="cuda", dtype=torch.int64)
    od_t = torch.empty((4, 10), device="cuda", dtype=torch.float32)
    stream = torch.cuda.current_stream().cuda_stream
    for nq in (1, 3):
        for lp in ("2", "0"):
            opt("lat_path", lp)
            oi_t.fill_(-7)
            ix.search_device(qd.data_ptr(), nq, 10, 8, oi_t.data_ptr(), od_t.data_ptr(), stream)
            torch.cuda.synchronize()
            hi, hd = ix.search(q[:nq], 10, "nprobe=8")
            same(oi_t[:nq].cpu().numpy(), od_t[:nq].cpu().numpy(), hi, hd)


GEOMETRIES = ["outlier_neighbour", "ties_across_the_cut", "k_equals_list_length", "short_lists_only", "zero_norm_rows",
              "query_is_a_centroid"]


def _adversarial_lists(geometry, rng, d=64, nlist=64):
    """(centroids, rows, queries, k, nprobe) of a hand-made IVF structure (msvs_index_set_centroids: rows go to their nearest
    centroid) that aims at one assumption of the radius bounds each."""
    centres = (6.0 * rng.standard_normal((nlist, d))).astype(np.float32)
    sizes = rng.integers(150, 500, nlist)
    k, nprobe = 10, 16
    if geometry == "k_equals_list_length":
        sizes[::4] = k      # lists of exactly k rows: the smallest list that may give a bound
        sizes[1::8] = 3     # shorter than k: no bound from them
        sizes[2] = 0
    if geometry == "short_lists_only":
        sizes[:] = rng.integers(4, 31, nlist)  # every list shorter than one shadow block, k larger than every list
        k = 40
    x = np.concatenate([centres[i] + rng.standard_normal((int(sizes[i]), d)).astype(np.float32) for i in range(nlist) if sizes[i]]).astype(np.float32)
    q = (centres[rng.integers(0, nlist, 320)] + rng.standard_normal((320, d))).astype(np.float32)
    if geometry == "outlier_neighbour":
        # one far row per list sets the list's radius -- and is the nearest row of a query sitting next to it
        u = rng.standard_normal((nlist, d)).astype(np.float32)
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        far = (centres + 25.0 * u).astype(np.float32)
        x = np.concatenate([x, far]).astype(np.float32)
        q[:nlist] = far + 0.05 * rng.standard_normal((nlist, d)).astype(np.float32)
        q[nlist:2 * nlist] = (centres + 12.0 * u).astype(np.float32)  # half way out: the outlier is among the k nearest of few of them
    if geometry == "ties_across_the_cut":
        # for the first queries: 6 copies of the row that would be their 8th neighbour (ranks 8 .. 13 tie across k = 10), and a row
        # repeated in the rows of several lists
        for j in range(40):
            dist = np.linalg.norm(x - q[j], axis=1)
            r8 = int(np.argsort(dist)[7])
            x = np.concatenate([x, np.repeat(x[r8:r8 + 1], 5, axis=0)]).astype(np.float32)
        x[10:14] = x[len(x) - 1]
    if geometry == "zero_norm_rows":
        x[rng.choice(len(x), 200, replace=False)] = 0.0  # a cosine index keeps them unnormalised: |x| = 0
        q[5] = 0.0
    if geometry == "query_is_a_centroid":
        q[:nlist] = centres
        q[nlist:nlist + 8] = x[:8]  # ... and queries equal to stored rows
    return centres, x, q, k, nprobe


@pytest.mark.parametrize("metric", [capi.METRIC_L2, capi.METRIC_COSINE, capi.METRIC_IP])
@pytest.mark.parametrize("geometry", GEOMETRIES)
def test_radius_bounds_on_adversarial_geometry(geometry, metric, opt):
    """The pruning proofs (h16_scan_kernels.hpp: pre-pruning by the list radius, the sample-based second stage; latency_kernels.hpp:
    lat_cut) are only as good as their premises: a list whose radius is one far outlier that IS the query's neighbour, ties across
    the k-th rank, lists of exactly k rows / shorter than k / shorter than a shadow block, k larger than every probed list, zero
    rows in a cosine index, queries equal to centroids or stored rows.  A wrong drop is silent -- so: ids and distance bits of the
    pruned batch search, of the unpruned one and of the few-query path == the oracle's scan of every probed list, and the counters
    show that the pruning did look at the batch (and, where the geometry allows a proof, that it dropped pairs)."""
    rng = np.random.default_rng(7 + GEOMETRIES.index(geometry))
    centres, x, q, k, nprobe = _adversarial_lists(geometry, rng)
    d, nlist = x.shape[1], centres.shape[0]
    ix = capi.Index(capi.INDEX_IVFFLAT, metric, d, "ncentroids=%d" % nlist)
    ix.set_centroids(centres)
    ix.add(x)
    ix.build()
    oi, od, _ = oracle_on_exported(ix, q, nprobe, k, metric)
    opt("rerank_stats", "1")
    dropped = {}
    for prune, pre in (("2", "1"), ("2", "0"), ("0", "0")):
        opt("h16_prune", prune)
        opt("h16_preprune", pre)
        s0 = capi.debug_prune_stats()
        ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
        same(ids, dis, oi, od)
        s1 = capi.debug_prune_stats()
        dropped[(prune, pre)] = (s1[0] - s0[0], s1[1] - s0[1])
        for nqs in (40, 7):  # a small batch (canonical coarse scan feeds the pre-pruning), a handful
            ids, dis = ix.search(q[:nqs], k, "nprobe=%d" % nprobe)
            same(ids, dis, oi[:nqs], od[:nqs])
    # (lists shorter than k give no bound and the shadow pass may not be chosen at all; inner-product indexes of this size keep the
    # canonical scan: results only)
    if geometry != "short_lists_only" and metric != capi.METRIC_IP:
        assert dropped[("2", "1")][1] >= q.shape[0] * nprobe, "the pruning did not look at the batch: %r" % (dropped,)
        if metric == capi.METRIC_L2:
            assert dropped[("2", "1")][0] > 0, "well separated lists and nothing dropped: %r" % (dropped,)
    assert dropped[("0", "0")][0] == 0
    opt("ivf_eps_scale", "1e12")  # no certificate: the canonical fallback over the pre-pruned probe lists
    opt("h16_prune", "2")
    opt("h16_preprune", "1")
    ids, dis = ix.search(q[:64], k, "nprobe=%d" % nprobe)
    same(ids, dis, oi[:64], od[:64])
    opt("ivf_eps_scale", None)
    # one and two queries per call: the latency path with its radius cut (L2), and without
    opt("lat_path", "2")
    for lp in ("1", "0"):
        opt("lat_prune", lp)
        for j0 in (0, 2, 64, 65, 130):
            for nqc in (1, 2):
                ids, dis = ix.search(q[j0:j0 + nqc], k, "nprobe=%d" % nprobe)
                same(ids, dis, oi[j0:j0 + nqc], od[j0:j0 + nqc])
    for name in ("lat_prune", "lat_path", "h16_prune", "h16_preprune", "rerank_stats"):
        opt(name, None)


def test_few_query_path_radius_pruning_keeps_the_oracle_result(opt):
    """One or two queries per call on an L2 index: stage 1's last block drops the probed lists that the list radius rules out
    ((||q - c_l|| - r_l)^2 beyond the smallest (||q - c_p|| + r_p)^2 over probed lists of >= k rows) and cuts stage 2's work from
    the survivors.  Well separated blobs (almost every probe goes), lists shorter than k (they give no bound), k larger than any
    list, a query in the middle of nowhere, duplicates of a row across lists: == the unpruned path == the oracle, bit for bit."""
    rng = np.random.default_rng(4242)
    d, nlist = 96, 64
    centres = 6.0 * rng.standard_normal((nlist, d), dtype=np.float32)
    sizes = rng.integers(200, 900, nlist)
    sizes[5], sizes[9] = 3, 0  # a list shorter than k, an empty one
    x = np.concatenate([centres[i] + rng.standard_normal((int(sizes[i]), d), dtype=np.float32) for i in range(nlist)]).astype(np.float32)
    x[10:14] = x[len(x) - 1]  # the same row in several lists
    q = (centres[rng.integers(0, nlist, 6)] + rng.standard_normal((6, d), dtype=np.float32)).astype(np.float32)
    q[4] = centres[5] + 0.1 * rng.standard_normal(d).astype(np.float32)  # sits on the 3-row list
    q[5] = 40.0 * rng.standard_normal(d).astype(np.float32)  # far from everything
    ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, "ncentroids=%d" % nlist)
    ix.set_centroids(centres)
    ix.add(x)
    ix.build()
    opt("lat_path", "2")
    for k, nprobe in ((10, 16), (1, 64), (64, 8), (10, 1)):
        oi, od, _ = oracle_on_exported(ix, q, nprobe, k, capi.METRIC_L2)
        for nq0 in range(0, 6, 2):
            for nqc in (1, 2):
                for lp in ("1", "0"):
                    opt("lat_prune", lp)
                    ids, dis = ix.search(q[nq0:nq0 + nqc], k, "nprobe=%d" % nprobe)
                    same(ids, dis, oi[nq0:nq0 + nqc], od[nq0:nq0 + nqc])
    opt("lat_prune", None)
    opt("lat_path", None)


@pytest.mark.parametrize("lat_select", ["1", "0"])
def test_few_query_path_probe_ties_are_broken_by_centroid_id(lat_select, opt):
    """A zero query under inner product is equally far from every centroid: the probe list is the nprobe smallest centroid
    ids (the oracle's (distance, id) order) -- the register selection of the two-launch path resolves the tie group by a
    second search over the ids; the list-merge form (lat_select = 0) by the order of its keys."""
    opt("lat_select", lat_select)
    rng = np.random.default_rng(808)
    n, d, nlist = 20000, 64, 200
    x = rng.standard_normal((n, d), dtype=np.float32)
    ix = build_ivf(x, capi.METRIC_IP, nlist)
    q = np.zeros((2, d), np.float32)
    q[1] = rng.standard_normal(d).astype(np.float32)
    for nprobe in (1, 7, 32, 64):
        opt("lat_path", "2")
        ids, dis = ix.search(q, 10, "nprobe=%d" % nprobe)
        opt("lat_path", "0")
        gi, gd = ix.search(q, 10, "nprobe=%d" % nprobe)
        same(ids, dis, gi, gd)
        oi, od, _ = oracle_on_exported(ix, q, nprobe, 10, capi.METRIC_IP)
        same(ids, dis, oi, od)


def test_few_query_path_from_many_host_threads(opt):
    """Concurrent client threads (one non-blocking stream and one set of pinned buffers per host thread): every thread
    gets its own queries' results."""
    import threading

    rng = np.random.default_rng(77)
    n, d, nlist = 40000, 64, 64
    centers = rng.standard_normal((nlist, d), dtype=np.float32) * 2
    x = (centers[rng.integers(0, nlist, n)] + rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    q = (centers[rng.integers(0, nlist, 256)] + rng.standard_normal((256, d), dtype=np.float32)).astype(np.float32)
    ix = build_ivf(x, capi.METRIC_L2, nlist)
    opt("lat_path", "0")
    exp_i, exp_d = ix.search(q, 10, "nprobe=8")
    opt("lat_path", "1")
    errors = []

    def worker(t):
        try:
            for rep in range(3):
                for j in range(t, 256, 16):
                    i1, d1 = ix.search(q[j:j + 1], 10, "nprobe=8")
                    if not ((i1 == exp_i[j:j + 1]).all() and (d1.view(np.uint32) == exp_d[j:j + 1].view(np.uint32)).all()):
                        errors.append((t, j))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert errors == []


def test_concurrent_single_query_callers_are_combined_into_batches(opt):
    """msvs_index_search from 48 host threads, one query per call (the reference's calling pattern): callers beyond the 8
    direct ones are served by batches (msvs_combine_stats), every caller gets exactly the rows of its own query, whatever
    batch it landed in; different k / parameters never share a batch; an argument error reaches every caller of the batch
    in its own thread (msvs_last_error is thread-local); combine = 0 is the plain path."""
    import threading

    rng = np.random.default_rng(31)
    n, d, nlist = 40000, 96, 128
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((384, d), dtype=np.float32)
    ix = build_ivf(x, capi.METRIC_L2, nlist)
    opt("combine", "0")
    exp = {(10, 8): ix.search(q, 10, "nprobe=8"), (5, 16): ix.search(q, 5, "nprobe=16")}
    # native threads: real concurrency (python threads hold the GIL between calls and rarely have 8 calls in flight)
    opt("combine", None)
    c0 = capi.combine_stats()
    sec, lat, ci, cd = mhost.concurrent_search(ix, q, 48, 40, 10, "nprobe=8")
    same(ci, cd, *exp[(10, 8)])
    c1 = capi.combine_stats()
    assert c1[0] - c0[0] == 48 * 40
    assert c1[1] > c0[1] and c1[2] - c0[2] >= 2 * (c1[1] - c0[1])  # batches of several callers were formed
    # python threads with ONE direct slot: everybody else is served by batches; mixed k / parameters; errors
    opt("combine", "1")
    c0 = capi.combine_stats()
    errors = []

    def worker(t):
        try:
            k, nprobe = (10, 8) if t % 3 else (5, 16)
            ei, ed = exp[(k, nprobe)]
            for rep in range(4):
                for j in range(t, 384, 48):
                    i1, d1 = ix.search(q[j:j + 1], k, "nprobe=%d" % nprobe)
                    if not ((i1 == ei[j:j + 1]).all() and (d1.view(np.uint32) == ed[j:j + 1].view(np.uint32)).all()):
                        errors.append((t, j))
            try:
                ix.search(q[t:t + 1], 10, "bogus=1")
                errors.append((t, "no error for a bad parameter"))
            except capi.MsvsError as e:
                if e.code != capi.ERR_INVALID_ARGUMENT or "bogus" not in str(e):
                    errors.append((t, repr(e)))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(48)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert errors == []
    c1 = capi.combine_stats()
    assert c1[0] - c0[0] == 48 * (4 * 8 + 1)
    assert c1[1] > c0[1] and c1[2] - c0[2] >= 2 * (c1[1] - c0[1])  # batches of several callers were formed
    # two- and four-query calls take part too
    i2, d2 = ix.search(q[:4], 10, "nprobe=8")
    same(i2, d2, exp[(10, 8)][0][:4], exp[(10, 8)][1][:4])
    # the index goes (its worker thread with it); the next one -- possibly at the same address -- starts its own
    ix.close()
    for _ in range(2):
        ix2 = build_ivf(x[:20000], capi.METRIC_L2, 64)
        opt("combine", "0")
        e2 = ix2.search(q, 10, "nprobe=8")
        opt("combine", None)
        c0 = capi.combine_stats()
        sec, lat, ci, cd = mhost.concurrent_search(ix2, q, 32, 24, 10, "nprobe=8")
        same(ci, cd, *e2)
        assert capi.combine_stats()[1] > c0[1]
        ix2.close()


# ---------------------------------------------------------------------------------------- the certificate's premise, on hardware

def _key_values(hi, ip):
    o_ = (~hi if ip else hi).astype(np.uint32)
    bits = np.where(o_ & np.uint32(0x80000000), o_ & np.uint32(0x7FFFFFFF), ~o_).astype(np.uint32)
    return bits.view(np.float32)


ADVERSARIAL = {
    "gaussian": lambda r, n, d: r.standard_normal((n, d)),
    "near_duplicates_of_the_queries": lambda r, n, d: np.tile(r.standard_normal((1, d)) * 30, (n, 1)) + r.standard_normal((n, d)) * 1e-3,
    "wide_dynamic_range": lambda r, n, d: r.standard_normal((n, d)) * np.exp2(r.integers(-12, 12, (n, d))),
    "one_sign_long_sums": lambda r, n, d: np.abs(r.standard_normal((n, d))) + 1.0,
    "fp16_rounding_boundaries": lambda r, n, d: (r.integers(1024, 2048, (n, d)) * 2 + 1) * np.exp2(-11.0) * r.choice([-1.0, 1.0], (n, d)),
    "sparse_spikes": lambda r, n, d: np.where(r.random((n, d)) < 0.02, r.standard_normal((n, d)) * 1e3, 0.0) + 1e-3,
    "tiny_values": lambda r, n, d: r.standard_normal((n, d)) * 1e-6,
}


@pytest.mark.parametrize("name", sorted(ADVERSARIAL))
@pytest.mark.parametrize("metric,d", [(capi.METRIC_L2, 768), (capi.METRIC_IP, 768), (capi.METRIC_L2, 1024), (capi.METRIC_L2, 100)])
def test_mfma_accumulation_error_bound_on_hardware(name, metric, d, opt):
    """The certificate (mfma_scan_kernels.hpp / h16_scan_kernels.hpp) rests on a per-pair bound of what the fp16-shadow
    MFMA pass computes: |approximate - true| <= 2 c_dot |x||q| + c_norm (|x|^2 + |q|^2) for L2, c_dot |x||q| for IP, with
    the constants of set_error_model_h16.  Round 1 only checked a numpy EMULATION of the arithmetic; this reads the
    kernel's OWN approximate keys back (every probed row, h16_nocut) on adversarial inputs and measures the ratio."""
    import ctypes as C

    rng = np.random.default_rng(sorted(ADVERSARIAL).index(name) * 100 + metric * 10 + d)
    n, nlist, nq = 2048, 4, 64
    keys0 = np.zeros((nq, 1), np.uint64)
    capi.lib().msvs_debug_h16_keys(keys0.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_size_t(1), np.zeros(nq, np.uint32).ctypes.data_as(
        C.POINTER(C.c_uint32)), C.c_size_t(nq))  # drop the record of an earlier pass
    x = ADVERSARIAL[name](rng, n, d).astype(np.float32)
    q = ADVERSARIAL[name](rng, nq, d).astype(np.float32)
    if name == "near_duplicates_of_the_queries":
        q = (x[:nq] + rng.standard_normal((nq, d)).astype(np.float32) * 1e-3).astype(np.float32)
    ix = build_ivf(x, metric, nlist)
    opt("ivf_pass", "2")
    opt("h16_nocut", "1")
    opt("cand_cap", "16384")
    p0 = capi.prefilter_stats()
    ids, dis = ix.search(q, 10, "nprobe=%d" % nlist)
    p1 = capi.prefilter_stats()
    assert p1[0] - p0[0] == nq, "the shadow pass did not run"
    cap = 4096
    keys = np.zeros((nq, cap), np.uint64)
    cnt = np.zeros(nq, np.uint32)
    rc = capi.lib().msvs_debug_h16_keys(keys.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_size_t(cap), cnt.ctypes.data_as(C.POINTER(C.c_uint32)),
                                        C.c_size_t(nq))
    assert rc == 0, capi.lib().msvs_last_error()
    assert (cnt == n).all(), "every row of every probed list is a candidate"  # nprobe = nlist, no cut
    cd, cn, cc = C.c_double(), C.c_double(), C.c_double()
    capi.lib().msvs_debug_error_model_h16.restype = None
    capi.lib().msvs_debug_error_model_h16(C.c_size_t(d), C.byref(cd), C.byref(cn), C.byref(cc))
    _, _, vecs, _ = ix.export()  # storage order: the low word of a key is a position here
    ip = metric != capi.METRIC_L2
    worst = 0.0
    x64, q64 = vecs.astype(np.float64), q.astype(np.float64)
    xn = np.sqrt((x64 * x64).sum(1))
    for qi in range(nq):
        kk = keys[qi, :n]
        pos = (kk & np.uint64(0xFFFFFFFF)).astype(np.int64)
        approx = _key_values((kk >> np.uint64(32)).astype(np.uint32), ip).astype(np.float64)
        qn = np.sqrt((q64[qi] * q64[qi]).sum())
        if ip:
            true = x64[pos] @ q64[qi]
            eps = cd.value * xn[pos] * qn
        else:
            diff = x64[pos] - q64[qi]
            true = (diff * diff).sum(1)
            eps = 2 * cd.value * xn[pos] * qn + cn.value * (xn[pos] ** 2 + qn ** 2)
        ratio = np.abs(approx - true) / (eps + 1e-300)
        worst = max(worst, float(ratio.max()))
    assert worst < 1.0, "approximate keys leave the certified band: max |approx - true| / eps = %.3f" % worst
    # round 4: the bound the certificate really uses -- the MEASURED rounding error of the stored rows (h16_rho_kernel, at build) and
    # of each query image (h16_prep_queries_kernel) instead of 2^-11 per element: rho_x + rho_q + rho_x rho_q in place of 2^-10
    rho_t, cdt, qsc = C.c_double(), C.c_double(), C.c_double()
    capi.lib().msvs_debug_error_model_h16_measured.restype = None
    capi.lib().msvs_debug_error_model_h16_measured(ix._h, C.c_int(0), C.byref(rho_t), C.byref(cdt), C.byref(qsc))
    assert 0.0 <= rho_t.value < 2.0 ** -10 and qsc.value > 1.0

    def image_rho(v32):  # max over rows of |fp16(v s) / s - v| / |v|, one power-of-two scale for the whole array (numpy rounds like the device)
        mx = float(np.abs(v32).max())
        if mx == 0.0:
            return np.zeros(len(v32))
        s_ = np.float32(2.0 ** (14 - np.frexp(np.float32(mx))[1]))
        back = (v32 * s_).astype(np.float16).astype(np.float64) / float(s_)
        v64 = v32.astype(np.float64)
        den = np.sqrt((v64 * v64).sum(1))
        return np.where(den > 0, np.sqrt(((v64 - back) ** 2).sum(1)) / np.maximum(den, 1e-300), 0.0)

    rho_rows = image_rho(vecs)
    assert rho_rows.max() <= rho_t.value * (1 + 1e-9) and rho_rows.max() >= rho_t.value * (1 - 1e-5), (rho_rows.max(), rho_t.value)
    worst_m = 0.0
    for qi in range(nq):
        kk = keys[qi, :n]
        pos = (kk & np.uint64(0xFFFFFFFF)).astype(np.int64)
        approx = _key_values((kk >> np.uint64(32)).astype(np.uint32), ip).astype(np.float64)
        qn = np.sqrt((q64[qi] * q64[qi]).sum())
        cdq = cdt.value + qsc.value * float(image_rho(q[qi:qi + 1])[0]) * 1.000001
        if ip:
            true = x64[pos] @ q64[qi]
            eps = cdq * xn[pos] * qn
        else:
            diff = x64[pos] - q64[qi]
            true = (diff * diff).sum(1)
            eps = 2 * cdq * xn[pos] * qn + cn.value * (xn[pos] ** 2 + qn ** 2)
        worst_m = max(worst_m, float((np.abs(approx - true) / (eps + 1e-300)).max()))
    assert worst_m < 1.0, "approximate keys leave the MEASURED band: max |approx - true| / eps = %.3f" % worst_m
    print("h16 error: %.3f of the worst-case bound, %.3f of the measured one (rho_x = %.3f u)" % (worst, worst_m, rho_t.value * 2048))
    # and the results are exact all the same
    oi, od, _ = oracle_on_exported(ix, q, nlist, 10, metric)
    same(ids, dis, oi, od)


@pytest.mark.parametrize("metric", [capi.METRIC_L2, capi.METRIC_IP])
@pytest.mark.parametrize("name", sorted(ADVERSARIAL))
def test_centroid_shadow_error_bound_on_hardware(name, metric, opt):
    """The probe pruning and the coarse band rest on the COARSE pass's words -- coarse_h16_kernel over the centroid table's fp16
    shadow -- being within eps_c of the real centroid distances: |approximate - true| <= 2 c_dot |c||q| + c_norm (|c|^2 + |q|^2)
    (L2), c_dot |c||q| (inner product), the constants of set_error_model_h16.  Reads the kernel's OWN words back
    (msvs_debug_coarse_words) for adversarial centroids and queries and measures the ratio, like the row shadow's test above."""
    import ctypes as C

    d, nlist, nq, n = 200, 256, 256, 8192
    rng = np.random.default_rng(sorted(ADVERSARIAL).index(name) * 31 + metric)
    cent = ADVERSARIAL[name](rng, nlist, d).astype(np.float32)
    q = ADVERSARIAL[name](rng, nq, d).astype(np.float32)
    if name == "near_duplicates_of_the_queries":
        q = (cent[:nq] + rng.standard_normal((nq, d)).astype(np.float32) * 1e-3).astype(np.float32)
    # rows: the centroids' own magnitudes (the shadows share ONE scale taken from the rows: a centroid above it has no shadow)
    x = (cent[rng.integers(0, nlist, n)] * np.float32(1.001)).astype(np.float32)
    ix = capi.Index(capi.INDEX_IVFFLAT, metric, d, "ncentroids=%d" % nlist)
    ix.set_centroids(cent)
    ix.add(x)
    ix.build()
    c0 = capi.coarse_stats()
    ids, dis = ix.search(q, 5, "nprobe=8")
    assert capi.coarse_stats()[0] - c0[0] == nq, "the centroid-shadow pass did not run"
    npad = (nlist + 31) // 32 * 32
    words = np.zeros((nq, npad), np.uint32)
    fn = capi.lib().msvs_debug_coarse_words
    fn.argtypes = [C.POINTER(C.c_uint32), C.c_size_t, C.c_size_t]
    assert fn(words.ctypes.data_as(C.POINTER(C.c_uint32)), nq, npad) == 0, capi.lib().msvs_last_error()
    cd, cn, cc = C.c_double(), C.c_double(), C.c_double()
    capi.lib().msvs_debug_error_model_h16.restype = None
    capi.lib().msvs_debug_error_model_h16(C.c_size_t(d), C.byref(cd), C.byref(cn), C.byref(cc))
    ip = metric != capi.METRIC_L2
    c64, q64 = cent.astype(np.float64), q.astype(np.float64)
    cnorm, qnorm = np.sqrt((c64 * c64).sum(1)), np.sqrt((q64 * q64).sum(1))
    approx = _key_values(words[:, :nlist].reshape(-1), ip).astype(np.float64).reshape(nq, nlist)
    if ip:
        true = q64 @ c64.T
        eps = cd.value * np.outer(qnorm, cnorm)
    else:
        true = (q64 * q64).sum(1)[:, None] + (c64 * c64).sum(1)[None, :] - 2.0 * (q64 @ c64.T)
        eps = 2 * cd.value * np.outer(qnorm, cnorm) + cn.value * (qnorm[:, None] ** 2 + cnorm[None, :] ** 2)
    worst = float((np.abs(approx - true) / (eps + 1e-300)).max())
    assert worst < 1.0, "coarse words leave the certified band: max |approx - true| / eps_c = %.3f" % worst
    oi, od, _ = oracle_on_exported(ix, q, 8, 5, metric)
    same(ids, dis, oi, od)


# ---------------------------------------------------------------------------------------- filters (PREWHERE -> bitmap, strategy)

def test_filter_producers_match_numpy():
    """msvs_filter_*: bitmap from row offsets (getFilterFromPipeline's loop), from `column OP constant` on the device for
    every column type, combined with AND / OR / AND NOT; sizes that are not multiples of 64; NaN compares false."""
    rng = np.random.default_rng(11)
    for n in (1, 63, 64, 1000, 100_003):
        off = np.unique(rng.integers(0, n, max(1, n // 3))).astype(np.uint64)
        f = capi.Filter.from_offsets(off, n)
        exp = np.zeros(n, bool)
        exp[off.astype(np.int64)] = True
        assert f.to_bool().tolist() == exp.tolist() and f.count() == (int(exp.sum()), n)
        for dt in (np.uint8, np.uint16, np.uint32, np.uint64, np.int8, np.int16, np.int32, np.int64, np.float32, np.float64):
            if np.dtype(dt).kind == "f":
                col = rng.standard_normal(n).astype(dt) * 50
                col[rng.integers(0, n, max(1, n // 50))] = np.nan
            else:
                info = np.iinfo(dt)
                col = rng.integers(max(info.min, -100), min(info.max, 100), n).astype(dt)
            lo, hi = (-3.5, 17.25) if np.dtype(dt).kind == "f" else (3, 40)
            for op, fn in (("==", lambda c: c == lo), ("!=", lambda c: c != lo), ("<", lambda c: c < lo), ("<=", lambda c: c <= lo),
                           (">", lambda c: c > lo), (">=", lambda c: c >= lo), ("between", lambda c: (c >= lo) & (c <= hi))):
                with np.errstate(invalid="ignore"):
                    want = fn(col.astype(np.float64) if np.dtype(dt).kind == "f" else col.astype(np.int64))
                g = capi.Filter.from_predicate(col, op, lo, hi)
                assert g.to_bool().tolist() == want.tolist(), (n, dt, op)
                assert g.count()[0] == int(want.sum())
        a = rng.random(n) < 0.5
        b = rng.random(n) < 0.3
        for mode, fn in ((capi.FILTER_AND, lambda: a & b), (capi.FILTER_OR, lambda: a | b), (capi.FILTER_AND_NOT, lambda: a & ~b)):
            fa = capi.Filter.from_bool(a).combine(capi.Filter.from_bool(b), mode)
            assert fa.to_bool().tolist() == fn().tolist() and fa.count()[0] == int(fn().sum())


@pytest.mark.parametrize("kind,metric", [("ivf", capi.METRIC_L2), ("ivf", capi.METRIC_COSINE), ("flat", capi.METRIC_IP)])
def test_filtered_search_strategies_agree_with_each_other_and_the_oracle(kind, metric, opt):
    """A filtered search through the compacted view (filter_compact_below = 1: always), through the bit test (0: never)
    and with the default crossover: identical results, equal to the oracle's filtered scan; with a delete bitmap on top;
    selectivities from 0.1 % to 50 %, batches that take the one-query, the tiled and the candidate-pass paths."""
    rng = np.random.default_rng(300 + metric)
    n, d, nlist = 60000, 96, 64
    centers = rng.standard_normal((nlist, d), dtype=np.float32) * 2
    x = (centers[rng.integers(0, nlist, n)] + rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    q = (centers[rng.integers(0, nlist, 300)] + rng.standard_normal((300, d), dtype=np.float32)).astype(np.float32)
    if kind == "ivf":
        ix = build_ivf(x, metric, nlist)
        params, nprobe = "nprobe=8", 8
    else:
        ix = capi.Index(capi.INDEX_FLAT, metric, d)
        ix.add(x)
        ix.build()
        params, nprobe = "", 0
    price = rng.integers(0, 1000, n).astype(np.int32)  # the PREWHERE column
    deleted_alive = rng.random(n) < 0.9
    for sel in (1, 50, 500):  # price < sel: 0.1 %, 5 %, 50 % of the rows
        flt = capi.Filter.from_predicate(price, "<", sel)
        alive = price < sel
        assert flt.count()[0] == int(alive.sum())
        for nq in (1, 16, 300):
            for with_delete in (False, True):
                ix.set_delete_bitmap(deleted_alive if with_delete else None)
                eff = alive & deleted_alive if with_delete else alive
                res = []
                for below in ("1", "0", None):
                    opt("filter_compact_below", below)
                    res.append(ix.search_filter(q[:nq], 10, params, flt))
                    res.append(ix.search(q[:nq], 10, params, alive=alive))
                for r in res[1:]:
                    same(r[0], r[1], res[0][0], res[0][1])
                if nq <= 16:
                    if kind == "ivf":
                        oi, od, _ = oracle_on_exported(ix, q[:nq], nprobe, 10, metric, alive=eff)
                    else:
                        oi, od = o.knn(q[:nq], x, 10, OM[metric], alive=eff)
                    same(res[0][0], res[0][1], oi, od)
        flt.close()
    ix.set_delete_bitmap(None)


@pytest.mark.parametrize("kind", ["ivf", "flat"])
def test_compacted_view_with_repeated_labels_does_not_overrun(kind, opt):
    """msvs_index_add does not forbid repeated labels: three rows with label 5 pass a filter with ONE bit set, so the filter's
    population count is no bound on the rows of the compacted view (its row map was sized by it and written without a bound).
    Both strategies must return the same rows as the oracle, and nothing past the row map may be written."""
    rng = np.random.default_rng(55)
    n, d, nlist = 6000, 32, 16
    x = rng.standard_normal((n, d), dtype=np.float32)
    labels = (np.arange(n) // 3).astype(np.int64)  # every label three times
    q = rng.standard_normal((20, d), dtype=np.float32)
    if kind == "ivf":
        ix = build_ivf(x, capi.METRIC_L2, nlist, ids=labels)
        params = "nprobe=%d" % nlist
    else:
        ix = capi.Index(capi.INDEX_FLAT, capi.METRIC_L2, d)
        ix.add(x, labels)
        ix.build()
        params = ""
    alive = np.zeros(n // 3, bool)
    alive[rng.integers(0, n // 3, 40)] = True  # 40 bits -> 120 passing rows
    flt = capi.Filter.from_bool(alive)
    # the oracle on the rows themselves: a row passes when its label's bit is set; results carry the labels
    row_alive = alive[labels]
    oi, od = o.knn(q, x, 10, o.METRIC_L2, alive=row_alive)
    oi = np.where(oi >= 0, labels[np.maximum(oi, 0)], -1)
    for below in ("1", "0"):
        opt("filter_compact_below", below)
        ids, dis = ix.search_filter(q, 10, params, flt)
        assert (dis.view(np.uint32) == od.view(np.uint32)).all()
        assert (ids == oi).all()
    flt.close()


# ---------------------------------------------------------------------------------------- seam B: BM25

def bm25_both(docs_texts, query, k, alive=None):
    idx = TextIndex(docs_texts, o.fieldnorm_id)
    terms = [t for t in tokenize(query) if t in idx.vocab]
    qt = [idx.vocab[t] for t in terms]
    df = [idx.doc_freq(t) for t in terms]
    ps = capi.Postings(idx.post_off, idx.doc_ids, idx.tfs, idx.fieldnorm_ids)
    got = ps.bm25_search(qt, df, idx.num_docs, idx.total_tokens, k, alive=alive)
    exp = o.bm25_search(idx.post_off, idx.doc_ids, idx.tfs, idx.fieldnorm_ids, qt, df, idx.num_docs,
                        idx.total_tokens, k, alive=alive)
    return got, exp


def test_bm25_goldens_on_gpu():
    c = G["00040_hybrid"]
    got, exp = bm25_both([d["texts"] for d in c["docs"]], c["text_query"], 5)
    assert got[0].tolist() == exp[0].tolist() == [13, 0]
    assert got[1].tolist() == f32_of(c["text_search"][1]).tolist()
    c = G["00040_text_array"]
    got, exp = bm25_both([d["texts"] for d in c["docs"]], c["text_query"], 5)
    assert [c["docs"][int(r)]["id"] for r in got[0]] == c["text_search"][0]
    assert got[1].tolist() == f32_of(c["text_search"][1]).tolist()


def test_bm25_synthetic_corpus_matches_oracle():
    rng = np.random.default_rng(21)
    vocab = ["w%d" % i for i in range(3000)]
    p = 1.0 / np.arange(1, len(vocab) + 1) ** 1.1
    p /= p.sum()
    docs = []
    for _ in range(40000):
        n = max(1, rng.poisson(30))
        docs.append([" ".join(vocab[j] for j in rng.choice(len(vocab), n, p=p))])
    alive = rng.random(len(docs)) < 0.7
    for query in ("w3 w17 w250", "w1", "w40 w41 w42 w43", "w2999 w5"):
        for k in (10, 100):
            got, exp = bm25_both(docs, query, k)
            assert got[0].tolist() == exp[0].tolist()
            assert (got[1].view(np.uint32) == exp[1].view(np.uint32)).all()
        got, exp = bm25_both(docs, query, 10, alive=alive)
        assert got[0].tolist() == exp[0].tolist()
        assert (got[1].view(np.uint32) == exp[1].view(np.uint32)).all()


def bm25_scorer(opt, which):
    """"r": the posting scorer over score-ready records forced (bm25r_kernel; frequent terms take its split path), "p": the
    same over doc ids / tfs / gathered fieldnorms (bm25p_kernel, round 3), "w": the wave-private dense accumulator, "0": the
    block scorer it replaced.  (The default routes a batch to "r" or "w" by its posting density.)"""
    opt("bm25_wave", "0" if which == "0" else "1")
    opt("bm25_posting", "2" if which in ("p", "r") else "0")
    opt("bm25_rec", "1" if which == "r" else "0")


@pytest.mark.parametrize("wave", ["r", "p", "w", "0"])
@pytest.mark.parametrize("mode", ["emit", "lists", "forced_fallback"])
def test_bm25_many_doc_blocks(mode, wave, opt):
    """>= 64 document blocks (here 700k documents = 86 blocks): sample -> cut -> emit -> select; the same with the
    candidate list too small for any query (every query takes the exact fallback); and per-block lists only."""
    bm25_scorer(opt, wave)
    if mode == "lists":
        opt("bm25_emit", "0")
    if mode == "forced_fallback":
        opt("bm25_cand_cap", "3")
    rng = np.random.default_rng(33)
    n_docs, vocab = 700_000, 1500
    p = 1.0 / np.arange(1, vocab + 1) ** 1.1
    lens = np.maximum(1, rng.poisson(12, n_docs))
    toks = rng.choice(vocab, int(lens.sum()), p=p / p.sum())
    doc_of = np.repeat(np.arange(n_docs, dtype=np.int64), lens)
    uk, tf = np.unique(toks.astype(np.int64) * n_docs + doc_of, return_counts=True)
    term, doc = uk // n_docs, (uk % n_docs).astype(np.uint32)
    post_off = np.zeros(vocab + 1, np.int64)
    np.cumsum(np.bincount(term, minlength=vocab), out=post_off[1:])
    fn_of_len = np.array([o.fieldnorm_id(int(n)) for n in range(int(lens.max()) + 1)], np.uint8)
    fn_ids = fn_of_len[lens]
    ps = capi.Postings(post_off, doc, tf.astype(np.uint32), fn_ids)
    df_all = np.diff(post_off)
    alive = rng.random(n_docs) < 0.5
    q0, f0 = capi.bm25_stats()
    n_q = 0
    for qt in ([3, 40, 700], [0], [1400, 1499, 5, 90], [1499]):
        df = [int(df_all[t]) for t in qt]
        for k, al in ((10, None), (100, None), (10, alive), (256, None)):
            got = ps.bm25_search(qt, df, n_docs, int(lens.sum()), k, alive=al)
            exp = o.bm25_search(post_off, doc, tf.astype(np.uint32), fn_ids, qt, df, n_docs, int(lens.sum()), k, alive=al)
            assert got[0].tolist() == exp[0].tolist()
            assert (got[1].view(np.uint32) == exp[1].view(np.uint32)).all()
            n_q += 1
    q1, f1 = capi.bm25_stats()
    if mode == "emit":
        assert q1 - q0 == n_q and f1 - f0 <= 1
    if mode == "forced_fallback":
        assert q1 - q0 == n_q and f1 - f0 == n_q
    if mode == "lists":
        assert q1 == q0


def synthetic_postings(rng, n_docs, vocab, mean_len, num_fields=1):
    """Zipf corpus as flat arrays.  Several fields: term id = field * vocab + token."""
    p = 1.0 / np.arange(1, vocab + 1) ** 1.1
    p /= p.sum()
    terms, docs, tfs, fns, tokens = [], [], [], [], []
    for f in range(num_fields):
        lens = np.maximum(1, rng.poisson(mean_len * (1 + f), n_docs))
        toks = rng.choice(vocab, int(lens.sum()), p=p)
        doc_of = np.repeat(np.arange(n_docs, dtype=np.int64), lens)
        uk, tf = np.unique((toks.astype(np.int64) + f * vocab) * n_docs + doc_of, return_counts=True)
        terms.append(uk // n_docs)
        docs.append((uk % n_docs).astype(np.uint32))
        tfs.append(tf.astype(np.uint32))
        fn_of_len = np.array([o.fieldnorm_id(int(n)) for n in range(int(lens.max()) + 1)], np.uint8)
        fns.append(fn_of_len[lens])
        tokens.append(int(lens.sum()))
    term = np.concatenate(terms)
    post_off = np.zeros(vocab * num_fields + 1, np.int64)
    np.cumsum(np.bincount(term, minlength=vocab * num_fields), out=post_off[1:])
    term_field = np.repeat(np.arange(num_fields, dtype=np.uint8), vocab)
    return post_off, np.concatenate(docs), np.concatenate(tfs), np.stack(fns), term_field, np.asarray(tokens, np.uint64)


@pytest.mark.parametrize("wave", ["r", "p", "w", "0"])
def test_bm25_batch_equals_single_queries_and_oracle(wave, opt):
    """msvs_bm25_search_batch: every query of a batch == the one-query entry point == the oracle, bit for bit; the
    resident alive bitmap (msvs_postings_set_alive) ANDs with the per-call one."""
    bm25_scorer(opt, wave)
    rng = np.random.default_rng(91)
    n_docs, vocab = 300_000, 2000
    post_off, doc, tf, fn, _, tokens = synthetic_postings(rng, n_docs, vocab, 14)
    ps = capi.Postings(post_off, doc, tf, fn[0])
    df_all = np.diff(post_off)
    queries = [list(rng.choice(vocab, rng.integers(1, 6), replace=False)) for _ in range(67)]
    queries += [[0, 1, 2, 3, 4, 5, 6, 7], [vocab - 1], []]
    dfs = [[int(df_all[t]) for t in q] for q in queries]
    total_tokens = int(tokens[0])
    alive = rng.random(n_docs) < 0.6
    part_alive = rng.random(n_docs) < 0.9
    for k in (10, 100):
        for al in (None, alive):
            got = ps.bm25_search_batch(queries, dfs, n_docs, total_tokens, k, alive=al)
            for q, dfq, (gr, gs) in zip(queries, dfs, got):
                er, es = o.bm25_search(post_off, doc, tf, fn[0], q, dfq, n_docs, total_tokens, k, alive=al)
                assert gr.tolist() == er.tolist()
                assert (gs.view(np.uint32) == es.view(np.uint32)).all()
            for qi in (0, 5, 67):
                sr, ss = ps.bm25_search(queries[qi], dfs[qi], n_docs, total_tokens, k, alive=al)
                assert sr.tolist() == got[qi][0].tolist() and (ss.view(np.uint32) == got[qi][1].view(np.uint32)).all()
    ps.set_alive(part_alive)
    for al, eff in ((None, part_alive), (alive, alive & part_alive)):
        got = ps.bm25_search_batch(queries[:16], dfs[:16], n_docs, total_tokens, 10, alive=al)
        for q, dfq, (gr, gs) in zip(queries, dfs, got):
            er, es = o.bm25_search(post_off, doc, tf, fn[0], q, dfq, n_docs, total_tokens, 10, alive=eff)
            assert gr.tolist() == er.tolist() and (gs.view(np.uint32) == es.view(np.uint32)).all()
    ps.set_alive(None)
    got = ps.bm25_search_batch(queries[:4], dfs[:4], n_docs, total_tokens, 10)
    er, _ = o.bm25_search(post_off, doc, tf, fn[0], queries[0], dfs[0], n_docs, total_tokens, 10)
    assert got[0][0].tolist() == er.tolist()
    # stream-ordered device form
    import torch

    oi = torch.empty((len(queries), 10), device="cuda", dtype=torch.int64)
    od = torch.empty((len(queries), 10), device="cuda", dtype=torch.float32)
    ps.bm25_search_batch_device(queries, dfs, n_docs, total_tokens, 10, oi.data_ptr(), od.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    host = ps.bm25_search_batch(queries, dfs, n_docs, total_tokens, 10)
    for qi, (gr, gs) in enumerate(host):
        ids = oi[qi].cpu().numpy()
        assert ids[:len(gr)].tolist() == gr.astype(np.int64).tolist() and (ids[len(gr):] == -1).all()
        assert (od[qi].cpu().numpy()[:len(gr)].view(np.uint32) == gs.view(np.uint32)).all()


@pytest.mark.parametrize("wave", ["r", "p", "w", "0"])
@pytest.mark.parametrize("num_fields", [1, 3])
def test_bm25_and_operator_and_text_columns(num_fields, wave, opt):
    """operator_or = false (every token must match, in any column) and an index over several text columns: one term
    per (column, token), one token group per query token -- against the oracle's restatement."""
    bm25_scorer(opt, wave)
    rng = np.random.default_rng(92 + num_fields)
    n_docs, vocab = 120_000, 400
    post_off, doc, tf, fn, term_field, tokens = synthetic_postings(rng, n_docs, vocab, 6, num_fields)
    ps = capi.Postings(post_off, doc, tf, fn, term_field=term_field)
    df_all = np.diff(post_off)
    alive = rng.random(n_docs) < 0.8
    queries, groups = [], []
    for _ in range(40):
        toks = rng.choice(60, rng.integers(1, 5), replace=False)
        queries.append([int(f * vocab + t) for t in toks for f in range(num_fields)])
        groups.append([g for g in range(len(toks)) for _ in range(num_fields)])
    dfs = [[int(df_all[t]) for t in q] for q in queries]
    for op_or in (False, True):
        for al in (None, alive):
            got = ps.bm25_search_batch(queries, dfs, n_docs, tokens, 20, alive=al, groups=groups, operator_or=op_or)
            n_hits = 0
            for q, g, dfq, (gr, gs) in zip(queries, groups, dfs, got):
                er, es = o.bm25_search_ex(post_off, doc, tf, fn, q, dfq, n_docs, tokens, 20, alive=al, term_field=term_field,
                                          qgroups=g, operator_or=op_or)
                assert gr.tolist() == er.tolist()
                assert (gs.view(np.uint32) == es.view(np.uint32)).all()
                n_hits += len(gr)
            assert n_hits > 0


@pytest.mark.parametrize("rec", ["1", "0", "16384"])
@pytest.mark.parametrize("sub_docs", ["0", "8192", "512"])
def test_bm25_posting_scorer_windows_duplicates_and_density_routing(sub_docs, rec, opt):
    """The default routing: a batch of sparse terms goes to the posting-as-unit scorer (bm25p_kernel), whatever sub-range
    size it cuts its windows from; terms repeated inside a query, terms sharing most of their documents, 40-term queries
    and empty posting lists -- every hit and score bit == the oracle's dense term-order accumulation."""
    opt("bm25_sub_docs", sub_docs)
    opt("bm25_rec", "0" if rec == "0" else "1")  # 0: bm25p_kernel; else bm25r_kernel with 8192 / 16384 hash slots
    if rec == "16384":
        opt("bm25_slots", rec)
    rng = np.random.default_rng(123)
    n_docs, vocab = 600_000, 4000
    # terms 0..39: ~0.25 % of the documents each (the 40-term query stays under 1/8 posting per document), term 1 = half of
    # term 0's documents plus a few (shared documents);
    # the rest: rare.  Term vocab - 1 has no postings at all.
    lists = []
    base = np.sort(rng.choice(n_docs, 1800, replace=False)).astype(np.uint32)
    lists.append(base)
    lists.append(np.unique(np.concatenate([base[::2], rng.choice(n_docs, 600, replace=False).astype(np.uint32)])))
    for t in range(2, 40):
        lists.append(np.sort(rng.choice(n_docs, int(rng.integers(1000, 2000)), replace=False)).astype(np.uint32))
    for t in range(40, vocab - 1):
        lists.append(np.sort(rng.choice(n_docs, int(rng.integers(1, 400)), replace=False)).astype(np.uint32))
    lists.append(np.zeros(0, np.uint32))
    post_off = np.zeros(vocab + 1, np.int64)
    np.cumsum([len(x) for x in lists], out=post_off[1:])
    doc = np.concatenate(lists)
    tf = rng.integers(1, 6, len(doc)).astype(np.uint32)
    lens = np.maximum(1, rng.poisson(20, n_docs))
    fn = np.array([o.fieldnorm_id(int(n)) for n in range(int(lens.max()) + 1)], np.uint8)[lens]
    total_tokens = int(lens.sum())
    ps = capi.Postings(post_off, doc, tf, fn)
    df_all = np.diff(post_off)
    queries = [[0, 1], [1, 0, 1], [5, 5, 5], [3, 900, 17], [vocab - 1, 7], [vocab - 1], list(range(40)), [2000, 2001, 2002, 2003],
               [39, 38, 0, 1500]]
    queries += [list(rng.choice(vocab - 1, rng.integers(1, 5), replace=False)) for _ in range(55)]
    dfs = [[int(df_all[t]) for t in q] for q in queries]
    alive = rng.random(n_docs) < 0.5
    q0, f0 = capi.bm25_stats()
    for k in (10, 100):
        for al in (None, alive):
            got = ps.bm25_search_batch(queries, dfs, n_docs, total_tokens, k, alive=al)
            for q, dfq, (gr, gs) in zip(queries, dfs, got):
                er, es = o.bm25_search(post_off, doc, tf, fn, q, dfq, n_docs, total_tokens, k, alive=al)
                assert gr.tolist() == er.tolist(), q
                assert (gs.view(np.uint32) == es.view(np.uint32)).all(), q
    q1, f1 = capi.bm25_stats()
    assert q1 - q0 == 4 * len(queries) and f1 - f0 <= 8  # the sample/emit path, a rare exact fallback


@pytest.mark.parametrize("sub_docs", ["0", "512", "8192"])
@pytest.mark.parametrize("operator_or", [True, False])
def test_bm25_four_term_scorer_paths(sub_docs, operator_or, opt):
    """bm25l_kernel (every query of the batch has <= 4 terms): the look-ahead with a stride (sparse queries beside a dense one),
    the fall-back to single sub-ranges inside a burst, the split of a sub-range by document id (a burst of consecutive documents,
    frequent terms under bm25_posting = 2), terms repeated inside a query (every record shares its document: the flagged path with
    hundreds of records), terms sharing half their documents, an empty posting list, one-term queries, filters, AND -- every hit and
    score bit == the oracle's dense term-order accumulation; and == the general record scorer (bm25_lean = 0)."""
    opt("bm25_sub_docs", sub_docs)
    opt("bm25_posting", "2")  # the posting scorer whatever the density: dense sub-ranges are split
    rng = np.random.default_rng(321)
    n_docs, vocab = 600_000, 3000  # (>= 500 000 documents: the sample / cut / emit flow)
    lists = []
    base = np.sort(rng.choice(n_docs, 2500, replace=False)).astype(np.uint32)
    lists.append(base)                                                                                    # 0
    lists.append(np.unique(np.concatenate([base[::2], rng.choice(n_docs, 900, replace=False).astype(np.uint32)])))  # 1: shares half of 0
    burst = np.arange(123_000, 127_000, dtype=np.uint32)  # 4000 consecutive documents
    lists.append(np.unique(np.concatenate([burst, rng.choice(n_docs, 1500, replace=False).astype(np.uint32)])))     # 2: sparse + a burst
    lists.append(np.unique(np.concatenate([burst[::3], rng.choice(n_docs, 800, replace=False).astype(np.uint32)]))) # 3: a third of the burst
    lists.append(np.sort(rng.choice(n_docs, 180_000, replace=False)).astype(np.uint32))                   # 4: 30 % of the documents
    lists.append(np.sort(rng.choice(n_docs, 90_000, replace=False)).astype(np.uint32))                    # 5: 15 %
    for t in range(6, vocab - 1):
        lists.append(np.sort(rng.choice(n_docs, int(rng.integers(1, 600)), replace=False)).astype(np.uint32))
    lists.append(np.zeros(0, np.uint32))                                                                  # vocab - 1: empty
    post_off = np.zeros(vocab + 1, np.int64)
    np.cumsum([len(x) for x in lists], out=post_off[1:])
    doc = np.concatenate(lists)
    tf = rng.integers(1, 6, len(doc)).astype(np.uint32)
    lens = np.maximum(1, rng.poisson(20, n_docs))
    fn = np.array([o.fieldnorm_id(int(n)) for n in range(int(lens.max()) + 1)], np.uint8)[lens]
    total_tokens = int(lens.sum())
    ps = capi.Postings(post_off, doc, tf, fn)
    df_all = np.diff(post_off)
    queries = [[0, 1], [1, 0, 1], [7, 7, 7, 7], [2, 3], [3, 2, 900, 17], [4, 5], [5, 4, 2, 0], [4], [vocab - 1, 7], [vocab - 1], [2],
               [2000, 2001, 2002, 2003], [4, 4]]
    queries += [list(rng.choice(vocab - 1, rng.integers(1, 5), replace=False)) for _ in range(51)]
    assert max(len(q) for q in queries) <= 4
    dfs = [[int(df_all[t]) for t in q] for q in queries]
    alive = rng.random(n_docs) < 0.5
    for k, emit in ((10, "1"), (100, "1"), (10, "0")):  # (emit 0: per-chunk top-k lists -- the scorer's other output mode)
        opt("bm25_emit", emit)
        for al in (None, alive):
            ref = None
            for lean in ("1", "0"):
                opt("bm25_lean", lean)
                got = ps.bm25_search_batch(queries, dfs, n_docs, total_tokens, k, alive=al, operator_or=operator_or)
                if ref is None:
                    ref = got
                    for q, dfq, (gr, gs) in zip(queries, dfs, got):
                        er, es = o.bm25_search_ex(post_off, doc, tf, fn, q, dfq, n_docs, total_tokens, k, alive=al, operator_or=operator_or)
                        assert gr.tolist() == er.tolist(), q
                        assert (gs.view(np.uint32) == es.view(np.uint32)).all(), q
                else:
                    for (gr, gs), (rr, rs) in zip(got, ref):
                        assert gr.tolist() == rr.tolist() and (gs.view(np.uint32) == rs.view(np.uint32)).all()
    opt("bm25_lean", None)
    opt("bm25_emit", None)


def test_bm25_records_follow_the_statistics_of_the_call(opt):
    """bm25r_kernel reads (doc, tf / (tf + cache[fieldnorm])) records derived for ONE fieldnorm cache, i.e. one average field
    length: searches that alternate between two corpus statistics (a part alone / the sum over parts, BM25InfoInDataParts.cpp)
    must each see records for their own -- against the oracle, bit for bit -- and a w = 0 term (df == N on a large corpus:
    ln(1 + 0.5 / (N + 0.5)) rounds to 0 in f32) scores 0 without breaking ownership of shared documents."""
    opt("bm25_posting", "2")
    rng = np.random.default_rng(321)
    n_docs, vocab = 400_000, 800
    post_off, doc, tf, fn, _, tokens = synthetic_postings(rng, n_docs, vocab, 10)
    ps = capi.Postings(post_off, doc, tf, fn[0])
    df_all = np.diff(post_off)
    queries = [list(rng.choice(vocab, 3, replace=False)) for _ in range(24)] + [[0, 1], [2, 2], [700]]
    dfs = [[int(df_all[t]) for t in q] for q in queries]
    stats = [(n_docs, int(tokens[0])), (3 * n_docs, 5 * int(tokens[0])), (n_docs, int(tokens[0]))]
    for total_docs, total_tokens in stats:
        got = ps.bm25_search_batch(queries, dfs, total_docs, total_tokens, 20)
        for q, dfq, (gr, gs) in zip(queries, dfs, got):
            er, es = o.bm25_search(post_off, doc, tf, fn[0], q, dfq, total_docs, total_tokens, 20)
            assert gr.tolist() == er.tolist(), q
            assert (gs.view(np.uint32) == es.view(np.uint32)).all(), q
    # a term present in EVERY document of a 9M-document corpus statistics: idf = ln(1 + 0.5 / (N + 0.5)) = 0 in f32
    big = 9_000_000
    q = [[5, 9], [9, 5, 11]]
    dfq = [[big, int(df_all[9])], [int(df_all[9]), big, int(df_all[11])]]
    got = ps.bm25_search_batch(q, dfq, big, 10 * big, 15)
    for qq, dd, (gr, gs) in zip(q, dfq, got):
        er, es = o.bm25_search(post_off, doc, tf, fn[0], qq, dd, big, 10 * big, 15)
        assert gr.tolist() == er.tolist() and (gs.view(np.uint32) == es.view(np.uint32)).all()


@pytest.mark.parametrize("fusion", ["rrf", "rsf"])
def test_device_fusion_of_a_hybrid_batch_equals_the_host_fusion(fusion):
    """msvs_hybrid_fuse_device == msvs_host_hybrid_search_batch (the flat-array form of hybridSearch + RankFusion /
    RelativeScoreFusion, itself held against the map-based mirror in test_host_mirror.py): labels, order and score BITS, for
    lists that overlap a little, a lot, not at all, short lists (ids -1), equal scores (RSF: all-equal lists normalise to 1)."""
    import torch

    rng = np.random.default_rng(77 if fusion == "rrf" else 78)
    nq, kv, kt, topk = 37, 100, 100, 10
    vi = np.full((nq, kv), -1, np.int64)
    ti = np.full((nq, kt), -1, np.int64)
    vd = np.zeros((nq, kv), np.float32)
    td = np.zeros((nq, kt), np.float32)
    for q in range(nq):
        nv = int(rng.integers(0, kv + 1)) if q % 5 else kv
        nt = int(rng.integers(0, kt + 1)) if q % 7 else kt
        pool = rng.choice(5000, 260, replace=False)
        share = [0.0, 0.1, 0.9, 1.0][q % 4]
        v = pool[:nv]
        t_take = np.where(rng.random(nt) < share, 1, 0)
        t = np.array([pool[j] if (t_take[j] and j < nv) else pool[130 + j] for j in range(nt)], np.int64)
        vi[q, :nv], ti[q, :nt] = v, t
        vd[q, :nv] = np.sort(rng.random(nv).astype(np.float32))            # distances ascending
        td[q, :nt] = -np.sort(-rng.random(nt).astype(np.float32) * 20)      # scores descending
        if q % 6 == 1 and nt:
            td[q, :nt] = np.float32(3.25)                                   # all equal
        if q % 6 == 2 and nv > 3:
            vd[q, 1:4] = vd[q, 1]                                           # ties inside a list
    for direction, weight, fk in ((1, 0.5, 60), (-1, 0.3, 7)):
        exp = mhost.hybrid_search_batch(fusion, vd, vi, td, ti, topk, fusion_k=fk, fusion_weight=weight, vector_scan_direction=direction)
        g = lambda a: torch.from_numpy(a).cuda()
        dvd, dvi, dtd, dti = g(vd), g(vi), g(td), g(ti)
        os_ = torch.empty((nq, topk), device="cuda", dtype=torch.float32)
        ol = torch.empty((nq, topk), device="cuda", dtype=torch.int64)
        on = torch.empty((nq,), device="cuda", dtype=torch.int32)
        capi.hybrid_fuse_device(fusion, dvd.data_ptr(), dvi.data_ptr(), kv, dtd.data_ptr(), dti.data_ptr(), kt, nq, topk,
                                os_.data_ptr(), ol.data_ptr(), on.data_ptr(), torch.cuda.current_stream().cuda_stream,
                                fusion_k=fk, fusion_weight=weight, vector_scan_direction=direction)
        torch.cuda.synchronize()
        hs, hl, hn = os_.cpu().numpy(), ol.cpu().numpy(), on.cpu().numpy()
        es, el, ec = exp
        for q in range(nq):
            n = int(ec[q])
            assert int(hn[q]) == n, q
            assert hl[q, :n].tolist() == [int(x) for x in el[q, :n]], q
            assert (hs[q, :n].view(np.uint32) == es[q, :n].view(np.uint32)).all(), q
            assert (hl[q, n:] == -1).all()


def build_store(docs, columns=("doc",)):
    """docs: list of {column: text | [texts]}; row id = position."""
    st = mhost.TextIndexStore(list(columns))
    for r, d in enumerate(docs):
        names, texts = [], []
        for c in columns:
            for t in (d[c] if isinstance(d[c], list) else [d[c]]):
                names.append(c)
                texts.append(t)
        st.add_doc(r, names, texts)
    st.commit()
    return st


def test_distributed_bm25_statistics_two_shards_of_two_parts():
    """The DFS statistics path of a Distributed-table text search (StorageFtsIndex.cpp:150-213, CommonUtils.cpp:190-330):
    every shard sums its parts (ftsIndex row), the initiator sums the shards' rows, the shards score with the merged
    statistics -- and the union of their hits is what ONE index over all documents returns, score bits included."""
    rng = np.random.default_rng(91)
    vocab = ["w%02d" % i for i in range(60)]
    docs = [" ".join(rng.choice(vocab, size=int(rng.integers(3, 30)))) for _ in range(400)]
    cuts = [0, 90, 200, 330, 400]  # shard 0 = parts 0, 1; shard 1 = parts 2, 3
    parts = [build_store([{"doc": d} for d in docs[cuts[i]:cuts[i + 1]]]) for i in range(4)]
    whole = build_store([{"doc": d} for d in docs])
    query = "w03 w17 w41 nosuchword w03"
    rows = [mhost.fts_index_statistics(parts[:2], query), mhost.fts_index_statistics(parts[2:], query)]
    merged = mhost.fts_statistics_merge(rows)
    ref = whole.statistics(query)
    assert merged.total_num_docs == ref.total_num_docs == 400
    assert merged.total_num_tokens == ref.total_num_tokens
    assert sorted(merged.docs_freq) == sorted(ref.docs_freq)
    assert merged.docs_freq == sorted(merged.docs_freq, key=lambda t: (t[1], t[0]))  # (field_id, term) order
    assert ("nosuchword", 0, 0) in merged.docs_freq
    # python-side sum of per-part statistics (BM25InfoInDataParts) agrees
    py = mhost.Statistics.sum([p.statistics(query) for p in parts])
    assert sorted(py.docs_freq) == sorted(merged.docs_freq) and dict(py.total_num_tokens) == dict(merged.total_num_tokens)
    hits = []
    for i, p in enumerate(parts):
        r, sc = p.bm25_search(query, 10, statistics=merged)
        hits += [(-float(x), int(rr) + cuts[i], x) for rr, x in zip(r, sc)]
    hits.sort(key=lambda h: (h[0], h[1]))
    r, sc = whole.bm25_search(query, 10, statistics=ref)
    assert [h[1] for h in hits[:10]] == [int(x) for x in r]
    assert np.array([h[2] for h in hits[:10]], np.float32).view(np.uint32).tolist() == sc.view(np.uint32).tolist()
    # a part without a committed index: the reference's "Fts index file does not exist"
    st = mhost.TextIndexStore(["doc"])
    with pytest.raises(capi.MsvsError) as e:
        mhost.fts_index_statistics([parts[0], st], query)
    assert e.value.code == capi.ERR_NOT_IMPLEMENTED


def test_text_store_replays_the_bm25_goldens(tmp_path):
    """Seam B end to end through the native host side (text_store.cpp: exporter -> export file -> loader -> device):
    the sentence goes in as a string like TantivyIndexStore::bm25Search's, the goldens of 00040 / 00041 come out."""
    c = G["00040_hybrid"]
    docs = c["docs"]
    st = build_store([{"doc": d["texts"]} for d in docs])
    path = str(tmp_path / "part0.mspost")
    st.save(path)
    st2 = mhost.TextIndexStore.load(path)
    for s_ in (st, st2):
        rows, scores = s_.bm25_search(c["text_query"], c["limit"], statistics=s_.statistics(c["text_query"]))
        assert [docs[int(r)]["id"] for r in rows] == c["text_search"][0]
        assert scores.tolist() == f32_of(c["text_search"][1]).tolist()
        alive = np.array([d["id"] < 10 for d in docs])
        rows, scores = s_.bm25_search(c["text_query"], c["limit"], alive=alive)
        assert [docs[int(r)]["id"] for r in rows] == c["text_search_where_id_lt_10"][0]
        assert scores.tolist() == f32_of(c["text_search_where_id_lt_10"][1]).tolist()
        # the same filter, resident (lightweight delete of the part)
        s_.set_alive(alive)
        rows2, scores2 = s_.bm25_search(c["text_query"], c["limit"])
        assert rows2.tolist() == rows.tolist() and scores2.tolist() == scores.tolist()
        s_.set_alive(None)
    # Array(String) column
    a = G["00040_text_array"]
    st = build_store([{"doc": d["texts"]} for d in a["docs"]])
    rows, scores = st.bm25_search(a["text_query"], a["limit"])
    assert [a["docs"][int(r)]["id"] for r in rows] == a["text_search"][0]
    assert scores.tolist() == f32_of(a["text_search"][1]).tolist()
    # 00041: two parts, table-level statistics summed over the parts (BM25InfoInDataParts)
    t = G["00041_two_parts"]
    docs = t["docs"]
    n0 = t["part_sizes"][0]
    parts = [build_store([{"doc": d["texts"]} for d in docs[:n0]]), build_store([{"doc": d["texts"]} for d in docs[n0:]])]
    stats = mhost.Statistics.sum([p.statistics(t["text_query"]) for p in parts])
    hits = []
    for pi, p in enumerate(parts):
        rows, scores = p.bm25_search(t["text_query"], t["limit"], statistics=stats)
        hits += [(float(s), docs[int(r) + pi * n0]["id"]) for r, s in zip(rows, scores)]
    hits.sort(key=lambda h: (-h[0], h[1]))
    assert [h[1] for h in hits[:2]] == t["text_search_2parts"][0]
    assert np.array([h[0] for h in hits[:2]], np.float32).tolist() == f32_of(t["text_search_2parts"][1]).tolist()
    # a corrupt export is refused
    raw = bytearray(open(path, "rb").read())
    for cut in (10, 64, len(raw) - 3):
        bad = str(tmp_path / "bad.mspost")
        open(bad, "wb").write(bytes(raw[:cut]))
        with pytest.raises(capi.MsvsError):
            mhost.TextIndexStore.load(bad)
    raw[0] ^= 0xFF
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(capi.MsvsError):
        mhost.TextIndexStore.load(bad)
    with pytest.raises(capi.MsvsError):
        st.bm25_search("x", 5, enable_nlq=True)


def test_text_store_two_columns_and_operator_against_oracle():
    """An fts index over two columns + operator_or = false: the store's term / group construction against the oracle
    fed with a python-built index of the same documents."""
    rng = np.random.default_rng(77)
    words = ["w%d" % i for i in range(120)]
    p = 1.0 / np.arange(1, len(words) + 1)
    p /= p.sum()
    docs = [{"a": " ".join(rng.choice(words, rng.integers(1, 12), p=p)), "b": " ".join(rng.choice(words, rng.integers(1, 30), p=p))}
            for _ in range(20000)]
    st = build_store(docs, columns=("a", "b"))
    ia = TextIndex([[d["a"]] for d in docs], o.fieldnorm_id)
    ib = TextIndex([[d["b"]] for d in docs], o.fieldnorm_id)
    # the python twin of the export: terms of column a, then of column b, each sorted by bytes
    order_a, order_b = sorted(ia.vocab), sorted(ib.vocab)
    term_id = {("a", t): i for i, t in enumerate(order_a)}
    term_id.update({("b", t): len(order_a) + i for i, t in enumerate(order_b)})

    def lists(ix, order):
        return [(ix.doc_ids[ix.post_off[ix.vocab[t]]:ix.post_off[ix.vocab[t] + 1]], ix.tfs[ix.post_off[ix.vocab[t]]:ix.post_off[ix.vocab[t] + 1]])
                for t in order]

    pl = lists(ia, order_a) + lists(ib, order_b)
    post_off = np.zeros(len(pl) + 1, np.int64)
    np.cumsum([len(x[0]) for x in pl], out=post_off[1:])
    doc_ids = np.concatenate([x[0] for x in pl]).astype(np.uint32)
    tfs = np.concatenate([x[1] for x in pl]).astype(np.uint32)
    fn = np.stack([ia.fieldnorm_ids, ib.fieldnorm_ids])
    term_field = np.array([0] * len(order_a) + [1] * len(order_b), np.uint8)
    tokens = [ia.total_tokens, ib.total_tokens]
    assert st.total_num_tokens() == [(0, ia.total_tokens), (1, ib.total_tokens)] and st.total_num_docs() == len(docs)
    alive = rng.random(len(docs)) < 0.7
    for sentence in ("w0 w3", "w1 W7 w40", "w5", "w2 w2 w9", "w100 w90 nosuchword"):
        for op_or in (True, False):
            for cols in (None, ["b"], ["a", "b"]):
                fields = ["a", "b"] if cols is None else cols
                q, g, df, grp, dead = [], [], [], 0, False
                for t in dict.fromkeys(tokenize(sentence)):
                    hit = False
                    for f in fields:
                        if (f, t) in term_id:
                            q.append(term_id[(f, t)])
                            g.append(grp)
                            df.append(int(post_off[q[-1] + 1] - post_off[q[-1]]))
                            hit = True
                    grp += hit
                    dead |= (not hit) and not op_or
                if dead:
                    q, g, df = [], [], []
                for al in (None, alive):
                    rows, scores = st.bm25_search(sentence, 15, column_names=cols, alive=al, operator_or=op_or)
                    er, es = o.bm25_search_ex(post_off, doc_ids, tfs, fn, q, df, len(docs), tokens, 15, alive=al,
                                              term_field=term_field, qgroups=g, operator_or=op_or)
                    assert rows.tolist() == er.tolist(), (sentence, op_or, cols)
                    assert (scores.view(np.uint32) == es.view(np.uint32)).all()
    assert st.doc_freq("w0 nosuchword")[0] == ("nosuchword", 0, 0)


# ---------------------------------------------------------------------------------------- BASELINE-size properties

def test_full_size_1m_x_768_properties(opt):
    """BASELINE config 2 shape (1M x 768, nlist 1024, nprobe 32, top-10): the oracle cannot redo all of it in
    seconds, so check size-independent properties + a handful of oracle-verified queries."""
    import torch

    n, d, nlist, nq, k, nprobe = 1_000_000, 768, 1024, 64, 10, 32
    g = torch.Generator(device="cuda").manual_seed(1234)
    x_dev = torch.randn((n, d), generator=g, device="cuda", dtype=torch.float32)
    q_dev = torch.randn((nq, d), generator=g, device="cuda", dtype=torch.float32)
    ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, "ncentroids=%d,kmeans_iters=4,train_sample=32768" % nlist)
    ix.train(x_dev.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    ix.add(x_dev.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    ix.build()
    assert ix.num_data == n
    q = q_dev.cpu().numpy()
    ids, dis = ix.search(q, k, "nprobe=%d" % nprobe)
    # (1) sortedness, uniqueness, valid ids
    assert (np.diff(dis, axis=1) >= 0).all()
    assert all(len(set(r)) == k for r in ids.tolist()) and ids.min() >= 0 and ids.max() < n
    # (2) idempotence / batch independence: one query alone == the same query inside the batch
    i1, d1 = ix.search(q[5:6], k, "nprobe=%d" % nprobe)
    same(i1, d1, ids[5:6], dis[5:6])
    # (3) the reported distance is the canonical distance of the reported row
    x_rows = x_dev[torch.from_numpy(ids[:4].reshape(-1)).cuda()].cpu().numpy().reshape(4, k, d)
    for qi in range(4):
        for j in range(k):
            assert o.l2sqr(q[qi], x_rows[qi, j]) == dis[qi, j]
    # (4) oracle on the exported structure for a few queries
    cent, off, vecs, lids = ix.export()
    oi, od, _ = o.ivf_search(cent, off, vecs, lids, q[:6], nprobe, k, o.METRIC_L2, threads=8)
    same(ids[:6], dis[:6], oi, od)
    # (5) probing all lists == exhaustive FLAT scan of the resident rows (ids and distances)
    flat = capi.Index(capi.INDEX_FLAT, capi.METRIC_L2, d)
    flat.add(x_dev.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    flat.build()
    fi, fd = flat.search(q[:8], k)
    ai, ad = ix.search(q[:8], k, "nprobe=256")
    oe, de = o.knn(q[:2], x_dev.cpu().numpy(), k, o.METRIC_L2, threads=8)
    same(fi[:2], fd[:2], oe, de)
    # nprobe=256 of 1024 lists on iid gaussians is not exhaustive; it can only be worse or equal, never better
    assert (ad >= fd - 0).all()
    # (6) a 2048-query batch goes through the matrix-core candidate pass (coarse quantiser included); a query's answer
    #     must not depend on the batch it travels in: compare with the same queries searched 16 at a time (canonical
    #     kernels) and with the oracle
    g2 = torch.Generator(device="cuda").manual_seed(99)
    big = torch.randn((2048, d), generator=g2, device="cuda", dtype=torch.float32).cpu().numpy()
    q0, f0 = capi.prefilter_stats()
    bi, bd = ix.search(big, k, "nprobe=%d" % nprobe)
    q1, f1 = capi.prefilter_stats()
    assert q1 - q0 == 2048 and f1 - f0 <= 64
    pick = np.arange(0, 2048, 64)
    opt("ivf_pass", "0")
    si, sd = ix.search(big[pick[:16]], k, "nprobe=%d" % nprobe)
    same(bi[pick[:16]], bd[pick[:16]], si, sd)
    si, sd = ix.search(big[pick[16:]], k, "nprobe=%d" % nprobe)
    same(bi[pick[16:]], bd[pick[16:]], si, sd)
    assert capi.prefilter_stats()[0] == q1  # those really were canonical runs
    opt("ivf_pass", None)
    # small batches take the shadow pass too (from a quarter of a pair per list on): same answer
    si, sd = ix.search(big[pick[:16]], k, "nprobe=%d" % nprobe)
    same(bi[pick[:16]], bd[pick[:16]], si, sd)
    assert capi.prefilter_stats()[0] == q1 + 16
    oi, od, _ = o.ivf_search(cent, off, vecs, lids, big[pick[:4]], nprobe, k, o.METRIC_L2, threads=8)
    same(bi[pick[:4]], bd[pick[:4]], oi, od)
    # (7) the exact FLAT scan of a batch (two-phase table pass) == the canonical exhaustive scan
    fi2, fd2 = flat.search(big[:128], k)
    fs, fds = flat.search(big[:8], k)
    same(fi2[:8], fd2[:8], fs, fds)


# ---------------------------------------------------------------------------------------- hybrid (config 5 shape)

def test_hybrid_goldens_end_to_end_on_gpu():
    """00040: vector top-k (FLAT index on the GPU) + BM25 (GPU) + RRF / RSF (host mirror) == the reference's output."""
    import myscaledb_amd.host as host
    c = G["00040_hybrid"]
    docs, limit = c["docs"], c["limit"]
    vecs = np.array([d["vector"] for d in docs], np.float32)
    ix = capi.Index(capi.INDEX_FLAT, capi.METRIC_L2, 3)
    ix.add(vecs)
    ix.build()
    vi, vd = ix.search(np.array([c["vec_query"]], np.float32), limit)
    (tr, ts), _ = bm25_both([d["texts"] for d in docs], c["text_query"], limit)
    z = lambda n: np.zeros(n, np.uint64)
    for kind in ("rsf", "rrf"):
        s, _, l = host.hybrid_search(kind, (vd[0], z(limit), vi[0]), (ts, z(len(tr)), tr), limit)
        order = sorted(range(len(l)), key=lambda j: (-float(s[j]), docs[int(l[j])]["id"]))
        assert [docs[int(l[j])]["id"] for j in order] == c[kind][0]
        assert [np.float32(s[j]) for j in order] == f32_of(c[kind][1]).tolist()


def test_hybrid_pipeline_synthetic_matches_oracle_pipeline():
    """vector top-100 (IVFFLAT, cosine) + BM25 top-100 + RRF(k=60) -> top-10, GPU pipeline vs oracle pipeline."""
    import myscaledb_amd.host as host
    rng = np.random.default_rng(31)
    n, d, nlist = 20000, 64, 32
    x = rng.standard_normal((n, d), dtype=np.float32)
    vocab = ["t%d" % i for i in range(800)]
    p = 1.0 / np.arange(1, 801) ** 1.1
    p /= p.sum()
    docs = [[" ".join(vocab[j] for j in rng.choice(800, max(1, rng.poisson(20)), p=p))] for _ in range(n)]
    ix = build_ivf(x, capi.METRIC_COSINE, nlist)
    idx = TextIndex(docs, o.fieldnorm_id)
    ps = capi.Postings(idx.post_off, idx.doc_ids, idx.tfs, idx.fieldnorm_ids)
    cent, off, vecs, lids = ix.export()
    for qi in range(5):
        q = rng.standard_normal((1, d), dtype=np.float32)
        terms = ["t%d" % t for t in rng.integers(5, 200, 3)]
        qt = [idx.vocab[t] for t in terms if t in idx.vocab]
        df = [idx.doc_freq(t) for t in terms if t in idx.vocab]
        gi, gd = ix.search(q, 100, "nprobe=8")
        gr, gs = ps.bm25_search(qt, df, idx.num_docs, idx.total_tokens, 100)
        oi, od, _ = o.ivf_search(cent, off, vecs, lids, o.normalize_rows(q), 8, 100, o.METRIC_IP)
        od = (np.float32(1) - od).astype(np.float32)
        orr, osc = o.bm25_search(idx.post_off, idx.doc_ids, idx.tfs, idx.fieldnorm_ids, qt, df, idx.num_docs,
                                 idx.total_tokens, 100)
        same(gi, gd, oi, od)
        assert gr.tolist() == orr.tolist() and (gs.view(np.uint32) == osc.view(np.uint32)).all()
        z = lambda m: np.zeros(m, np.uint64)
        a = host.hybrid_search("rrf", (gd[0], z(100), gi[0]), (gs, z(len(gr)), gr), 10, fusion_k=60)
        b = o.hybrid_fusion("rrf", (od[0], z(100), oi[0]), (osc, z(len(orr)), orr), 10, fusion_k=60)
        assert a[2].tolist() == b[2].tolist() and (a[0] == b[0]).all()  # integer ranks / row ids bit-exact


def test_strided_device_merge_of_packed_exchange_buffers():
    """The multi-GPU exchange layout: every rank's {ids[nq*k] i64 | dis[nq*k] f32} packed back to back (what ONE
    all-gather delivers), merged in place by msvs_merge_topk_device_strided == the host-pointer merge."""
    import ctypes as C

    import torch
    rng = np.random.default_rng(41)
    W, nq, k = 5, 37, 10
    ids = rng.permutation(W * nq * k).reshape(W, nq, k).astype(np.int64)
    dis = rng.integers(0, 50, (W, nq, k)).astype(np.float32)
    ids[3, :, 6:] = -1  # a shard with fewer than k local hits
    for metric in (capi.METRIC_L2, capi.METRIC_IP):
        order = np.argsort(dis if metric == capi.METRIC_L2 else -dis, axis=2, kind="stable")
        si, sd = np.take_along_axis(ids, order, 2), np.take_along_axis(dis, order, 2)
        part = (nq * k * 12 + 7) // 8 * 8
        buf = np.zeros(W * part, np.uint8)
        for p in range(W):
            buf[p * part:p * part + nq * k * 8] = si[p].reshape(-1).view(np.uint8)
            buf[p * part + nq * k * 8:p * part + nq * k * 12] = sd[p].reshape(-1).view(np.uint8)
        g = torch.from_numpy(buf).cuda()
        oi = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        od = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        capi._check(capi.lib().msvs_merge_topk_device_strided(
            C.c_void_p(g.data_ptr()), C.c_size_t(part // 8), C.c_void_p(g.data_ptr() + nq * k * 8),
            C.c_size_t(part // 4), C.c_size_t(W), C.c_size_t(nq), C.c_size_t(k), metric,
            C.c_void_p(oi.data_ptr()), C.c_void_p(od.data_ptr()), None))
        torch.cuda.synchronize()
        ri, rd = capi.merge_topk(si, sd, metric)
        same(oi.cpu().numpy(), od.cpu().numpy(), ri, rd)


@pytest.mark.parametrize("W,k", [(2, 10), (4, 40), (3, 100), (70, 10)])
def test_merge_topk_keeps_duplicate_keys_without_holes(W, k):
    """Overlapping parts / replicated shards hand the merge the SAME (distance, id) more than once: every copy gets a
    slot (the reference's multimap keeps all of them) and no -1 hole appears before the valid entries end."""
    rng = np.random.default_rng(W * 100 + k)
    nq = 9
    base_i = np.sort(rng.permutation(1000)[:k]).astype(np.int64)
    base_d = np.sort(rng.integers(0, 30, k)).astype(np.float32)
    ids = np.tile(base_i, (W, nq, 1))
    dis = np.tile(base_d, (W, nq, 1))
    ids[-1, :, k // 2:] = -1  # one part with fewer hits
    dis[-1, :, k // 2:] = np.finfo(np.float32).max
    oi, od = capi.merge_topk(ids, dis, capi.METRIC_L2)
    # expected: the k smallest of the concatenation by (distance, id), duplicates included
    for q in range(nq):
        pairs = sorted((float(d), int(i)) for p in range(W) for d, i in zip(dis[p, q], ids[p, q]) if i >= 0)[:k]
        assert [i for _, i in pairs] == oi[q].tolist()
        assert [d for d, _ in pairs] == od[q].tolist()


def test_empty_filter_bitmap_means_no_rows_and_large_dimension_tiles_fit_lds():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3000, 24), dtype=np.float32)
    for typ, params in ((capi.INDEX_FLAT, ""), (capi.INDEX_IVFFLAT, "ncentroids=8")):
        ix = capi.Index(typ, capi.METRIC_L2, 24, params)
        if typ == capi.INDEX_IVFFLAT:
            ix.train(x)
        ix.add(x)
        ix.build()
        ids, dis = ix.search(x[:5], 4, "nprobe=8" if typ == capi.INDEX_IVFFLAT else "", alive=np.zeros(0, bool))
        assert (ids == -1).all()  # a present filter with zero bits: nothing passes (it is NOT "no filter")
    # d = 3072, k = 256, 9 queries: the 8-query tile would need 180 KB of LDS; the planner shrinks it
    d, k = 3072, 256
    y = rng.standard_normal((1500, d), dtype=np.float32)
    q = rng.standard_normal((9, d), dtype=np.float32)
    for metric in (capi.METRIC_L2, capi.METRIC_IP):
        ids, dis = capi.knn(q, y, k, metric)
        oi, od = o.knn(q, y, k, OM[metric])
        same(ids, dis, oi, od)
    ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, "ncentroids=4,kmeans_iters=2")
    ix.train(y)
    ix.add(y)
    ix.build()
    ids, dis = ix.search(q, k, "nprobe=4")
    oi, od, _ = oracle_on_exported(ix, q, 4, k, capi.METRIC_L2)
    same(ids, dis, oi, od)


# ---------------------------------------------------------------------------------------- binary vectors (f4)

@pytest.mark.parametrize("case", ["00038_binary_hamming", "00038_binary_jaccard"])
def test_binary_vector_goldens_on_gpu(case):
    """00038 through msvs_knn_bin: single query, batch, WHERE filter, lightweight delete."""
    c = G[case]
    n = np.arange(c["rows"], dtype=np.int64)
    y = np.repeat((n % 256).astype(np.uint8)[:, None], c["nbytes"], axis=1)
    metric = capi.METRIC_HAMMING if c["metric"] == "Hamming" else capi.METRIC_JACCARD
    ids, dis = capi.knn_bin(np.array([c["query"]], np.uint8), y, c["k"], metric)
    assert ids[0].tolist() == c["ids"] and dis[0].tolist() == f32_of(c["dists"]).tolist()
    ids, dis = capi.knn_bin(np.array(c["batch_queries"], np.uint8), y, c["batch_k"], metric)
    for q in range(3):
        assert ids[q].tolist() == c["batch_ids"][q] and dis[q].tolist() == f32_of(c["batch_dists"][q]).tolist()
    alive = eval_filter(c["filter"], np.arange(c["rows"]))
    ids, dis = capi.knn_bin(np.array([c["query"]], np.uint8), y, c["k"], metric, alive=alive)
    m = len(c["filter_ids"])
    assert ids[0, :m].tolist() == c["filter_ids"] and dis[0, :m].tolist() == f32_of(c["filter_dists"]).tolist()
    assert ids[0, m] == -1
    if "lwd_ids" in c:
        alive = np.arange(c["rows"]) >= c["lwd_deleted_below"]
        ids, dis = capi.knn_bin(np.array([c["query"]], np.uint8), y, 10, metric, alive=alive)
        assert ids[0].tolist() == c["lwd_ids"] and dis[0].tolist() == f32_of(c["lwd_dists"]).tolist()
    # the same through the BinaryFLAT index object (rows resident, added in two chunks with their row offsets as labels; the
    # goldens of 00038 are produced with index types BinaryFLAT / BinaryMSTG)
    bix = capi.BinIndex(c["nbytes"], metric)
    half = c["rows"] // 2
    bix.add(y[:half], n[:half])
    bix.add(y[half:], n[half:])
    assert bix.num_data == c["rows"]
    ids, dis = bix.search(np.array([c["query"]], np.uint8), c["k"])
    assert ids[0].tolist() == c["ids"] and dis[0].tolist() == f32_of(c["dists"]).tolist()
    ids, dis = bix.search(np.array(c["batch_queries"], np.uint8), c["batch_k"])
    for q in range(3):
        assert ids[q].tolist() == c["batch_ids"][q] and dis[q].tolist() == f32_of(c["batch_dists"][q]).tolist()
    alive = eval_filter(c["filter"], np.arange(c["rows"]))
    ids, dis = bix.search(np.array([c["query"]], np.uint8), c["k"], alive=alive)
    assert ids[0, :m].tolist() == c["filter_ids"] and ids[0, m] == -1
    # labels that are not the storage order (a decoupled part's row ids): the filter and the results speak labels
    rng = np.random.default_rng(38)
    perm = rng.permutation(c["rows"])
    bix2 = capi.BinIndex(c["nbytes"], metric)
    bix2.add(y[perm], n[perm])
    ids, dis = bix2.search(np.array([c["query"]], np.uint8), c["k"], alive=alive)
    assert ids[0, :m].tolist() == c["filter_ids"] and dis[0, :m].tolist() == f32_of(c["filter_dists"]).tolist()
    bix.close()
    bix2.close()


@pytest.mark.parametrize("nbytes,ny,nx,k", [(4, 1024, 3, 10), (16, 5000, 2, 64), (32, 100000, 5, 10), (100, 20000, 1, 200),
                                            (128, 30000, 9, 30), (512, 4000, 2, 10), (1, 300, 2, 5), (33, 7, 1, 10)])
@pytest.mark.parametrize("metric", [capi.METRIC_HAMMING, capi.METRIC_JACCARD])
def test_knn_bin_matches_oracle(nbytes, ny, nx, k, metric):
    rng = np.random.default_rng(nbytes * 7 + ny + k)
    y = rng.integers(0, 256, (ny, nbytes), dtype=np.uint8)
    y[::17] = 0  # all-zero rows: Jaccard against a zero query is 1 by definition
    x = y[rng.integers(0, ny, nx)] ^ rng.integers(0, 4, (nx, nbytes), dtype=np.uint8)
    x[0] = 0
    om = o.METRIC_HAMMING if metric == capi.METRIC_HAMMING else o.METRIC_JACCARD
    ids, dis = capi.knn_bin(x, y, k, metric)
    oi, od = o.knn_bin(x, y, k, om)
    same(ids, dis, oi, od)
    alive = rng.random(ny) < 0.3
    ids, dis = capi.knn_bin(x, y, k, metric, alive=alive)
    oi, od = o.knn_bin(x, y, k, om, alive=alive)
    same(ids, dis, oi, od)
    with pytest.raises(capi.MsvsError) as e:
        capi.knn_bin(x, y, k, capi.METRIC_L2)
    assert e.value.code == capi.ERR_NOT_IMPLEMENTED


def test_scratch_arenas_shrink_and_release():
    """Round-1 review: the per-(thread, stream) arenas were grow-only.  One large batch followed by small ones gives the
    memory back (lazy shrink after a window of 64 small operations); msvs_release_scratch frees the thread's arenas at once."""
    import torch

    rng = np.random.default_rng(8)
    x = rng.standard_normal((20000, 64), dtype=np.float32)
    ix = build_ivf(x, capi.METRIC_L2, 32)
    big = rng.standard_normal((60000, 64), dtype=np.float32)
    capi.release_scratch()
    free0 = torch.cuda.mem_get_info()[0]
    i_big, d_big = ix.search(big, 100, "nprobe=32")  # hundreds of MB of partial lists
    held = free0 - torch.cuda.mem_get_info()[0]
    assert held > (100 << 20)
    for i in range(300):  # (8 queries: the general path, whose arena the large batch grew)
        ix.search(big[8 * i:8 * i + 8], 10, "nprobe=4")
    after = free0 - torch.cuda.mem_get_info()[0]
    assert after < held // 2, (held, after)
    assert capi.release_scratch() > 0
    i2, d2 = ix.search(big[:50], 100, "nprobe=32")  # and everything still works afterwards
    same(i2, d2, i_big[:50], d_big[:50])
