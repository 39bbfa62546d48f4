"""Host mirror (libmsvs_host.so): the pure-host pieces are checked on CPU against the oracle and the goldens; the
pieces that launch device work (searchWithoutIndex / searchWrapper) replay the brute-force goldens on the GPU."""
import numpy as np
import pytest

import myscaledb_amd.capi as capi
import myscaledb_amd.host as host
from golden_util import eval_filter, f32_of, load_goldens, materialize
from host_model import brute_force_part
from oracle import oracle as o

G = load_goldens()


def test_host_library_exports_every_declared_symbol():
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "include", "msvs_host.h")) as f:
        decl = sorted(set(re.findall(r"MSVS_HOST_API\s+[\w\s\*]+?\b(msvs_(?:host|text|fts)_\w+)\s*\(", f.read())))
    assert decl == sorted(host.SYMBOLS)
    for s in decl:
        assert hasattr(host.lib(), s)


def test_total_topk_matches_oracle_and_multimap_tie_order():
    rng = np.random.default_rng(4)
    s = rng.integers(0, 20, 200).astype(np.float32)  # many ties
    parts = np.repeat(np.arange(4), 50)
    labels = np.tile(np.arange(50), 4)
    for desc in (False, True):
        a = host.total_topk(s, parts, labels, 30, desc)
        b = o.total_topk(s, parts, labels, 30, desc)
        assert all((x == y).all() for x, y in zip(a, b))


def test_merge_topk_host_matches_canonical_order():
    rng = np.random.default_rng(5)
    ids = rng.permutation(3 * 4 * 10).reshape(3, 4, 10).astype(np.int64)
    dis = rng.integers(0, 6, (3, 4, 10)).astype(np.float32)
    ids[2, :, 7:] = -1
    for metric in (capi.METRIC_L2, capi.METRIC_IP):
        order = np.argsort(dis if metric == capi.METRIC_L2 else -dis, axis=2, kind="stable")
        si, sd = np.take_along_axis(ids, order, 2), np.take_along_axis(dis, order, 2)
        mi, md = host.merge_topk(si, sd, metric)
        for q in range(4):
            cand = [(d if metric == capi.METRIC_L2 else -d, i) for p in range(3) for i, d in zip(si[p, q], sd[p, q]) if i >= 0]
            cand.sort()
            assert mi[q].tolist() == [c[1] for c in cand[:10]]


def test_fusion_goldens_through_host_mirror():
    from test_oracle_golden import order_by_score_desc_id, text_search, vec_topk
    c = G["00040_hybrid"]
    docs = c["docs"]
    limit = c["limit"]
    vi, vd = vec_topk(docs, c["vec_query"], limit)
    tr, ts = text_search(docs, c["text_query"], limit)
    z = lambda n: np.zeros(n, np.uint64)
    for kind in ("rsf", "rrf"):
        s, p, l = host.hybrid_search(kind, (vd, z(len(vi)), vi), (ts, z(len(tr)), tr), limit)
        so, po, lo = o.hybrid_fusion(kind, (vd, z(len(vi)), vi), (ts, z(len(tr)), tr), limit)
        assert (s == so).all() and (l == lo).all()
        ids, sc = order_by_score_desc_id([docs[int(x)]["id"] for x in l], s, limit)
        assert ids == c[kind][0]
        assert np.array(sc, np.float32).tolist() == f32_of(c[kind][1]).tolist()


def test_merge_search_result_join():
    labels = np.array([40, 7, 19, 3], np.uint32)           # the part's search result, best first
    part_offsets = np.array([2, 3, 4, 19, 40, 41], np.uint64)  # rows the reader materialised
    assert host.merge_search_result(part_offsets, labels).tolist() == [-1, 3, -1, 2, 0, -1]


def _rows(vecs, empty):
    return [None if e else v.tolist() for v, e in zip(vecs, empty)]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["00012_brute_force", "00009_brute_force_filter", "00010_brute_force_filter",
                                  "00011_brute_force_filter", "00014_cosine_bruteforce"])
def test_brute_force_goldens_through_vector_scan_without_index(name):
    """The C++ per-mark loop (ColumnArray in, result columns out) against the reference's .reference files."""
    c = G[name]
    ids, vecs, empty = materialize(c["base"])
    filt = eval_filter(c["filter"], ids) if c.get("filter") else None
    labels, _, dist = host.vector_scan_without_index(_rows(vecs, empty), vecs.shape[1], c.get("index_granularity", 8192),
                                                     c["queries"], c["k"], capi.METRICS[c["metric"]], filt=filt)
    assert ids[labels].tolist() == c["ids"][0]
    assert dist.tolist() == f32_of(c["dists"][0]).tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["00002_batch_l2", "00002_batch_ip"])
def test_batch_distance_golden_through_vector_scan_without_index(name):
    c = G[name]
    k, nq = c["k"], len(c["queries"])
    per_query = [[] for _ in range(nq)]
    for segs in c["parts"]:
        ids, vecs, empty = materialize(segs)
        labels, qids, dist = host.vector_scan_without_index(_rows(vecs, empty), 3, 8192, c["queries"], k,
                                                            capi.METRICS[c["metric"]], is_batch=True)
        for l, q, d in zip(labels, qids, dist):
            per_query[int(q)].append((float(d), int(ids[l])))
    for qi in range(nq):  # ORDER BY dist.1, dist.2 [DESC] LIMIT 10 BY dist.1
        rows = sorted(per_query[qi], key=lambda t: -t[0] if c["metric"] == "IP" else t[0])[:k]
        assert [r[1] for r in rows] == c["ids"][qi]
        assert [np.float32(r[0]) for r in rows] == f32_of(c["dists"][qi]).tolist()


def test_sum_bm25_stats():
    per_part = np.array([[10, 73, 1, 0], [10, 70, 1, 2]], np.uint64)
    assert host.sum_bm25_stats(per_part).tolist() == [20, 143, 2, 2]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["00012_brute_force", "00009_brute_force_filter", "00010_brute_force_filter",
                                  "00011_brute_force_filter", "00014_cosine_bruteforce"])
def test_brute_force_goldens_through_host_search_wrapper(name):
    c = G[name]
    ids, vecs, empty = materialize(c["base"])
    filt = eval_filter(c["filter"], ids) if c.get("filter") else None
    fid, fdist = brute_force_part(host.search_wrapper, vecs, empty, c.get("index_granularity", 8192), c["queries"],
                                  c["k"], c["metric"], filt=filt)
    valid = fid[0] > -1
    assert ids[fid[0][valid]].tolist() == c["ids"][0]
    assert fdist[0][valid].tolist() == f32_of(c["dists"][0]).tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("metric", ["L2", "IP", "Cosine"])
def test_search_wrapper_with_lightweight_deletes_matches_oracle(metric):
    rng = np.random.default_rng(8)
    n, d, nq, k, gran = 3000, 48, 3, 7, 512
    vecs = rng.standard_normal((n, d), dtype=np.float32)
    empty = np.zeros(n, bool)
    q = rng.standard_normal((nq, d), dtype=np.float32)
    row_exists = rng.random(n) > 0.2
    a = brute_force_part(host.search_wrapper, vecs, empty, gran, q, k, metric, row_exists=row_exists)
    b = brute_force_part(o.search_wrapper, vecs, empty, gran, q, k, metric, row_exists=row_exists)
    assert (a[0] == b[0]).all() and (a[1].view(np.uint32) == b[1].view(np.uint32)).all()
    filt = rng.random(n) < 0.3
    a = brute_force_part(host.search_wrapper, vecs, empty, gran, q, k, metric, filt=filt)
    b = brute_force_part(o.search_wrapper, vecs, empty, gran, q, k, metric, filt=filt)
    assert (a[0] == b[0]).all() and (a[1].view(np.uint32) == b[1].view(np.uint32)).all()
    # heavy lightweight deletes: k + delete_id_num is in the thousands per mark (the oracle over-fetches like the
    # reference; the host mirror switches to the filtered scan) -- results must still be identical
    row_exists = rng.random(n) > 0.7
    a = brute_force_part(host.search_wrapper, vecs, empty, 1500, q, k, metric, row_exists=row_exists)
    b = brute_force_part(o.search_wrapper, vecs, empty, 1500, q, k, metric, row_exists=row_exists)
    assert (a[0] == b[0]).all() and (a[1].view(np.uint32) == b[1].view(np.uint32)).all()


def test_generate_vector_dataset_float64_and_batch_queries():
    """a2: distance() / batch_distance() query constants are Array(Float32 | Float64) [inside Array()]: flattened to
    nq x dim float32 by static_cast (round to nearest), a wrong length is an error (MergeTreeVSManager.cpp:59-181)."""
    rng = np.random.default_rng(8)
    d = 7
    q64 = rng.standard_normal(d) * 1e3 + 1e-9
    out = host.generate_vector_dataset(q64, None, d)
    assert out.dtype == np.float32 and (out[0] == q64.astype(np.float32)).all()
    # values that are not representable in float32 round to nearest even, huge ones go to inf like static_cast<float>
    edge = np.array([1 + 2.0 ** -24, 1 + 3 * 2.0 ** -24, 1e39, -1e39, 5e-46, 0.1, -0.0], np.float64)
    with np.errstate(over="ignore"):
        want = edge.astype(np.float32)
    assert (host.generate_vector_dataset(edge, None, 7)[0].view(np.uint32) == want.view(np.uint32)).all()
    # batch: Array(Array(Float64)) with ColumnArray offsets
    qs = rng.standard_normal((5, d))
    offs = np.arange(1, 6, dtype=np.uint64) * d
    got = host.generate_vector_dataset(qs.reshape(-1), offs, d)
    assert (got == qs.astype(np.float32)).all()
    got32 = host.generate_vector_dataset(qs.astype(np.float32).reshape(-1), offs, d)
    assert (got32 == qs.astype(np.float32)).all()
    bad = offs.copy()
    bad[2] -= 1  # one query a value short, the next one a value long
    with pytest.raises(capi.MsvsError) as e:
        host.generate_vector_dataset(qs.reshape(-1), bad, d)
    assert e.value.code == capi.ERR_INVALID_ARGUMENT


@pytest.mark.gpu
def test_float64_query_through_the_index_matches_float32_cast():
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2000, 16), dtype=np.float32)
    q64 = rng.standard_normal((3, 16))
    ix = capi.Index(capi.INDEX_FLAT, capi.METRIC_L2, 16)
    ix.add(x)
    ix.build()
    q = host.generate_vector_dataset(q64.reshape(-1), np.arange(1, 4, dtype=np.uint64) * 16, 16)
    ids, dis = ix.search(q, 10)
    oi, od = o.knn(q64.astype(np.float32), x, 10, o.METRIC_L2)
    assert (ids == oi).all() and (dis == od).all()


# ---------------------------------------------------------------------------------------- resident blocks (f1)

@pytest.mark.gpu
@pytest.mark.parametrize("name", ["00012_brute_force", "00009_brute_force_filter", "00010_brute_force_filter",
                                  "00011_brute_force_filter", "00014_cosine_bruteforce", "00016_lwd",
                                  "00032_lwd_small_ranges"])
def test_brute_force_goldens_through_resident_blocks(name):
    """The same goldens through the resident-block scan: the first query uploads the part's marks, the second one only
    sends its vector; filters and lightweight deletes are row bitmaps over the resident blocks."""
    c = G[name]
    ids, vecs, empty = materialize(c["base"])
    filt = eval_filter(c["filter"], ids) if c.get("filter") else None
    exists = ~np.isin(ids, c["deleted"]) if c.get("deleted") else None
    gran = c.get("index_granularity", 8192)
    cache = capi.Cache(64 << 20)
    marks = -(-len(ids) // gran)
    for round_ in range(2):
        labels, _, dist = host.vector_scan_without_index(_rows(vecs, empty), vecs.shape[1], gran, c["queries"], c["k"],
                                                         capi.METRICS[c["metric"]], filt=filt, row_exists=exists,
                                                         cache=cache, part_key="all_1_1_0/" + name)
        assert ids[labels].tolist() == c["ids"][0]
        assert dist.tolist() == f32_of(c["dists"][0]).tolist()
        st = cache.stats()
        assert st["blocks"] == marks and st["misses"] == marks and st["hits"] == round_ * marks
    assert cache.evict("all_1_1_0/") == marks and cache.stats()["blocks"] == 0 and cache.stats()["bytes"] == 0
    cache.close()


@pytest.mark.gpu
def test_resident_cache_lru_pins_and_eviction():
    rng = np.random.default_rng(3)
    d, n = 64, 1000
    blk = n * d * 4
    cache = capi.Cache(3 * blk + 100)  # room for three blocks
    xs = [rng.standard_normal((n, d), dtype=np.float32) for _ in range(5)]
    q = rng.standard_normal((2, d), dtype=np.float32)
    hs = [cache.upload("t/p1", m, xs[m]) for m in range(3)]
    for h in hs:
        cache.release(h)
    assert cache.stats()["blocks"] == 3
    h0 = cache.lookup("t/p1", 0)  # touch + pin mark 0: marks 1, 2 are now the least recently used
    h3 = cache.upload("t/p1", 3, xs[3])
    assert cache.lookup("t/p1", 1) is None and cache.stats()["evictions"] == 1
    h4 = cache.upload("t/p2", 4, xs[4])
    assert cache.lookup("t/p1", 2) is None  # evicted; the pinned mark 0 survived although it was the oldest upload
    ids, dis = capi.knn_resident(h0, q, 5, capi.METRIC_L2, d)
    oi, od = o.knn(q, xs[0], 5, o.METRIC_L2)
    assert (ids == oi).all() and (dis == od).all()
    alive = rng.random(n) < 0.5
    ids, dis = capi.knn_resident(h3, q, 5, capi.METRIC_IP, d, alive=alive)
    oi, od = o.knn(q, xs[3], 5, o.METRIC_IP, alive=alive)
    assert (ids == oi).all() and (dis == od).all()
    # an evicted-while-pinned block stays usable until its last release
    assert cache.evict("t/p1") == 2
    ids, dis = capi.knn_resident(h0, q, 5, capi.METRIC_L2, d)
    oi, od = o.knn(q, xs[0], 5, o.METRIC_L2)
    assert (ids == oi).all()
    for h in (h0, h3, h4):
        cache.release(h)
    st = cache.stats()
    assert st["blocks"] == 1 and st["bytes"] == blk
    # cosine: the block is stored normalised
    hc = cache.upload("t/cos", 0, xs[1], normalize=True)
    qn = o.normalize_rows(q)
    ids, dis = capi.knn_resident(hc, qn, 5, capi.METRIC_IP, d)
    oi, od = o.knn(qn, o.normalize_rows(xs[1]), 5, o.METRIC_IP)
    assert (ids == oi).all() and (dis == od).all()
    cache.release(hc)
    cache.close()


@pytest.mark.gpu
def test_resident_blocks_of_one_part_under_two_metrics_do_not_mix():
    """A per-query metric setting may search the same part under Cosine (block stored normalised) and under L2 / IP (stored
    raw): the two forms are different cache entries, each search gets its own, and both match the non-resident scan."""
    rng = np.random.default_rng(12)
    n, d, gran = 700, 24, 256
    vecs = (rng.standard_normal((n, d)) * rng.uniform(0.2, 5.0, (n, 1))).astype(np.float32)
    q = rng.standard_normal((1, d)).astype(np.float32)
    rows = [v for v in vecs]
    cache = capi.Cache(64 << 20)
    marks = -(-n // gran)
    for metric in (capi.METRIC_COSINE, capi.METRIC_L2, capi.METRIC_IP, capi.METRIC_COSINE):
        ref = host.vector_scan_without_index(rows, d, gran, q[0].tolist(), 7, metric)
        got = host.vector_scan_without_index(rows, d, gran, q[0].tolist(), 7, metric, cache=cache, part_key="p_1_1_0")
        assert got[0].tolist() == ref[0].tolist() and got[2].tolist() == ref[2].tolist()
    assert cache.stats()["blocks"] == 2 * marks  # one normalised and one raw copy per mark
    assert cache.evict("p_1_1_0") == 2 * marks
    # a stale block under the key of a part whose mark has another size now is an error, not a silent wrong answer
    h = cache.upload("q_1_1_0/raw", 0, vecs[:100])
    cache.release(h)
    with pytest.raises(Exception):
        host.vector_scan_without_index(rows, d, gran, q[0].tolist(), 7, capi.METRIC_L2, cache=cache, part_key="q_1_1_0")
    cache.close()


def test_default_tokenizer_on_non_ascii_text_matches_a_python_restatement():
    """SimpleTokenizer splits on code points that are not alphanumeric, LowerCaser lowercases beyond ASCII, RemoveLong drops
    tokens of 40 bytes or more: checked against Python's unicodedata on text with Unicode punctuation, NBSP, CJK, Cyrillic,
    Greek, full-width digits, an over-long token and malformed UTF-8."""
    import unicodedata

    def py_tokens(text):
        # RemoveLongFilter(40) sees the token as SimpleTokenizer cut it (its raw UTF-8 length), LowerCaser runs after it
        out, cur, raw = [], "", 0
        for ch in text:
            c = unicodedata.category(ch)
            if c[0] == "L" or c in ("Nd", "Nl", "No"):
                lo = ch.lower()
                cur += lo if ch == "\u0130" else (lo[0] if lo != ch else ch)  # U+0130 lower-cases to TWO code points (i + U+0307)
                raw += len(ch.encode("utf-8"))
            else:
                if cur and raw < 40:
                    out.append(cur)
                cur, raw = "", 0
        if cur and raw < 40:
            out.append(cur)
        return out

    texts = ["History's LESSONS \u2014 r\u00e9sum\u00e9\u00a0na\u00efve caf\u00c9, \u00dcBER-stra\u00dfe",
             "\u5317\u4eac\uff0c\u4e0a\u6d77\u3002\u6771\u4eac\u30bf\u30ef\u30fc 2024\u5e74", "\u041c\u043e\u0441\u043a\u0432\u0410 \u2013 \u0391\u0398\u0397\u039d\u0391 \u03c3\u03c4\u03b7\u03bd",
             "\uff11\uff12\uff13 abc\u00b2 x\u2082 \u2167 \u00bd", "a" * 39 + " " + "b" * 40 + " \u00e9" * 3 + " " + "\u00e9" * 20,
             "", "   \u3000\u2003 ", "\U0001d400\U0001d401 emoji \U0001f600 done",
             # lower-casing that changes the byte length: KELVIN SIGN (3 bytes -> k) in a token of exactly 40 raw bytes (dropped: the
             # filter runs first) and of 39 (kept); I WITH DOT ABOVE -> "i" + combining dot (2 -> 3 bytes) in a 39-byte token (kept)
             "x" * 37 + "\u212a yes", "x" * 36 + "\u212a yes", "\u0130stanbul " + "y" * 37 + "\u0130"]
    for t in texts:
        assert host.tokenize(t) == py_tokens(t), t
    # malformed sequences are separators: a lone continuation byte, a truncated 3-byte sequence, an overlong encoding
    assert host.tokenize(b"ab\x80cd \xe4\xb8 ef \xc0\xafgh") == ["ab", "cd", "ef", "gh"]


@pytest.mark.gpu
@pytest.mark.parametrize("typ", [capi.INDEX_FLAT, capi.INDEX_IVFFLAT])
def test_index_meta_delete_bitmap_and_decoupled_part_row_ids(typ):
    """VIWithMeta on a cached index: resident delete bitmap (goldens 00016 / 00032) and the row-id maps of a decoupled
    part -- the filter arrives in the merged part's row space (getRealBitmap), results leave in it (transferToNewRowIds)."""
    for name in ("00016_lwd", "00032_lwd_small_ranges"):
        c = G[name]
        ids, vecs, empty = materialize(c["base"])
        ix = capi.Index(typ, capi.METRIC_L2, 3, "ncentroids=4,kmeans_iters=3")
        if typ == capi.INDEX_IVFFLAT:
            ix.train(vecs)
        ix.add(vecs)
        ix.build()
        ix.set_delete_bitmap(~np.isin(ids, c["deleted"]))
        got, dist = ix.search(np.array(c["queries"], np.float32), c["k"], "nprobe=4" if typ == capi.INDEX_IVFFLAT else "")
        assert got[0].tolist() == c["ids"][0] and dist[0].tolist() == f32_of(c["dists"][0]).tolist()
        # a per-search filter is ANDed with it on the device
        flt = ids % 2 == 0
        got, _ = ix.search(np.array(c["queries"], np.float32), c["k"], "nprobe=4" if typ == capi.INDEX_IVFFLAT else "", alive=flt)
        oi, _ = o.knn(np.array(c["queries"], np.float32), vecs, c["k"], o.METRIC_L2, alive=flt & ~np.isin(ids, c["deleted"]))
        assert got[0].tolist() == oi[0].tolist()
        ix.set_delete_bitmap(None)
        got, _ = ix.search(np.array(c["queries"], np.float32), c["k"], "nprobe=4" if typ == capi.INDEX_IVFFLAT else "")
        oi, _ = o.knn(np.array(c["queries"], np.float32), vecs, c["k"], o.METRIC_L2)
        assert got[0].tolist() == oi[0].tolist()
    # decoupled part: this source part (own_id 1) owns the odd rows of a merged part of 2 n rows, shuffled
    rng = np.random.default_rng(12)
    n, d = 1500, 24
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((4, d), dtype=np.float32)
    new_rows = rng.permutation(2 * n)
    row_ids_map = np.sort(new_rows[:n])  # label i of the old part lives at merged row row_ids_map[i]
    inv_ids = np.zeros(2 * n, np.uint64)
    inv_src = np.zeros(2 * n, np.uint8)
    inv_ids[row_ids_map] = np.arange(n)
    inv_src[row_ids_map] = 1
    other = np.setdiff1d(np.arange(2 * n), row_ids_map)
    inv_ids[other] = np.arange(n)  # the other source part's labels
    inv_src[other] = 0
    ix = capi.Index(typ, capi.METRIC_L2, d, "ncentroids=8,kmeans_iters=3")
    if typ == capi.INDEX_IVFFLAT:
        ix.train(x)
    ix.add(x)
    ix.build()
    ix.set_merged_maps(row_ids_map, inv_ids, inv_src, 1)
    params = "nprobe=8" if typ == capi.INDEX_IVFFLAT else ""
    got, dist = ix.search(q, 10, params)
    oi, od = o.knn(q, x, 10, o.METRIC_L2)
    assert (got == row_ids_map[oi]).all() and (dist == od).all()
    flt_new = rng.random(2 * n) < 0.4  # a filter over the MERGED part's rows
    got, dist = ix.search(q, 10, params, alive=flt_new)
    oi, od = o.knn(q, x, 10, o.METRIC_L2, alive=flt_new[row_ids_map])
    assert (got == np.where(oi >= 0, row_ids_map[np.maximum(oi, 0)], -1)).all() and (dist == od).all()
    ix.set_delete_bitmap(np.arange(n) % 3 != 0)  # plus a delete bitmap in the OLD label space
    got, dist = ix.search(q, 10, params, alive=flt_new)
    oi, od = o.knn(q, x, 10, o.METRIC_L2, alive=flt_new[row_ids_map] & (np.arange(n) % 3 != 0))
    assert (got == np.where(oi >= 0, row_ids_map[np.maximum(oi, 0)], -1)).all() and (dist == od).all()


@pytest.mark.parametrize("kind", ["rrf", "rsf"])
@pytest.mark.parametrize("direction", [1, -1])
def test_distributed_fusion_transform_matches_the_oracle_fusion(kind, direction):
    """HybridSearchFusionTransform (the initiator's fusion of a Distributed-table hybrid search): row-range selection
    (last / first num_candidates distance rows by direction, first num_candidates bm25 rows), fusion keyed by
    (shard, part, offset), output order = bm25 rows then the remaining distance rows; scores == the oracle's fusion."""
    rng = np.random.default_rng(17 + direction)
    n_dist, n_txt, nc = 37, 29, 20
    ids = [(int(s), int(p), int(o)) for s, p, o in zip(rng.integers(1, 4, 200), rng.integers(0, 3, 200), rng.integers(0, 50, 200))]
    ids = list(dict.fromkeys(ids))
    d_ids, t_ids = ids[:n_dist], ids[20:20 + n_txt]  # 17 rows in both result sets
    d_scores = np.sort(rng.random(n_dist).astype(np.float32))[::-1].copy()  # the pipeline sorts by score DESC
    t_scores = np.sort(rng.random(n_txt).astype(np.float32) * 10)[::-1].copy()
    score = np.concatenate([d_scores, t_scores])
    stype = np.array([0] * n_dist + [1] * n_txt, np.uint8)
    allid = d_ids + t_ids
    rows, fused = host.fusion_transform(kind, score, stype, [i[0] for i in allid], [i[1] for i in allid], [i[2] for i in allid], nc,
                                        fusion_k=60, fusion_weight=0.3, vector_scan_direction=direction)
    # the oracle: best-first candidate lists, shard folded into the part number
    if direction == 1:  # smaller is better: the LAST nc distance rows, read backwards
        dsel = list(range(n_dist - 1, n_dist - 1 - nc, -1))
    else:
        dsel = list(range(nc))
    tsel = list(range(n_dist, n_dist + nc))
    key = lambda r: allid[r][0] * 1000 + allid[r][1]
    vs = (score[dsel], np.array([key(r) for r in dsel], np.uint64), np.array([allid[r][2] for r in dsel], np.uint64))
    ts = (score[tsel], np.array([key(r) for r in tsel], np.uint64), np.array([allid[r][2] for r in tsel], np.uint64))
    os_, op, ol = o.hybrid_fusion(kind, vs, ts, 2 * nc, fusion_k=60, fusion_weight=0.3, vector_scan_direction=direction)
    want = {(int(p), int(l)): np.float32(s) for s, p, l in zip(os_, op, ol)}
    got = {(key(int(r)), allid[int(r)][2]): np.float32(f) for r, f in zip(rows, fused)}
    assert got == want
    # order: every bm25 candidate row first, in place; then the distance rows that are not among them, in stored order
    assert rows[:nc].tolist() == tsel
    rest = [r for r in sorted(dsel) if allid[r] not in {allid[t] for t in tsel}]
    assert rows[nc:].tolist() == rest


def test_hybrid_search_batch_equals_the_per_query_fusion():
    """msvs_host_hybrid_search_batch over the device searches' [nq, k] output arrays (ids < 0 = no row) against
    msvs_host_hybrid_search query by query, RRF and RSF."""
    rng = np.random.default_rng(12)
    nq, kv, kt = 9, 20, 15
    vd = np.sort(rng.random((nq, kv)).astype(np.float32), axis=1)
    vi = np.stack([rng.permutation(60)[:kv] for _ in range(nq)]).astype(np.int64)
    td = -np.sort(-rng.random((nq, kt)).astype(np.float32) * 10, axis=1)
    ti = np.stack([rng.permutation(60)[:kt] for _ in range(nq)]).astype(np.int64)
    vi[3, 12:] = -1  # short lists
    ti[5, 4:] = -1
    ti[7, :] = -1
    # ties: equal scores inside a list, the same label at the same ranks of both lists, a label twice in one list
    vd[1] = np.repeat(np.arange(kv // 4, dtype=np.float32), 4)[:kv]
    td[1] = 3.0
    ti[1, :kt] = vi[1, :kt]
    vi[2, 5] = vi[2, 1]
    ti[2, 3] = ti[2, 0]
    vd[4] = 0.5  # all distances equal: normalised scores are all 1
    z = np.zeros(max(kv, kt), np.uint64)
    for fusion, direction in (("rrf", 1), ("rsf", 1), ("rsf", -1)):
        bs, bl, bn = host.hybrid_search_batch(fusion, vd, vi, td, ti, 10, fusion_k=60, fusion_weight=0.3, vector_scan_direction=direction)
        for q in range(nq):
            nv, nt = int((vi[q] >= 0).sum()), int((ti[q] >= 0).sum())
            s1, _, l1 = host.hybrid_search(fusion, (vd[q][:nv], z[:nv], vi[q][:nv].astype(np.uint64)),
                                           (td[q][:nt], z[:nt], ti[q][:nt].astype(np.uint64)), 10, fusion_k=60, fusion_weight=0.3,
                                           vector_scan_direction=direction)
            assert bn[q] == len(l1)
            assert bl[q][:bn[q]].tolist() == l1.tolist()
            assert bs[q][:bn[q]].view(np.uint32).tolist() == s1.view(np.uint32).tolist()


def test_hybrid_search_batch_random_ties_against_the_map_based_fusion():
    """Quantised scores and a small label range: many equal fused scores, most labels in both lists."""
    rng = np.random.default_rng(77)
    nq, kv, kt = 120, 12, 9
    vd = np.sort(np.round(rng.random((nq, kv)) * 4) / 4, axis=1).astype(np.float32)
    td = -np.sort(-np.round(rng.random((nq, kt)) * 3), axis=1).astype(np.float32)
    vi = np.stack([rng.permutation(16)[:kv] for _ in range(nq)]).astype(np.int64)
    ti = np.stack([rng.permutation(16)[:kt] for _ in range(nq)]).astype(np.int64)
    for q in range(0, nq, 7):
        vi[q, rng.integers(3, kv):] = -1
    z = np.zeros(max(kv, kt), np.uint64)
    for fusion, direction, w in (("rrf", 1, 0.5), ("rsf", 1, 0.5), ("rsf", -1, 0.25)):
        bs, bl, bn = host.hybrid_search_batch(fusion, vd, vi, td, ti, 7, fusion_k=3, fusion_weight=w, vector_scan_direction=direction)
        for q in range(nq):
            nv, nt = int((vi[q] >= 0).sum()), int((ti[q] >= 0).sum())
            s1, _, l1 = host.hybrid_search(fusion, (vd[q][:nv], z[:nv], vi[q][:nv].astype(np.uint64)),
                                           (td[q][:nt], z[:nt], ti[q][:nt].astype(np.uint64)), 7, fusion_k=3, fusion_weight=w,
                                           vector_scan_direction=direction)
            assert bn[q] == len(l1) and bl[q][:bn[q]].tolist() == l1.tolist(), (fusion, q)
            assert bs[q][:bn[q]].view(np.uint32).tolist() == s1.view(np.uint32).tolist()

