"""ctypes binding of the CPU oracle (oracle/msvs_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under myscaledb_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmsvs_oracle.so")

METRIC_L2, METRIC_IP, METRIC_COSINE, METRIC_HAMMING, METRIC_JACCARD = 0, 1, 2, 3, 4
FLT_MAX = np.finfo(np.float32).max


def build(force=False):
    src = os.path.join(_HERE, "msvs_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.oracle_l2sqr.restype = C.c_float
        _lib.oracle_ip.restype = C.c_float
        _lib.oracle_bm25_idf.restype = C.c_float
        _lib.oracle_fieldnorm_id.restype = C.c_uint8
        _lib.oracle_fieldnorm_of_id.restype = C.c_uint32
        for n in ("oracle_total_topk", "oracle_bm25_search", "oracle_bm25_search_ex", "oracle_hybrid_fusion"):
            getattr(_lib, n).restype = C.c_size_t
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def pack_bits(mask):
    """bool[n] -> LSB-first uint64 words (bit i of word i//64 = mask[i])."""
    mask = np.asarray(mask, dtype=bool)
    n = mask.size
    padded = np.zeros(((n + 63) // 64) * 64, dtype=bool)
    padded[:n] = mask
    return np.packbits(padded.reshape(-1, 8), axis=1, bitorder="little").reshape(-1).view(np.uint64).copy()


def l2sqr(x, y):
    x, y = _f32(x), _f32(y)
    return np.float32(lib().oracle_l2sqr(_p(x, C.c_float), _p(y, C.c_float), C.c_size_t(x.size)))


def ip(x, y):
    x, y = _f32(x), _f32(y)
    return np.float32(lib().oracle_ip(_p(x, C.c_float), _p(y, C.c_float), C.c_size_t(x.size)))


def normalize_rows(x):
    x = _f32(x).copy()
    x2 = x.reshape(-1, x.shape[-1])
    lib().oracle_normalize_rows(_p(x2, C.c_float), C.c_size_t(x2.shape[0]), C.c_size_t(x2.shape[1]))
    return x


def knn(x, y, k, metric, labels=None, alive=None, threads=0):
    x, y = _f32(x).reshape(-1, np.shape(y)[-1]), _f32(y)
    nx, d = x.shape
    ny = y.shape[0]
    ids = np.empty((nx, k), dtype=np.int64)
    dis = np.empty((nx, k), dtype=np.float32)
    if threads and labels is None and alive is None:
        rc = lib().oracle_knn_mt(_p(x, C.c_float), _p(y, C.c_float), C.c_size_t(d), C.c_size_t(k), C.c_size_t(nx),
                                 C.c_size_t(ny), metric, _p(ids, C.c_int64), _p(dis, C.c_float), threads)
    else:
        lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.int64)
        bits = None if alive is None else pack_bits(alive)
        rc = lib().oracle_knn(_p(x, C.c_float), _p(y, C.c_float), C.c_size_t(d), C.c_size_t(k), C.c_size_t(nx),
                              C.c_size_t(ny), metric, _p(lab, C.c_int64), _p(bits, C.c_uint64), _p(ids, C.c_int64),
                              _p(dis, C.c_float))
    if rc:
        raise NotImplementedError("oracle_knn rc=%d" % rc)
    return ids, dis


def search_without_index(x, y, k, metric, alive=None):
    """VIWithColumnInPart::searchWithoutIndex (cosine normalises copies of x and y)."""
    x, y = _f32(x).reshape(-1, np.shape(y)[-1]).copy(), _f32(y).copy()
    nx, d = x.shape
    ids = np.empty((nx, k), dtype=np.int64)
    dis = np.empty((nx, k), dtype=np.float32)
    bits = None if alive is None else pack_bits(alive)
    rc = lib().oracle_search_without_index(_p(x, C.c_float), _p(y, C.c_float), C.c_size_t(d), C.c_size_t(k),
                                           C.c_size_t(nx), C.c_size_t(y.shape[0]), metric, _p(bits, C.c_uint64),
                                           _p(ids, C.c_int64), _p(dis, C.c_float))
    if rc:
        raise NotImplementedError("oracle_search_without_index rc=%d" % rc)
    return ids, dis


def search_wrapper(query, base, k, metric, final_id, final_distance, num_rows_read=0, actual_id_in_range=None,
                   row_exists=None, delete_id_num=0):
    """MergeTreeVSManager::searchWrapper: merges one block into (final_id, final_distance) in place."""
    base = _f32(base).copy()
    query = _f32(query).reshape(-1, base.shape[1]).copy()
    nq, d = query.shape
    assert final_id.dtype == np.int64 and final_distance.dtype == np.float32
    act = None if actual_id_in_range is None else np.ascontiguousarray(actual_id_in_range, dtype=np.uint64)
    bits = None if row_exists is None else pack_bits(row_exists)
    rc = lib().oracle_search_wrapper(int(act is not None), _p(query, C.c_float), _p(base, C.c_float),
                                     C.c_size_t(base.shape[0]), int(k), int(d), int(nq), int(num_rows_read),
                                     _p(final_id, C.c_int64), _p(final_distance, C.c_float), _p(act, C.c_uint64),
                                     metric, _p(bits, C.c_uint64), int(delete_id_num))
    if rc:
        raise NotImplementedError("oracle_search_wrapper rc=%d" % rc)


def total_topk(scores, parts, labels, top_k, desc):
    scores = _f32(scores)
    parts = np.ascontiguousarray(parts, dtype=np.uint64)
    labels = np.ascontiguousarray(labels, dtype=np.uint64)
    n = scores.size
    os_, op, ol = np.empty(top_k, np.float32), np.empty(top_k, np.uint64), np.empty(top_k, np.uint64)
    cnt = lib().oracle_total_topk(_p(scores, C.c_float), _p(parts, C.c_uint64), _p(labels, C.c_uint64), C.c_size_t(n),
                                  C.c_size_t(top_k), int(desc), _p(os_, C.c_float), _p(op, C.c_uint64),
                                  _p(ol, C.c_uint64))
    return os_[:cnt], op[:cnt], ol[:cnt]


def ivf_search(centroids, list_off, vecs, ids, queries, nprobe, k, metric, alive=None, threads=0):
    centroids, vecs = _f32(centroids), _f32(vecs)
    d = vecs.shape[1]
    queries = _f32(queries).reshape(-1, d)
    list_off = np.ascontiguousarray(list_off, dtype=np.int64)
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    nq, nlist = queries.shape[0], centroids.shape[0]
    out_ids = np.empty((nq, k), np.int64)
    out_dis = np.empty((nq, k), np.float32)
    npb = min(nprobe, nlist)
    probes = np.empty((nq, npb), np.int64)
    if threads and alive is None:
        rc = lib().oracle_ivf_search_mt(_p(centroids, C.c_float), C.c_size_t(nlist), _p(list_off, C.c_int64),
                                        _p(vecs, C.c_float), _p(ids, C.c_int64), _p(queries, C.c_float),
                                        C.c_size_t(nq), C.c_size_t(d), C.c_size_t(nprobe), C.c_size_t(k), metric,
                                        _p(out_ids, C.c_int64), _p(out_dis, C.c_float), threads)
        probes = None
    else:
        bits = None if alive is None else pack_bits(alive)
        rc = lib().oracle_ivf_search(_p(centroids, C.c_float), C.c_size_t(nlist), _p(list_off, C.c_int64),
                                     _p(vecs, C.c_float), _p(ids, C.c_int64), _p(queries, C.c_float), C.c_size_t(nq),
                                     C.c_size_t(d), C.c_size_t(nprobe), C.c_size_t(k), metric, _p(bits, C.c_uint64),
                                     _p(out_ids, C.c_int64), _p(out_dis, C.c_float), _p(probes, C.c_int64))
    if rc:
        raise NotImplementedError("oracle_ivf_search rc=%d" % rc)
    return out_ids, out_dis, probes


def kmeans(x, nlist, iters=10):
    x = _f32(x)
    cent = np.empty((nlist, x.shape[1]), np.float32)
    lib().oracle_kmeans(_p(x, C.c_float), C.c_size_t(x.shape[0]), C.c_size_t(x.shape[1]), C.c_size_t(nlist),
                        int(iters), _p(cent, C.c_float))
    return cent


def assign(x, centroids):
    x, centroids = _f32(x), _f32(centroids)
    out = np.empty(x.shape[0], np.int64)
    lib().oracle_assign(_p(x, C.c_float), C.c_size_t(x.shape[0]), C.c_size_t(x.shape[1]), _p(centroids, C.c_float),
                        C.c_size_t(centroids.shape[0]), _p(out, C.c_int64))
    return out


def build_ivf(x, ids, centroids):
    """list-major layout (rows of each list in ascending id order) from a nearest-centroid assignment."""
    x = _f32(x)
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    a = assign(x, centroids)
    order = np.lexsort((ids, a))
    counts = np.bincount(a, minlength=centroids.shape[0])
    list_off = np.zeros(centroids.shape[0] + 1, np.int64)
    np.cumsum(counts, out=list_off[1:])
    return list_off, x[order].copy(), ids[order].copy()


def fieldnorm_id(length):
    return int(lib().oracle_fieldnorm_id(C.c_uint32(int(length))))


def fieldnorm_of_id(i):
    return int(lib().oracle_fieldnorm_of_id(C.c_uint8(int(i))))


def bm25_idf(df, n):
    return np.float32(lib().oracle_bm25_idf(C.c_uint64(int(df)), C.c_uint64(int(n))))


def bm25_search(post_off, doc_ids, tfs, fieldnorm_ids, qterms, df, total_docs, total_tokens, k, alive=None):
    post_off = np.ascontiguousarray(post_off, dtype=np.int64)
    doc_ids = np.ascontiguousarray(doc_ids, dtype=np.uint32)
    tfs = np.ascontiguousarray(tfs, dtype=np.uint32)
    fieldnorm_ids = np.ascontiguousarray(fieldnorm_ids, dtype=np.uint8)
    qterms = np.ascontiguousarray(qterms, dtype=np.uint32)
    df = np.ascontiguousarray(df, dtype=np.uint64)
    bits = None if alive is None else pack_bits(alive)
    rows, scores = np.empty(k, np.uint64), np.empty(k, np.float32)
    cnt = lib().oracle_bm25_search(_p(post_off, C.c_int64), _p(doc_ids, C.c_uint32), _p(tfs, C.c_uint32),
                                   _p(fieldnorm_ids, C.c_uint8), C.c_size_t(fieldnorm_ids.size),
                                   _p(qterms, C.c_uint32), _p(df, C.c_uint64), C.c_size_t(qterms.size),
                                   C.c_uint64(int(total_docs)), C.c_uint64(int(total_tokens)), _p(bits, C.c_uint64),
                                   C.c_size_t(k), _p(rows, C.c_uint64), _p(scores, C.c_float))
    return rows[:cnt], scores[:cnt]


def bm25_search_ex(post_off, doc_ids, tfs, fieldnorm_ids, qterms, df, total_docs, total_tokens, k, alive=None,
                   term_field=None, qgroups=None, operator_or=True):
    """fieldnorm_ids [num_fields, num_docs]; total_tokens [num_fields]; see oracle_bm25_search_ex."""
    post_off = np.ascontiguousarray(post_off, dtype=np.int64)
    doc_ids = np.ascontiguousarray(doc_ids, dtype=np.uint32)
    tfs = np.ascontiguousarray(tfs, dtype=np.uint32)
    fieldnorm_ids = np.ascontiguousarray(np.atleast_2d(fieldnorm_ids), dtype=np.uint8)
    num_fields, num_docs = fieldnorm_ids.shape
    qterms = np.ascontiguousarray(qterms, dtype=np.uint32)
    df = np.ascontiguousarray(df, dtype=np.uint64)
    tokens = np.ascontiguousarray(np.broadcast_to(np.asarray(total_tokens, np.uint64), (num_fields,)))
    tf_ = None if term_field is None else np.ascontiguousarray(term_field, np.uint8)
    qg = None if qgroups is None else np.ascontiguousarray(qgroups, np.uint32)
    bits = None if alive is None else pack_bits(alive)
    rows, scores = np.empty(k, np.uint64), np.empty(k, np.float32)
    cnt = lib().oracle_bm25_search_ex(_p(post_off, C.c_int64), _p(tf_, C.c_uint8), _p(doc_ids, C.c_uint32),
                                      _p(tfs, C.c_uint32), _p(fieldnorm_ids, C.c_uint8), C.c_size_t(num_fields),
                                      C.c_size_t(num_docs), _p(qterms, C.c_uint32), _p(qg, C.c_uint32),
                                      _p(df, C.c_uint64), C.c_size_t(qterms.size), C.c_uint64(int(total_docs)),
                                      _p(tokens, C.c_uint64), 1 if operator_or else 0, _p(bits, C.c_uint64),
                                      C.c_size_t(k), _p(rows, C.c_uint64), _p(scores, C.c_float))
    return rows[:cnt], scores[:cnt]


def hybrid_fusion(fusion_type, vec, txt, topk, fusion_k=60, fusion_weight=0.5, vector_scan_direction=1):
    """vec / txt: (scores, parts, labels) already ordered best-first. fusion_type 'rrf' | 'rsf'."""
    vs, vp, vl = _f32(vec[0]), np.ascontiguousarray(vec[1], np.uint64), np.ascontiguousarray(vec[2], np.uint64)
    ts, tp, tl = _f32(txt[0]), np.ascontiguousarray(txt[1], np.uint64), np.ascontiguousarray(txt[2], np.uint64)
    os_, op, ol = np.empty(topk, np.float32), np.empty(topk, np.uint64), np.empty(topk, np.uint64)
    cnt = lib().oracle_hybrid_fusion(1 if fusion_type == "rsf" else 0, _p(vs, C.c_float), _p(vp, C.c_uint64),
                                     _p(vl, C.c_uint64), C.c_size_t(vs.size), _p(ts, C.c_float), _p(tp, C.c_uint64),
                                     _p(tl, C.c_uint64), C.c_size_t(ts.size), C.c_uint64(int(fusion_k)),
                                     C.c_float(fusion_weight), int(vector_scan_direction), C.c_size_t(topk),
                                     _p(os_, C.c_float), _p(op, C.c_uint64), _p(ol, C.c_uint64))
    return os_[:cnt], op[:cnt], ol[:cnt]


# ---------------------------------------------------------------- pure-numpy restatement (cross-check of the C code)

def np_dot_canonical(p):
    """p: f32 products [..., d] -> W64-tree sum along the last axis (vectorised over leading axes)."""
    p = np.asarray(p, dtype=np.float32)
    d = p.shape[-1]
    acc = np.zeros(p.shape[:-1] + (64,), np.float32)
    for k0 in range(0, d, 64):
        w = min(64, d - k0)
        acc[..., :w] = acc[..., :w] + p[..., k0:k0 + w]
    s = 1
    while s < 64:
        acc[..., ::2 * s] = acc[..., ::2 * s] + acc[..., s::2 * s]
        s *= 2
    return acc[..., 0]


def np_l2sqr(x, Y):
    t = (np.asarray(Y, np.float32) - np.asarray(x, np.float32)[None, :]).astype(np.float32)
    t = (np.asarray(x, np.float32)[None, :] - np.asarray(Y, np.float32)).astype(np.float32)
    return np_dot_canonical((t * t).astype(np.float32))


def np_ip(x, Y):
    return np_dot_canonical((np.asarray(x, np.float32)[None, :] * np.asarray(Y, np.float32)).astype(np.float32))


def knn_bin(x, y, k, metric, alive=None):
    """Binary vectors: x [nx, nbytes], y [ny, nbytes] uint8 -> (ids int64 [nx, k], dis f32 [nx, k]); metric
    METRIC_HAMMING / METRIC_JACCARD (oracle_knn_bin)."""
    y = np.ascontiguousarray(y, np.uint8)
    nb = y.shape[1]
    x = np.ascontiguousarray(x, np.uint8).reshape(-1, nb)
    ids = np.empty((x.shape[0], k), np.int64)
    dis = np.empty((x.shape[0], k), np.float32)
    bits = None if alive is None else pack_bits(alive)
    rc = lib().oracle_knn_bin(_p(x, C.c_uint8), _p(y, C.c_uint8), C.c_size_t(nb), C.c_size_t(k), C.c_size_t(x.shape[0]),
                              C.c_size_t(y.shape[0]), int(metric), _p(bits, C.c_uint64), _p(ids, C.c_int64),
                              _p(dis, C.c_float))
    if rc:
        raise RuntimeError("oracle_knn_bin rc=%d" % rc)
    return ids, dis


# ------------------------------------------------------------------ CPU baseline (simd_baseline.c; bench.py only)

_simd = None


def simd_lib(native=True):
    """libmsvs_simd_native.so (built HERE, on the machine that runs the bench, -march=native) or the portable AVX2 build."""
    global _simd
    if _simd is None:
        path = os.path.join(_HERE, "_build", "libmsvs_simd.so")
        if native:
            try:
                subprocess.check_call(["make", "-C", _HERE, "-s", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                path = os.path.join(_HERE, "_build", "libmsvs_simd_native.so")
            except Exception:
                pass
        _simd = C.CDLL(path)
    return _simd


def simd_ivf_search(cent, off, vecs, ids, q, nprobe, k, metric, threads):
    cent, vecs, q = _f32(cent), _f32(vecs), _f32(q)
    off = np.ascontiguousarray(off, np.int64)
    ids = np.ascontiguousarray(ids, np.int64)
    oi = np.empty((q.shape[0], k), np.int64)
    od = np.empty((q.shape[0], k), np.float32)
    simd_lib().simd_ivf_search(_p(cent, C.c_float), C.c_size_t(cent.shape[0]), _p(off, C.c_int64), _p(vecs, C.c_float),
                               _p(ids, C.c_int64), _p(q, C.c_float), C.c_size_t(q.shape[0]), C.c_size_t(q.shape[1]),
                               C.c_size_t(nprobe), C.c_size_t(k), int(metric), _p(oi, C.c_int64), _p(od, C.c_float),
                               int(threads))
    return oi, od


def simd_knn(x, y, k, metric, threads):
    x, y = _f32(x), _f32(y)
    oi = np.empty((x.shape[0], k), np.int64)
    od = np.empty((x.shape[0], k), np.float32)
    simd_lib().simd_knn(_p(x, C.c_float), _p(y, C.c_float), C.c_size_t(y.shape[1]), C.c_size_t(k), C.c_size_t(x.shape[0]),
                        C.c_size_t(y.shape[0]), int(metric), _p(oi, C.c_int64), _p(od, C.c_float), int(threads))
    return oi, od


def simd_lanes():
    return int(simd_lib().simd_lanes())
