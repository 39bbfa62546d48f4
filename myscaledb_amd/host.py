"""ctypes binding of libmsvs_host.so (include/msvs_host.h): the host-side mirror of the reference's operator
interface (searchWithoutIndex, searchWrapper, getTotalTopSearchResultImpl, hybridSearch, ...).  Plumbing only."""
import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmsvs_host.so")

SYMBOLS = ["msvs_host_search_without_index", "msvs_host_search_wrapper", "msvs_host_total_topk",
           "msvs_host_hybrid_search", "msvs_host_merge_topk", "msvs_host_sum_bm25_stats",
           "msvs_host_vector_scan_without_index", "msvs_host_merge_search_result",
           "msvs_host_generate_vector_dataset", "msvs_host_vector_scan_resident", "msvs_host_fusion_transform",
           "msvs_text_last_error", "msvs_text_index_create", "msvs_text_index_free", "msvs_text_index_add_doc",
           "msvs_text_index_commit", "msvs_text_index_save", "msvs_text_index_load", "msvs_text_index_total_num_docs",
           "msvs_text_index_total_num_tokens", "msvs_text_index_doc_freq", "msvs_text_index_set_alive",
           "msvs_text_index_bm25_search", "msvs_text_index_bm25_search_batch", "msvs_host_fts_index_statistics",
           "msvs_host_fts_statistics_merge", "msvs_fts_stats_view", "msvs_fts_stats_free", "msvs_host_concurrent_search",
           "msvs_host_hybrid_search_batch", "msvs_text_tokenize",
           "msvs_host_all_reduce_bm25_stats"]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libmsvs_host.so is not built (%s)" % LIB_PATH)
        capi.lib()  # libmsvs.so first (the host library links against it)
        _lib = C.CDLL(LIB_PATH)
        _lib.msvs_host_total_topk.restype = C.c_size_t
        _lib.msvs_host_hybrid_search.restype = C.c_size_t
        _lib.msvs_host_fusion_transform.restype = C.c_size_t
        _lib.msvs_host_sum_bm25_stats.restype = None
        _lib.msvs_text_last_error.restype = C.c_char_p
        _lib.msvs_text_tokenize.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
        _lib.msvs_text_index_total_num_docs.restype = C.c_uint64
        _lib.msvs_text_index_total_num_docs.argtypes = [C.c_void_p]
        _lib.msvs_text_index_free.restype = None
        _lib.msvs_text_index_free.argtypes = [C.c_void_p]
        _lib.msvs_fts_stats_view.restype = C.POINTER(_Stats)
        _lib.msvs_fts_stats_view.argtypes = [C.c_void_p]
        _lib.msvs_fts_stats_free.restype = None
        _lib.msvs_fts_stats_free.argtypes = [C.c_void_p]
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def search_without_index(x, y, k, metric):
    """VIWithColumnInPart::searchWithoutIndex; works on copies (the reference normalises in place)."""
    y = _f32(y).copy()
    x = _f32(x).reshape(-1, y.shape[1]).copy()
    nq, d = x.shape
    ids = np.empty((nq, k), np.int64)
    dis = np.empty((nq, k), np.float32)
    capi._check(lib().msvs_host_search_without_index(_p(x, C.c_float), _p(y, C.c_float), C.c_size_t(d), C.c_size_t(k),
                                                     C.c_size_t(nq), C.c_size_t(y.shape[0]), int(metric),
                                                     _p(ids, C.c_int64), _p(dis, C.c_float)))
    return ids, dis


def search_wrapper(query, base, k, metric, final_id, final_distance, num_rows_read=0, actual_id_in_range=None,
                   row_exists=None, delete_id_num=0):
    """MergeTreeVSManager::searchWrapper: merges one block into (final_id, final_distance) in place."""
    base = _f32(base).copy()
    d = base.shape[1]
    query = _f32(query).reshape(-1, d).copy()
    nq = query.shape[0]
    act = None if actual_id_in_range is None else _u64(actual_id_in_range)
    bits = None if row_exists is None else capi.pack_bits(row_exists)
    capi._check(lib().msvs_host_search_wrapper(int(act is not None), _p(query, C.c_float), _p(base, C.c_float),
                                               C.c_size_t(base.shape[0]), int(k), int(d), int(nq), int(num_rows_read),
                                               _p(final_id, C.c_int64), _p(final_distance, C.c_float),
                                               _p(act, C.c_uint64), int(metric), _p(bits, C.c_uint64),
                                               int(delete_id_num)))


def vector_scan_without_index(rows, dim, index_granularity, queries, k, metric, is_batch=False, filt=None,
                              row_exists=None, cache=None, part_key=""):
    """MergeTreeVSManager::vectorScanWithoutIndex over one part given as a list of per-row vectors (an empty list /
    None = a row without a vector), converted to the ColumnArray layout (offsets + flat data).
    cache (capi.Cache) + part_key: the resident-block variant (msvs_host_vector_scan_resident).
    -> (labels uint32[n], query_ids uint32[n] or None, distances f32[n])"""
    offsets = np.zeros(len(rows), np.uint64)
    flat = []
    end = 0
    for i, r in enumerate(rows):
        if r is not None and len(r):
            flat.extend(r)
            end += len(r)
        offsets[i] = end
    data = np.asarray(flat, np.float64).astype(np.float32) if flat else np.zeros(1, np.float32)
    q = _f32(queries).reshape(-1, dim)
    nq = q.shape[0]
    fb = None if filt is None else capi.pack_bits(filt)
    eb = None if row_exists is None else capi.pack_bits(row_exists)
    labels = np.empty(nq * k, np.uint32)
    qids = np.empty(nq * k, np.uint32)
    dist = np.empty(nq * k, np.float32)
    n = C.c_size_t(0)
    if cache is not None:
        capi._check(lib().msvs_host_vector_scan_resident(
            cache._h, part_key.encode(), _p(offsets, C.c_uint64), _p(data, C.c_float), C.c_size_t(len(rows)),
            C.c_size_t(dim), C.c_size_t(index_granularity), _p(q, C.c_float), C.c_size_t(nq), int(k), int(metric),
            int(is_batch), _p(fb, C.c_uint64), _p(eb, C.c_uint64), _p(labels, C.c_uint32), _p(qids, C.c_uint32),
            _p(dist, C.c_float), C.byref(n)))
        return labels[:n.value], (qids[:n.value] if is_batch else None), dist[:n.value]
    capi._check(lib().msvs_host_vector_scan_without_index(
        _p(offsets, C.c_uint64), _p(data, C.c_float), C.c_size_t(len(rows)), C.c_size_t(dim),
        C.c_size_t(index_granularity), _p(q, C.c_float), C.c_size_t(nq), int(k), int(metric), int(is_batch),
        _p(fb, C.c_uint64), _p(eb, C.c_uint64), _p(labels, C.c_uint32), _p(qids, C.c_uint32), _p(dist, C.c_float),
        C.byref(n)))
    return labels[:n.value], (qids[:n.value] if is_batch else None), dist[:n.value]


def merge_search_result(part_offsets, labels):
    po = _u64(part_offsets)
    lb = np.ascontiguousarray(labels, np.uint32)
    out = np.empty(po.size, np.int64)
    lib().msvs_host_merge_search_result(_p(po, C.c_uint64), C.c_size_t(po.size), _p(lb, C.c_uint32),
                                        C.c_size_t(lb.size), _p(out, C.c_int64))
    return out


def total_topk(scores, parts, labels, top_k, desc):
    scores, parts, labels = _f32(scores), _u64(parts), _u64(labels)
    os_, op, ol = np.empty(top_k, np.float32), np.empty(top_k, np.uint64), np.empty(top_k, np.uint64)
    n = lib().msvs_host_total_topk(_p(scores, C.c_float), _p(parts, C.c_uint64), _p(labels, C.c_uint64),
                                   C.c_size_t(scores.size), C.c_size_t(top_k), int(desc), _p(os_, C.c_float),
                                   _p(op, C.c_uint64), _p(ol, C.c_uint64))
    return os_[:n], op[:n], ol[:n]


def hybrid_search(fusion_type, vec, txt, topk, fusion_k=60, fusion_weight=0.5, vector_scan_direction=1):
    vs, vp, vl = _f32(vec[0]), _u64(vec[1]), _u64(vec[2])
    ts, tp, tl = _f32(txt[0]), _u64(txt[1]), _u64(txt[2])
    os_, op, ol = np.empty(topk, np.float32), np.empty(topk, np.uint64), np.empty(topk, np.uint64)
    n = lib().msvs_host_hybrid_search(1 if fusion_type == "rsf" else 0, _p(vs, C.c_float), _p(vp, C.c_uint64),
                                      _p(vl, C.c_uint64), C.c_size_t(vs.size), _p(ts, C.c_float), _p(tp, C.c_uint64),
                                      _p(tl, C.c_uint64), C.c_size_t(ts.size), C.c_uint64(int(fusion_k)),
                                      C.c_float(fusion_weight), int(vector_scan_direction), C.c_size_t(topk),
                                      _p(os_, C.c_float), _p(op, C.c_uint64), _p(ol, C.c_uint64))
    return os_[:n], op[:n], ol[:n]


def hybrid_search_batch(fusion_type, vec_dis, vec_ids, txt_scores, txt_ids, topk, fusion_k=60, fusion_weight=0.5,
                        vector_scan_direction=1):
    """msvs_host_hybrid_search_batch: [nq, kv] vector rows + [nq, kt] text rows (ids < 0 = no row) -> (scores [nq, topk],
    labels [nq, topk], counts [nq])."""
    vd, vi = _f32(vec_dis), np.ascontiguousarray(vec_ids, np.int64)
    ts, ti = _f32(txt_scores), np.ascontiguousarray(txt_ids, np.int64)
    nq, kv = vi.shape
    kt = ti.shape[1]
    os_, ol, cnt = np.zeros((nq, topk), np.float32), np.zeros((nq, topk), np.uint64), np.zeros(nq, np.uint32)
    rc = lib().msvs_host_hybrid_search_batch(1 if fusion_type == "rsf" else 0, _p(vd, C.c_float), _p(vi, C.c_int64), C.c_size_t(kv),
                                             _p(ts, C.c_float), _p(ti, C.c_int64), C.c_size_t(kt), C.c_size_t(nq),
                                             C.c_uint64(int(fusion_k)), C.c_float(fusion_weight), int(vector_scan_direction),
                                             C.c_size_t(topk), _p(os_, C.c_float), _p(ol, C.c_uint64), _p(cnt, C.c_uint32))
    if rc != 0:
        raise capi.MsvsError(rc, "msvs_host_hybrid_search_batch: invalid argument")
    return os_, ol, cnt


def fusion_transform(fusion_type, score, score_type, shard_num, part_index, part_offset, num_candidates, fusion_k=60,
                     fusion_weight=0.5, vector_scan_direction=1):
    """HybridSearchFusionTransform::generate over the merged shard rows (distance rows first, then bm25 rows) ->
    (indices into the rows, fused scores)."""
    score = _f32(score)
    st = np.ascontiguousarray(score_type, np.uint8)
    sh = np.ascontiguousarray(shard_num, np.uint32)
    pi, po = _u64(part_index), _u64(part_offset)
    n = score.size
    out_rows, out_scores = np.empty(2 * max(1, int(num_candidates)), np.uint64), np.empty(2 * max(1, int(num_candidates)), np.float32)
    cnt = lib().msvs_host_fusion_transform(1 if fusion_type == "rsf" else 0, _p(score, C.c_float), _p(st, C.c_uint8), _p(sh, C.c_uint32),
                                           _p(pi, C.c_uint64), _p(po, C.c_uint64), C.c_size_t(n), C.c_uint64(int(num_candidates)),
                                           C.c_uint64(int(fusion_k)), C.c_float(fusion_weight), int(vector_scan_direction),
                                           _p(out_rows, C.c_uint64), _p(out_scores, C.c_float))
    return out_rows[:cnt].astype(np.int64), out_scores[:cnt]


def merge_topk(ids, dis, metric):
    ids = np.ascontiguousarray(ids, np.int64)
    dis = _f32(dis)
    nparts, nq, k = ids.shape
    oi, od = np.empty((nq, k), np.int64), np.empty((nq, k), np.float32)
    capi._check(lib().msvs_host_merge_topk(_p(ids, C.c_int64), _p(dis, C.c_float), C.c_size_t(nparts), C.c_size_t(nq),
                                           C.c_size_t(k), int(metric), _p(oi, C.c_int64), _p(od, C.c_float)))
    return oi, od


def sum_bm25_stats(per_part):
    a = _u64(per_part)
    out = np.empty(a.shape[1], np.uint64)
    lib().msvs_host_sum_bm25_stats(_p(a, C.c_uint64), C.c_size_t(a.shape[0]), C.c_size_t(a.shape[1] - 2),
                                   _p(out, C.c_uint64))
    return out


def generate_vector_dataset(values, offsets, dim):
    """MergeTreeVSManager::generateVectorDataset: the query column (float32 or float64 values + ColumnArray offsets, or
    None for a single query) -> nq x dim float32."""
    values = np.ascontiguousarray(values)
    is64 = values.dtype == np.float64
    if not is64:
        values = np.ascontiguousarray(values, np.float32)
    if offsets is None:
        nq, offp = 1, None
    else:
        offsets = np.ascontiguousarray(offsets, np.uint64)
        nq, offp = offsets.size, offsets.ctypes.data_as(C.POINTER(C.c_uint64))
    out = np.empty((nq, dim), np.float32)
    rc = lib().msvs_host_generate_vector_dataset(values.ctypes.data_as(C.c_void_p), int(is64), offp, C.c_size_t(nq),
                                                 C.c_size_t(dim), out.ctypes.data_as(C.POINTER(C.c_float)))
    if rc != 0:
        raise capi.MsvsError(rc, "generateVectorDataset failed")
    return out


# ------------------------------------------------------------------------------------------------ seam B host side

class _DocFreq(C.Structure):
    _fields_ = [("term", C.c_char_p), ("field_id", C.c_uint32), ("doc_freq", C.c_uint64)]


class _FieldTokens(C.Structure):
    _fields_ = [("field_id", C.c_uint32), ("field_total_tokens", C.c_uint64)]


class _Stats(C.Structure):
    _fields_ = [("docs_freq", C.POINTER(_DocFreq)), ("n_docs_freq", C.c_size_t), ("total_num_tokens", C.POINTER(_FieldTokens)),
                ("n_fields", C.c_size_t), ("total_num_docs", C.c_uint64)]


def _tcheck(rc):
    if rc != 0:
        raise capi.MsvsError(rc, lib().msvs_text_last_error().decode())


def _strs(xs):
    arr = (C.c_char_p * max(1, len(xs)))()
    for i, x in enumerate(xs):
        arr[i] = x.encode()
    return arr


class Statistics:
    """TANTIVY::Statistics: table-level (term, field_id, doc_freq) triples, (field_id, total tokens) pairs, total docs."""

    def __init__(self, docs_freq, total_num_tokens, total_num_docs):
        self.docs_freq, self.total_num_tokens, self.total_num_docs = list(docs_freq), list(total_num_tokens), int(total_num_docs)

    def _c(self):
        df = (_DocFreq * max(1, len(self.docs_freq)))()
        for i, (t, f, d) in enumerate(self.docs_freq):
            df[i] = _DocFreq(t.encode(), f, d)
        tk = (_FieldTokens * max(1, len(self.total_num_tokens)))()
        for i, (f, n) in enumerate(self.total_num_tokens):
            tk[i] = _FieldTokens(f, n)
        st = _Stats(df, len(self.docs_freq), tk, len(self.total_num_tokens), self.total_num_docs)
        st._keep = (df, tk)
        return st

    @staticmethod
    def sum(parts):
        """BM25InfoInDataParts: add the parts' statistics up (src/VectorIndex/Common/BM25InfoInDataParts.cpp:40-93)."""
        df, tk, n = {}, {}, 0
        for p in parts:
            n += p.total_num_docs
            for t, f, d in p.docs_freq:
                df[(t, f)] = df.get((t, f), 0) + d
            for f, c in p.total_num_tokens:
                tk[f] = tk.get(f, 0) + c
        return Statistics([(t, f, d) for (t, f), d in df.items()], list(tk.items()), n)


def concurrent_search(index, queries, threads, calls_per_thread, k, params=""):
    """msvs_host_concurrent_search: native threads, one query per msvs_index_search call.
    -> (seconds, per-call latencies in us [threads * calls], ids [nq, k], dis [nq, k] of each query's last call)."""
    q = np.ascontiguousarray(queries, np.float32)
    nq, d = q.shape
    sec = C.c_double(0)
    lat = np.zeros(threads * calls_per_thread, np.float32)
    ids = np.full((nq, k), -2, np.int64)
    dis = np.zeros((nq, k), np.float32)
    rc = lib().msvs_host_concurrent_search(index._h, _p(q, C.c_float), C.c_size_t(nq), C.c_size_t(d), int(threads),
                                           C.c_size_t(calls_per_thread), int(k), params.encode(), C.byref(sec), _p(lat, C.c_float),
                                           _p(ids, C.c_int64), _p(dis, C.c_float))
    if rc != 0:
        raise capi.MsvsError(rc, capi.last_error())
    return sec.value, lat, ids, dis


def _stats_from_handle(h):
    v = lib().msvs_fts_stats_view(h).contents
    st = Statistics([(v.docs_freq[i].term.decode(), v.docs_freq[i].field_id, v.docs_freq[i].doc_freq) for i in range(v.n_docs_freq)],
                    [(v.total_num_tokens[i].field_id, v.total_num_tokens[i].field_total_tokens) for i in range(v.n_fields)],
                    v.total_num_docs)
    lib().msvs_fts_stats_free(h)
    return st


def fts_index_statistics(parts, query_text):
    """The row one shard answers to ftsIndex(db, table, column, query_text): its parts' statistics summed
    (msvs_host_fts_index_statistics; StorageFtsIndex.cpp:150-213).  parts: TextIndexStore objects."""
    arr = (C.c_void_p * max(1, len(parts)))(*[p._h for p in parts])
    h = C.c_void_p()
    _tcheck(lib().msvs_host_fts_index_statistics(arr, C.c_size_t(len(parts)), query_text.encode(), C.byref(h)))
    return _stats_from_handle(h)


def fts_statistics_merge(rows):
    """The initiator's sum over the shards' rows (msvs_host_fts_statistics_merge; CommonUtils.cpp:190-330)."""
    cs = [r._c() for r in rows]
    arr = (C.POINTER(_Stats) * max(1, len(cs)))(*[C.pointer(c) for c in cs])
    h = C.c_void_p()
    _tcheck(lib().msvs_host_fts_statistics_merge(arr, C.c_size_t(len(cs)), C.byref(h)))
    return _stats_from_handle(h)


class TextIndexStore:
    """TantivyIndexStore's search-side interface over a postings export (myscaledb_amd/host/text_store.cpp)."""

    def __init__(self, column_names=None, _handle=None):
        if _handle is not None:
            self._h = _handle
            return
        h = C.c_void_p()
        _tcheck(lib().msvs_text_index_create(_strs(column_names), C.c_size_t(len(column_names)), C.byref(h)))
        self._h = h

    @classmethod
    def load(cls, path):
        h = C.c_void_p()
        _tcheck(lib().msvs_text_index_load(path.encode(), C.byref(h)))
        return cls(_handle=h)

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.msvs_text_index_free(self._h)
            self._h = None

    __del__ = close

    def add_doc(self, row_id, column_names, docs):
        _tcheck(lib().msvs_text_index_add_doc(self._h, C.c_uint64(row_id), _strs(column_names), _strs(docs),
                                              C.c_size_t(len(docs))))

    def commit(self):
        _tcheck(lib().msvs_text_index_commit(self._h))

    def save(self, path):
        _tcheck(lib().msvs_text_index_save(self._h, path.encode()))

    def total_num_docs(self):
        return int(lib().msvs_text_index_total_num_docs(self._h))

    def total_num_tokens(self):
        out, n = (_FieldTokens * 4)(), C.c_size_t(0)
        _tcheck(lib().msvs_text_index_total_num_tokens(self._h, out, C.c_size_t(4), C.byref(n)))
        return [(out[i].field_id, out[i].field_total_tokens) for i in range(n.value)]

    def doc_freq(self, sentence):
        out, n = (_DocFreq * 256)(), C.c_size_t(0)
        _tcheck(lib().msvs_text_index_doc_freq(self._h, sentence.encode(), out, C.c_size_t(256), C.byref(n)))
        return [(out[i].term.decode(), out[i].field_id, out[i].doc_freq) for i in range(min(n.value, 256))]

    def statistics(self, sentence):
        return Statistics(self.doc_freq(sentence), self.total_num_tokens(), self.total_num_docs())

    def set_alive(self, alive):
        if alive is None:
            _tcheck(lib().msvs_text_index_set_alive(self._h, None, C.c_size_t(0)))
        else:
            b = np.packbits(np.asarray(alive, bool), bitorder="little")
            _tcheck(lib().msvs_text_index_set_alive(self._h, _p(b, C.c_uint8), C.c_size_t(b.size)))

    def bm25_search_batch(self, sentences, topk, statistics=None, column_names=None, alive=None, enable_nlq=False,
                          operator_or=True):
        """ffi_bm25_search for a list of sentences -> [(row_ids, scores)]; alive: bool per row (use_filter = True)."""
        nq = len(sentences)
        rows, scores, cnt = np.empty((nq, topk), np.uint64), np.empty((nq, topk), np.float32), np.zeros(nq, np.uint32)
        b = None if alive is None else np.packbits(np.asarray(alive, bool), bitorder="little")
        st = None if statistics is None else statistics._c()
        cols = column_names or []
        _tcheck(lib().msvs_text_index_bm25_search_batch(
            self._h, _strs(sentences), C.c_size_t(nq), _strs(cols) if cols else None, C.c_size_t(len(cols)), C.c_uint32(topk),
            _p(b, C.c_uint8), C.c_size_t(0 if b is None else b.size), int(alive is not None), int(enable_nlq), int(operator_or),
            None if st is None else C.byref(st), _p(rows, C.c_uint64), _p(scores, C.c_float), _p(cnt, C.c_uint32)))
        return [(rows[q, :cnt[q]].copy(), scores[q, :cnt[q]].copy()) for q in range(nq)]

    def bm25_search(self, sentence, topk, statistics=None, column_names=None, alive=None, enable_nlq=False, operator_or=True):
        return self.bm25_search_batch([sentence], topk, statistics, column_names, alive, enable_nlq, operator_or)[0]


def tokenize(text):
    """Tokens of `text` (str) under the text store's default tokenizer chain (msvs_text_tokenize)."""
    raw = text.encode("utf-8", errors="surrogatepass") if isinstance(text, str) else bytes(text)
    need = C.c_size_t(0)
    buf = C.create_string_buffer(max(64, 2 * len(raw) + 8))
    rc = lib().msvs_text_tokenize(raw, buf, len(buf), C.byref(need))
    if rc != 0:
        raise RuntimeError(lib().msvs_text_last_error().decode())
    if need.value > len(buf):
        buf = C.create_string_buffer(need.value)
        lib().msvs_text_tokenize(raw, buf, len(buf), C.byref(need))
    s = buf.value.decode("utf-8")
    return s.split("\n") if s else []


def all_reduce_bm25_stats(comm, total_docs, total_tokens, df, stream=None):
    """(N, total tokens, df per term) summed over the ranks of a capi.Comm (msvs_host_all_reduce_bm25_stats)."""
    v = np.array([int(total_docs), int(total_tokens)] + [int(x) for x in df], np.uint64)
    rc = lib().msvs_host_all_reduce_bm25_stats(comm._h, v.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_size_t(len(df)),
                                               C.c_void_p(stream) if stream else None)
    if rc != 0:
        raise capi.MsvsError(rc, capi.lib().msvs_last_error().decode())
    out = [int(x) for x in v]
    return out[0], out[1], out[2:]
