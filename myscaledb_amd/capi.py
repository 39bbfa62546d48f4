"""ctypes binding of libmsvs.so (include/msvs.h).  Plumbing only -- no compute happens here.

Fails loudly when the library is missing: there is deliberately no fallback path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmsvs.so")

METRIC_L2, METRIC_IP, METRIC_COSINE, METRIC_HAMMING, METRIC_JACCARD = 0, 1, 2, 3, 4
INDEX_FLAT, INDEX_IVFFLAT = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
MAX_K = 256
METRICS = {"L2": METRIC_L2, "IP": METRIC_IP, "Cosine": METRIC_COSINE, "COSINE": METRIC_COSINE, "cosine": METRIC_COSINE}

OK, ERR_INVALID_ARGUMENT, ERR_NOT_IMPLEMENTED, ERR_DEVICE, ERR_OOM, ERR_NOT_READY, ERR_UNSUPPORTED_K, ERR_ID_RANGE, \
    ERR_IO = range(9)

SYMBOLS = [
    "msvs_last_error", "msvs_version", "msvs_device_count", "msvs_set_device", "msvs_device_synchronize",
    "msvs_knn_f32", "msvs_normalize_f32", "msvs_index_create", "msvs_index_free", "msvs_index_train",
    "msvs_index_set_centroids", "msvs_index_set_cancel", "msvs_index_add", "msvs_index_build", "msvs_index_ready", "msvs_index_num_data",
    "msvs_index_num_lists", "msvs_index_memory_usage", "msvs_index_search", "msvs_index_search_device",
    "msvs_index_export", "msvs_index_export_list", "msvs_index_list_stats", "msvs_index_serialize", "msvs_index_load", "msvs_merge_topk", "msvs_merge_topk_device",
    "msvs_postings_create", "msvs_postings_create_fields", "msvs_postings_set_alive", "msvs_postings_free",
    "msvs_bm25_search", "msvs_bm25_search_batch", "msvs_bm25_search_batch_device", "msvs_bm25_stats",
    "msvs_release_scratch", "msvs_filter_from_bits", "msvs_filter_from_offsets", "msvs_filter_from_predicate", "msvs_filter_combine", "msvs_filter_count",
    "msvs_filter_to_bits", "msvs_filter_free", "msvs_index_search_filter", "msvs_index_search_filter_device", "msvs_index_scanned_rows",
    "msvs_profile_enable", "msvs_profile_get", "msvs_profile_reset", "msvs_merge_topk_device_strided",
    "msvs_knn_f32_filtered", "msvs_prefilter_stats", "msvs_coarse_stats", "msvs_combine_stats", "msvs_set_option", "msvs_index_serialize_io",
    "msvs_index_load_io", "msvs_index_version", "msvs_index_resource_usage", "msvs_knn_bin",
    "msvs_cache_create", "msvs_cache_free", "msvs_block_upload", "msvs_block_lookup", "msvs_block_release", "msvs_block_info", "msvs_bin_index_create", "msvs_bin_index_free",
    "msvs_bin_index_add", "msvs_bin_index_num_data", "msvs_bin_index_search", "msvs_bin_index_serialize_io", "msvs_bin_index_load_io",
    "msvs_cache_evict", "msvs_cache_stats", "msvs_knn_resident", "msvs_index_set_delete_bitmap",
    "msvs_index_set_merged_maps", "msvs_comm_unique_id", "msvs_comm_init", "msvs_comm_init_custom",
    "msvs_comm_free", "msvs_comm_all_reduce_u64", "msvs_comm_rank", "msvs_comm_size", "msvs_shard_search_device", "msvs_shard_search_device_async", "msvs_shard_search_drain", "msvs_shard_search_routed_device",
    "msvs_shard_search_routed_filtered_device", "msvs_shard_search_routed_device_async",
    "msvs_hybrid_fuse_device",
]


class MsvsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("msvs error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libmsvs.so is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "-- there is no CPU fallback" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.msvs_last_error.restype = C.c_char_p
        _lib.msvs_version.restype = C.c_char_p
        for n in ("msvs_index_num_data", "msvs_index_num_lists", "msvs_index_memory_usage"):
            getattr(_lib, n).restype = C.c_size_t
            getattr(_lib, n).argtypes = [C.c_void_p]
        _lib.msvs_index_free.argtypes = [C.c_void_p]
        _lib.msvs_index_free.restype = None
        _lib.msvs_postings_free.argtypes = [C.c_void_p]
        _lib.msvs_postings_free.restype = None
        _lib.msvs_index_ready.argtypes = [C.c_void_p]
    return _lib


def _check(rc):
    if rc != 0:
        raise MsvsError(rc, lib().msvs_last_error().decode())


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def pack_bits(mask):
    mask = np.asarray(mask, dtype=bool)
    n = mask.size
    padded = np.zeros(((n + 63) // 64) * 64, dtype=bool)
    padded[:n] = mask
    return np.packbits(padded.reshape(-1, 8), axis=1, bitorder="little").reshape(-1).view(np.uint64).copy()


def version():
    return lib().msvs_version().decode()


def device_count():
    n = C.c_int(0)
    _check(lib().msvs_device_count(C.byref(n)))
    return n.value


def set_device(i):
    _check(lib().msvs_set_device(int(i)))


def synchronize():
    _check(lib().msvs_device_synchronize())


def profile_enable(on):
    _check(lib().msvs_profile_enable(int(bool(on))))


def profile_reset():
    _check(lib().msvs_profile_reset())


def profile_get(name):
    """-> (calls, total_ms) of the kernel family `name` since the last reset (synchronises)."""
    c, t = C.c_uint64(0), C.c_double(0)
    _check(lib().msvs_profile_get(name.encode(), C.byref(c), C.byref(t)))
    return c.value, t.value


def set_option(name, value=None):
    """Experiment / test knob (msvs_set_option): value None restores the default."""
    _check(lib().msvs_set_option(name.encode(), None if value is None else str(value).encode()))


def prefilter_stats():
    """-> (queries, fallbacks) of the matrix-core candidate pass on the current device (msvs_prefilter_stats)."""
    q, f = C.c_uint64(0), C.c_uint64(0)
    _check(lib().msvs_prefilter_stats(C.byref(q), C.byref(f)))
    return q.value, f.value


def combine_stats():
    """-> (calls, batches, batched queries) of msvs_index_search's combining front end (msvs_combine_stats)."""
    a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    _check(lib().msvs_combine_stats(C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def coarse_stats():
    """-> (queries, fallbacks) of the coarse quantiser's candidate passes on the current device (msvs_coarse_stats)."""
    q, f = C.c_uint64(0), C.c_uint64(0)
    _check(lib().msvs_coarse_stats(C.byref(q), C.byref(f)))
    return q.value, f.value


def knn(x, y, k, metric, alive=None):
    """msvs_knn_f32 / msvs_knn_f32_filtered (seam A2).  x [nx,d], y [ny,d] host arrays, alive bool[ny] or None
    -> (ids int64 [nx,k], dis f32 [nx,k])."""
    y = _f32(y)
    d = y.shape[1]
    x = _f32(x).reshape(-1, d)
    nx = x.shape[0]
    ids = np.empty((nx, k), np.int64)
    dis = np.empty((nx, k), np.float32)
    if alive is None:
        _check(lib().msvs_knn_f32(_p(x, C.c_float), _p(y, C.c_float), C.c_size_t(d), C.c_size_t(k), C.c_size_t(nx),
                                  C.c_size_t(y.shape[0]), int(metric), _p(ids, C.c_int64), _p(dis, C.c_float)))
    else:
        bits = pack_bits(alive)
        _check(lib().msvs_knn_f32_filtered(_p(x, C.c_float), _p(y, C.c_float), C.c_size_t(d), C.c_size_t(k),
                                           C.c_size_t(nx), C.c_size_t(y.shape[0]), int(metric), _p(bits, C.c_uint64),
                                           _p(ids, C.c_int64), _p(dis, C.c_float)))
    return ids, dis


def knn_bin(x, y, k, metric, alive=None):
    """msvs_knn_bin: x [nx, nbytes], y [ny, nbytes] uint8 host arrays -> (ids int64 [nx, k], dis f32 [nx, k])."""
    y = np.ascontiguousarray(y, np.uint8)
    nb = y.shape[1]
    x = np.ascontiguousarray(x, np.uint8).reshape(-1, nb)
    nx = x.shape[0]
    ids = np.empty((nx, k), np.int64)
    dis = np.empty((nx, k), np.float32)
    bits = None if alive is None else pack_bits(alive)
    _check(lib().msvs_knn_bin(_p(x, C.c_uint8), _p(y, C.c_uint8), C.c_size_t(nb), C.c_size_t(k), C.c_size_t(nx),
                              C.c_size_t(y.shape[0]), int(metric), _p(bits, C.c_uint64), _p(ids, C.c_int64),
                              _p(dis, C.c_float)))
    return ids, dis


class BinIndex:
    """msvs_bin_index_t: BinaryFLAT over rows resident on the device (labels = ids given at add, filter indexed by label)."""

    def __init__(self, nbytes, metric):
        self._h = C.c_void_p()
        self.nbytes = int(nbytes)
        _check(lib().msvs_bin_index_create(C.c_size_t(nbytes), int(metric), C.byref(self._h)))

    def add(self, rows, ids=None):
        rows = np.ascontiguousarray(rows, np.uint8).reshape(-1, self.nbytes)
        idp = None
        if ids is not None:
            ids = np.ascontiguousarray(ids, np.int64)
            idp = _p(ids, C.c_int64)
        _check(lib().msvs_bin_index_add(self._h, _p(rows, C.c_uint8), idp, C.c_size_t(rows.shape[0])))

    @property
    def num_data(self):
        lib().msvs_bin_index_num_data.restype = C.c_size_t
        return lib().msvs_bin_index_num_data(self._h)

    def search(self, x, k, alive=None):
        x = np.ascontiguousarray(x, np.uint8).reshape(-1, self.nbytes)
        ids = np.empty((x.shape[0], k), np.int64)
        dis = np.empty((x.shape[0], k), np.float32)
        bits = None if alive is None else pack_bits(alive)
        _check(lib().msvs_bin_index_search(self._h, _p(x, C.c_uint8), C.c_size_t(x.shape[0]), C.c_size_t(k), _p(bits, C.c_uint64),
                                           C.c_size_t(0 if alive is None else len(alive)), _p(ids, C.c_int64), _p(dis, C.c_float)))
        return ids, dis

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.msvs_bin_index_free.argtypes = [C.c_void_p]
            _lib.msvs_bin_index_free.restype = None
            _lib.msvs_bin_index_free(self._h)
            self._h = None

    __del__ = close


def normalize(x):
    x = _f32(x).copy()
    x2 = x.reshape(-1, x.shape[-1])
    _check(lib().msvs_normalize_f32(_p(x2, C.c_float), C.c_size_t(x2.shape[0]), C.c_size_t(x2.shape[1])))
    return x


def merge_topk(ids, dis, metric):
    """ids/dis [nparts, nq, k] host arrays -> merged (ids [nq,k], dis [nq,k])."""
    ids = np.ascontiguousarray(ids, np.int64)
    dis = _f32(dis)
    nparts, nq, k = ids.shape
    oi, od = np.empty((nq, k), np.int64), np.empty((nq, k), np.float32)
    _check(lib().msvs_merge_topk(_p(ids, C.c_int64), _p(dis, C.c_float), C.c_size_t(nparts), C.c_size_t(nq),
                                 C.c_size_t(k), int(metric), _p(oi, C.c_int64), _p(od, C.c_float)))
    return oi, od


class Cache:
    """msvs_cache_t: LRU of resident brute-force blocks (seam A2)."""

    def __init__(self, capacity_bytes):
        h = C.c_void_p()
        _check(lib().msvs_cache_create(C.c_size_t(capacity_bytes), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.msvs_cache_free.argtypes = [C.c_void_p]
            _lib.msvs_cache_free.restype = None
            _lib.msvs_cache_free(self._h)
            self._h = None

    __del__ = close

    def upload(self, key, mark, rows, normalize=False):
        rows = _f32(rows)
        b = C.c_void_p()
        _check(lib().msvs_block_upload(self._h, key.encode(), C.c_uint64(mark), _p(rows, C.c_float),
                                       C.c_size_t(rows.shape[0]), C.c_size_t(rows.shape[1]), int(normalize), C.byref(b)))
        return b

    def lookup(self, key, mark):
        b = C.c_void_p()
        _check(lib().msvs_block_lookup(self._h, key.encode(), C.c_uint64(mark), C.byref(b)))
        return b if b.value else None

    @staticmethod
    def release(block):
        lib().msvs_block_release.argtypes = [C.c_void_p]
        lib().msvs_block_release.restype = None
        lib().msvs_block_release(block)

    def evict(self, prefix):
        n = C.c_size_t(0)
        _check(lib().msvs_cache_evict(self._h, prefix.encode(), C.byref(n)))
        return n.value

    def stats(self):
        """-> dict(bytes, blocks, hits, misses, evictions)"""
        by, bl = C.c_size_t(0), C.c_size_t(0)
        h, m, e = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _check(lib().msvs_cache_stats(self._h, C.byref(by), C.byref(bl), C.byref(h), C.byref(m), C.byref(e)))
        return dict(bytes=by.value, blocks=bl.value, hits=h.value, misses=m.value, evictions=e.value)


def knn_resident(block, x, k, metric, d, alive=None):
    x = _f32(x).reshape(-1, d)
    nx = x.shape[0]
    ids = np.empty((nx, k), np.int64)
    dis = np.empty((nx, k), np.float32)
    bits = None if alive is None else pack_bits(alive)
    _check(lib().msvs_knn_resident(block, _p(x, C.c_float), C.c_size_t(k), C.c_size_t(nx), int(metric),
                                   _p(bits, C.c_uint64), _p(ids, C.c_int64), _p(dis, C.c_float)))
    return ids, dis


COMM_ID_BYTES = 128
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


def comm_unique_id():
    """msvs_comm_unique_id (rank 0): the bytes every rank needs for Comm(id=...)."""
    buf = (C.c_ubyte * COMM_ID_BYTES)()
    _check(lib().msvs_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """msvs_comm_t: RCCL communicator owned by libmsvs (id = bytes from comm_unique_id()), or a caller-supplied
    all-gather (all_gather = python callable(d_send, d_recv, nbytes, stream) -> 0) for tests."""

    def __init__(self, nranks, rank, id=None, all_gather=None):
        h = C.c_void_p()
        self._cb = None
        if all_gather is not None:
            self._cb = ALLGATHER_FN(lambda ctx, s, r, n, st: int(all_gather(s, r, n, st) or 0))
            _check(lib().msvs_comm_init_custom(int(nranks), int(rank), self._cb, None, C.byref(h)))
        else:
            buf = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(id) if id is not None else None
            _check(lib().msvs_comm_init(buf, int(nranks), int(rank), C.byref(h)))
        self._h = h
        self.nranks, self.rank = nranks, rank

    def drain(self, stream=0):
        """msvs_shard_search_drain: `stream` waits for every batch of msvs_shard_search_device_async still in flight."""
        _check(lib().msvs_shard_search_drain(self._h, C.c_void_p(int(stream)) if stream else None))

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.msvs_comm_free.argtypes = [C.c_void_p]
            _lib.msvs_comm_free.restype = None
            _lib.msvs_comm_free(self._h)
            self._h = None

    __del__ = close


class _MsvsIO(C.Structure):
    _fields_ = [("ctx", C.c_void_p),
                ("open", C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_char_p, C.c_int)),
                ("write", C.CFUNCTYPE(C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)),
                ("read", C.CFUNCTYPE(C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)),
                ("close", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p))]


class _DictIO:
    """msvs_io_t over a dict NAME -> bytearray: stands in for the host's IDisk-backed streams in tests."""

    def __init__(self, store):
        self.store, self.streams, self.next = store, {}, 1

        def op(_ctx, name, write):
            name = name.decode()
            if write:
                self.store[name] = bytearray()
            elif name not in self.store:
                return None
            h = self.next
            self.next += 1
            self.streams[h] = [name, 0]
            return h

        def wr(_ctx, h, buf, n):
            self.store[self.streams[h][0]] += C.string_at(buf, n)
            return n

        def rd(_ctx, h, buf, n):
            name, pos = self.streams[h]
            data = self.store[name][pos:pos + n]
            C.memmove(buf, bytes(data), len(data))
            self.streams[h][1] = pos + len(data)
            return len(data)

        def cl(_ctx, h):
            self.streams.pop(h, None)
            return 0

        t = _MsvsIO
        self._keep = (t._fields_[1][1](op), t._fields_[2][1](wr), t._fields_[3][1](rd), t._fields_[4][1](cl))
        self.io = _MsvsIO(None, *self._keep)


def index_version():
    lib().msvs_index_version.restype = C.c_char_p
    return lib().msvs_index_version().decode()


class Index:
    """msvs_index_t (seam A1)."""

    def __init__(self, index_type, metric, dim, params="", _handle=None):
        self.dim = int(dim)
        self.metric = metric
        self.index_type = index_type
        if _handle is not None:
            self._h = _handle
            return
        h = C.c_void_p()
        _check(lib().msvs_index_create(int(index_type), int(metric), C.c_size_t(dim), params.encode(), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.msvs_index_free(self._h)
            self._h = None

    __del__ = close

    @staticmethod
    def _ptr(x, mem):
        if mem == MEM_DEVICE:
            return C.c_void_p(int(x))
        return x.ctypes.data_as(C.c_void_p)

    def train(self, x, n=None, mem=MEM_HOST):
        if mem == MEM_HOST:
            x = _f32(x)
            n = x.shape[0]
        _check(lib().msvs_index_train(self._h, self._ptr(x, mem), C.c_size_t(n), mem))

    def set_centroids(self, c):
        c = _f32(c)
        _check(lib().msvs_index_set_centroids(self._h, self._ptr(c, MEM_HOST), C.c_size_t(c.shape[0]), MEM_HOST))

    def add(self, x, ids=None, n=None, mem=MEM_HOST):
        if mem == MEM_HOST:
            x = _f32(x)
            n = x.shape[0]
            idp = None
            if ids is not None:
                ids = np.ascontiguousarray(ids, np.int64)
                idp = ids.ctypes.data_as(C.c_void_p)
        else:
            idp = None if ids is None else C.c_void_p(int(ids))
        _check(lib().msvs_index_add(self._h, self._ptr(x, mem), idp, C.c_size_t(n), mem))

    def build(self):
        _check(lib().msvs_index_build(self._h))

    def export_list(self, l, length):
        """(vecs [length, dim] f32, ids [length] i64) of inverted list l (length from export(with_vecs=False)'s offsets)."""
        vecs = np.empty((length, self.dim), np.float32)
        ids = np.empty(length, np.int64)
        _check(lib().msvs_index_export_list(self._h, C.c_size_t(l), _p(vecs, C.c_float), _p(ids, C.c_int64)))
        return vecs, ids

    def list_stats(self):
        """{nlist, min_len, max_len, imbalance (nlist * sum(len^2) / n^2), train_empty} of a built IVFFLAT index."""
        nl, mn, mx, te = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        imb = C.c_double()
        _check(lib().msvs_index_list_stats(self._h, C.byref(nl), C.byref(mn), C.byref(mx), C.byref(imb), C.byref(te)))
        return {"nlist": nl.value, "min_len": mn.value, "max_len": mx.value, "imbalance": imb.value, "train_empty": te.value}

    @property
    def ready(self):
        return bool(lib().msvs_index_ready(self._h))

    @property
    def num_data(self):
        return lib().msvs_index_num_data(self._h)

    @property
    def num_lists(self):
        return lib().msvs_index_num_lists(self._h)

    @property
    def memory_usage(self):
        return lib().msvs_index_memory_usage(self._h)

    _search_fn = None  # msvs_index_search with integer-address argtypes (the unfiltered fast path below)

    def search(self, queries, k, params="", alive=None):
        q = _f32(queries).reshape(-1, self.dim)
        nq = q.shape[0]
        if alive is None:
            # host-pointer call without the per-argument ctypes pointer objects (1.8 us each: a third of the wrapper's overhead on a
            # 40 us single-query call): one result block, raw addresses
            fn = Index._search_fn
            if fn is None:
                fn = lib()["msvs_index_search"]
                fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
                fn.restype = C.c_int
                Index._search_fn = fn
            out = np.empty(nq * k * 12, np.uint8)
            base = C.addressof(C.c_char.from_buffer(out))
            try:
                qaddr = C.addressof(C.c_char.from_buffer(q))  # (0.3 us; .ctypes.data builds an interface object: 0.9 us)
            except (TypeError, ValueError):  # read-only or empty array
                qaddr = q.ctypes.data
            rc = fn(self._h, qaddr, nq, k, params.encode(), None, 0, base, base + nq * k * 8)
            if rc:
                _check(rc)
            return np.ndarray((nq, k), np.int64, out, 0), np.ndarray((nq, k), np.float32, out, nq * k * 8)
        ids = np.empty((nq, k), np.int64)
        dis = np.empty((nq, k), np.float32)
        bits, nbits = None, 0
        if alive is not None:
            nbits = len(alive)
            bits = pack_bits(alive)
        _check(lib().msvs_index_search(self._h, _p(q, C.c_float), C.c_size_t(nq), int(k), params.encode(),
                                       _p(bits, C.c_uint64), C.c_size_t(nbits), _p(ids, C.c_int64), _p(dis, C.c_float)))
        return ids, dis

    def search_filter(self, queries, k, params, flt):
        """msvs_index_search_filter: the library picks bit test or compacted view from the filter's population count."""
        q = _f32(queries).reshape(-1, self.dim)
        nq = q.shape[0]
        ids = np.empty((nq, k), np.int64)
        dis = np.empty((nq, k), np.float32)
        _check(lib().msvs_index_search_filter(self._h, _p(q, C.c_float), C.c_size_t(nq), int(k), params.encode(), flt._h,
                                              _p(ids, C.c_int64), _p(dis, C.c_float)))
        return ids, dis

    def search_filter_device(self, d_queries, nq, k, nprobe, flt, d_ids, d_dis, stream=0):
        _check(lib().msvs_index_search_filter_device(self._h, C.c_void_p(int(d_queries)), C.c_size_t(nq), int(k), int(nprobe), flt._h,
                                                     C.c_void_p(int(d_ids)), C.c_void_p(int(d_dis)),
                                                     C.c_void_p(int(stream)) if stream else None))

    def search_device(self, d_queries, nq, k, nprobe, d_ids, d_dis, stream=0, d_alive=0, nbits=0):
        """All arguments are raw device addresses (ints); enqueues on `stream` and returns immediately."""
        _check(lib().msvs_index_search_device(self._h, C.c_void_p(int(d_queries)), C.c_size_t(nq), int(k), int(nprobe),
                                              C.c_void_p(int(d_alive)) if d_alive else None, C.c_size_t(nbits),
                                              C.c_void_p(int(d_ids)), C.c_void_p(int(d_dis)),
                                              C.c_void_p(int(stream)) if stream else None))

    def shard_search_device(self, comm, d_queries, nq, k, nprobe, d_ids, d_dis, stream=0, d_alive=0, nbits=0):
        """msvs_shard_search_device: the whole sharded search (coarse by query + probe all-gather + local scan + packed
        all-gather + merge) on `stream`; raw device addresses."""
        _check(lib().msvs_shard_search_device(self._h, comm._h, C.c_void_p(int(d_queries)), C.c_size_t(nq), int(k),
                                              int(nprobe), C.c_void_p(int(d_alive)) if d_alive else None,
                                              C.c_size_t(nbits), C.c_void_p(int(d_ids)), C.c_void_p(int(d_dis)),
                                              C.c_void_p(int(stream)) if stream else None))

    def shard_search_routed_device(self, comm, d_queries, nq, k, nprobe, d_ids, d_dis, stream=0, d_alive=0, nbits=0):
        """msvs_shard_search_routed[_filtered]_device: this rank's OWN nq queries (0 allowed), routed to the ranks that own lists they
        still need after the pre-pruning, searched there under that rank's filter (d_alive: THIS rank's bitmap); collective.  Returns
        the (query, rank) pairs this rank served."""
        served = C.c_uint64(0)
        q = C.c_void_p(int(d_queries)) if d_queries else None
        oi, od = C.c_void_p(int(d_ids)) if d_ids else None, C.c_void_p(int(d_dis)) if d_dis else None
        st = C.c_void_p(int(stream)) if stream else None
        if d_alive:
            _check(lib().msvs_shard_search_routed_filtered_device(self._h, comm._h, q, C.c_size_t(nq), int(k), int(nprobe), C.c_void_p(int(d_alive)),
                                                                  C.c_size_t(nbits), oi, od, st, C.byref(served)))
        else:
            _check(lib().msvs_shard_search_routed_device(self._h, comm._h, q, C.c_size_t(nq), int(k), int(nprobe), oi, od, st, C.byref(served)))
        return served.value

    def shard_search_routed_device_async(self, comm, d_queries, nq, k, nprobe, d_ids, d_dis, stream=0, d_alive=0, nbits=0, served=None,
                                         want_event=True):
        """msvs_shard_search_routed_device_async: two routed steps in flight.  Returns the done event of the PREVIOUS call's batch (None at
        the first call; never asked for with want_event=False: a caller that stays on one stream and drains at the end needs none, and an
        event record between two kernels is ~8 us of idle device); `served` (a ctypes c_uint64 the caller keeps alive) receives this
        batch's routed pairs when ITS back phase runs."""
        ev = C.c_void_p()
        _check(lib().msvs_shard_search_routed_device_async(self._h, comm._h, C.c_void_p(int(d_queries)) if d_queries else None, C.c_size_t(nq), int(k),
                                                           int(nprobe), C.c_void_p(int(d_alive)) if d_alive else None, C.c_size_t(nbits),
                                                           C.c_void_p(int(d_ids)) if d_ids else None, C.c_void_p(int(d_dis)) if d_dis else None,
                                                           C.c_void_p(int(stream)) if stream else None,
                                                           C.byref(served) if served is not None else None, C.byref(ev) if want_event else None))
        return ev.value

    def shard_search_device_async(self, comm, d_queries, nq, k, nprobe, d_ids, d_dis, stream=0, d_alive=0, nbits=0):
        """msvs_shard_search_device_async: two batches in flight; returns the batch's done event (a hipEvent_t address)."""
        ev = C.c_void_p()
        _check(lib().msvs_shard_search_device_async(self._h, comm._h, C.c_void_p(int(d_queries)), C.c_size_t(nq), int(k),
                                                    int(nprobe), C.c_void_p(int(d_alive)) if d_alive else None,
                                                    C.c_size_t(nbits), C.c_void_p(int(d_ids)), C.c_void_p(int(d_dis)),
                                                    C.c_void_p(int(stream)) if stream else None, C.byref(ev)))
        return ev.value

    def scanned_rows(self, queries, nprobe):
        q = _f32(queries).reshape(-1, self.dim)
        rows, streamed, unique = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _check(lib().msvs_index_scanned_rows(self._h, _p(q, C.c_float), C.c_size_t(q.shape[0]), int(nprobe),
                                             C.byref(rows), C.byref(streamed), C.byref(unique)))
        return rows.value, streamed.value, unique.value

    def export(self, with_vecs=True):
        """(centroids, list offsets, rows in storage order, ids in storage order); with_vecs=False leaves the rows on the
        device (a 10M x 768 index is 30 GB) and returns None for them."""
        n, nl, d = self.num_data, self.num_lists, self.dim
        cent = np.empty((nl, d), np.float32) if self.index_type == INDEX_IVFFLAT else None
        off = np.empty(nl + 1, np.int64)
        vecs = np.empty((n, d), np.float32) if with_vecs else None
        ids = np.empty(n, np.int64)
        _check(lib().msvs_index_export(self._h, _p(cent, C.c_float), _p(off, C.c_int64), _p(vecs, C.c_float),
                                       _p(ids, C.c_int64)))
        return cent, off, vecs, ids

    def serialize(self, path_prefix):
        """-> files <path_prefix>-data_bin.vidx3, <path_prefix>-id_list.vidx3 (stdio convenience)."""
        _check(lib().msvs_index_serialize(self._h, path_prefix.encode()))

    @classmethod
    def load(cls, path_prefix, index_type, metric, dim):
        h = C.c_void_p()
        _check(lib().msvs_index_load(path_prefix.encode(), C.byref(h)))
        return cls(index_type, metric, dim, _handle=h)

    def serialize_io(self, store):
        """msvs_index_serialize_io through stream callbacks; `store` = dict NAME -> bytearray (filled in)."""
        _check(lib().msvs_index_serialize_io(self._h, C.byref(_DictIO(store).io)))

    @classmethod
    def load_io(cls, store, index_type, metric, dim):
        h = C.c_void_p()
        _check(lib().msvs_index_load_io(C.byref(_DictIO(store).io), C.byref(h)))
        return cls(index_type, metric, dim, _handle=h)

    def set_delete_bitmap(self, alive):
        """VIWithMeta::setDeleteBitmap: bool[n] over the index labels (1 = alive), resident; None clears."""
        if alive is None:
            _check(lib().msvs_index_set_delete_bitmap(self._h, None, C.c_size_t(0)))
        else:
            bits = pack_bits(alive)
            _check(lib().msvs_index_set_delete_bitmap(self._h, _p(bits, C.c_uint64), C.c_size_t(len(alive))))

    def set_merged_maps(self, row_ids_map, inverted_row_ids_map, inverted_row_sources_map, own_id):
        a = np.ascontiguousarray(row_ids_map, np.uint64)
        b = np.ascontiguousarray(inverted_row_ids_map, np.uint64)
        c = np.ascontiguousarray(inverted_row_sources_map, np.uint8)
        _check(lib().msvs_index_set_merged_maps(self._h, _p(a, C.c_uint64), C.c_size_t(a.size), _p(b, C.c_uint64),
                                                _p(c, C.c_uint8), C.c_size_t(b.size), C.c_uint32(own_id)))

    def resource_usage(self):
        """-> (memory_usage_bytes, disk_usage_bytes, build_memory_usage_bytes)"""
        m, dk, b = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        _check(lib().msvs_index_resource_usage(self._h, C.byref(m), C.byref(dk), C.byref(b)))
        return m.value, dk.value, b.value


class _Scalar(C.Structure):
    _fields_ = [("i", C.c_int64), ("f", C.c_double)]


_DTYPES = {np.dtype(np.uint8): 0, np.dtype(np.uint16): 1, np.dtype(np.uint32): 2, np.dtype(np.uint64): 3, np.dtype(np.int8): 4,
           np.dtype(np.int16): 5, np.dtype(np.int32): 6, np.dtype(np.int64): 7, np.dtype(np.float32): 8, np.dtype(np.float64): 9}
OPS = {"==": 0, "!=": 1, "<": 2, "<=": 3, ">": 4, ">=": 5, "between": 6}
FILTER_AND, FILTER_OR, FILTER_AND_NOT = 0, 1, 2


class Filter:
    """msvs_filter_t: a search filter (one bit per row offset of the part) resident on the device."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_bool(cls, alive):
        bits = pack_bits(alive)
        h = C.c_void_p()
        _check(lib().msvs_filter_from_bits(_p(bits, C.c_uint64), C.c_size_t(len(alive)), C.byref(h)))
        return cls(h)

    @classmethod
    def from_offsets(cls, offsets, nbits):
        """getFilterFromPipeline: the passing `_part_offset`s."""
        off = np.ascontiguousarray(offsets, np.uint64)
        h = C.c_void_p()
        _check(lib().msvs_filter_from_offsets(_p(off, C.c_uint64), C.c_size_t(off.size), C.c_size_t(nbits), MEM_HOST, C.byref(h)))
        return cls(h)

    @classmethod
    def from_predicate(cls, column, op, lo, hi=0, device_ptr=None):
        """`column OP lo` (OP in OPS; "between": lo <= x <= hi).  column: numpy array (uploaded), or pass its dtype-carrying
        empty view plus device_ptr / len via (dtype, nrows) tuple for a column already on the device."""
        if device_ptr is not None:
            dt, n = column
            ptr, mem = C.c_void_p(int(device_ptr)), MEM_DEVICE
        else:
            col = np.ascontiguousarray(column)
            dt, n = col.dtype, col.size
            ptr, mem = col.ctypes.data_as(C.c_void_p), MEM_HOST
        isf = np.dtype(dt).kind == "f"
        sl = _Scalar(0 if isf else int(lo), float(lo) if isf else 0.0)
        sh = _Scalar(0 if isf else int(hi), float(hi) if isf else 0.0)
        h = C.c_void_p()
        lib().msvs_filter_from_predicate.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int, _Scalar, _Scalar,
                                                     C.POINTER(C.c_void_p)]
        _check(lib().msvs_filter_from_predicate(ptr, _DTYPES[np.dtype(dt)], n, mem, OPS[op], sl, sh, C.byref(h)))
        return cls(h)

    def combine(self, other, mode=FILTER_AND):
        _check(lib().msvs_filter_combine(self._h, other._h, int(mode)))
        return self

    def count(self):
        a, n = C.c_uint64(0), C.c_size_t(0)
        _check(lib().msvs_filter_count(self._h, C.byref(a), C.byref(n)))
        return a.value, n.value

    def to_bool(self):
        alive, nbits = self.count()
        words = np.zeros(max(1, (nbits + 63) // 64), np.uint64)
        _check(lib().msvs_filter_to_bits(self._h, _p(words, C.c_uint64)))
        return np.unpackbits(words.view(np.uint8), bitorder="little")[:nbits].astype(bool)

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.msvs_filter_free.argtypes = [C.c_void_p]
            _lib.msvs_filter_free.restype = None
            _lib.msvs_filter_free(self._h)
            self._h = None

    __del__ = close


def release_scratch():
    f = C.c_size_t(0)
    _check(lib().msvs_release_scratch(C.byref(f)))
    return f.value


def hybrid_fuse_device(fusion, d_vec_dis, d_vec_ids, kv, d_txt_scores, d_txt_ids, kt, nq, topk, d_out_scores, d_out_labels, d_n_out,
                       stream=0, fusion_k=60, fusion_weight=0.5, vector_scan_direction=1):
    """msvs_hybrid_fuse_device: RRF ("rrf") / RSF ("rsf") fusion of a batch's vector and text result lists, all device
    pointers (ints), stream-ordered.  Outputs [nq][topk] f32 scores, i64 labels (-1 past n_out[q]), u32 n_out."""
    _check(lib().msvs_hybrid_fuse_device(C.c_int(1 if fusion == "rsf" else 0), C.c_void_p(d_vec_dis), C.c_void_p(d_vec_ids),
                                         C.c_size_t(kv), C.c_void_p(d_txt_scores), C.c_void_p(d_txt_ids), C.c_size_t(kt),
                                         C.c_size_t(nq), C.c_uint64(fusion_k), C.c_float(fusion_weight), C.c_int(vector_scan_direction),
                                         C.c_size_t(topk), C.c_void_p(d_out_scores), C.c_void_p(d_out_labels), C.c_void_p(d_n_out),
                                         C.c_void_p(stream)))


def debug_prune_stats():
    """(pairs the probe pruning of the shadow list scan dropped, pairs it looked at) -- counted only under rerank_stats = 1."""
    out = (C.c_uint64 * 2)()
    _check(lib().msvs_debug_prune_stats(out))
    return int(out[0]), int(out[1])


def debug_scan_rows():
    """(rows the shadow main launches read, rows their sample launches read), cumulative -- counted only under rerank_stats = 1."""
    out = (C.c_uint64 * 2)()
    _check(lib().msvs_debug_scan_rows(out))
    return int(out[0]), int(out[1])


def bm25_stats():
    """(queries through the sample / emit path, of which fallbacks)."""
    q, f = C.c_uint64(0), C.c_uint64(0)
    _check(lib().msvs_bm25_stats(C.byref(q), C.byref(f)))
    return q.value, f.value


class Postings:
    """msvs_postings_t (seam B): flat-array export of one part's inverted index.  fieldnorm_ids: [num_docs] (one text
    column) or [num_fields, num_docs] with term_field[t] = the column of term t."""

    def __init__(self, post_off, doc_ids, tfs, fieldnorm_ids, term_field=None):
        post_off = np.ascontiguousarray(post_off, np.int64)
        doc_ids = np.ascontiguousarray(doc_ids, np.uint32)
        tfs = np.ascontiguousarray(tfs, np.uint32)
        fieldnorm_ids = np.ascontiguousarray(np.atleast_2d(fieldnorm_ids), np.uint8)
        self.num_fields, self.num_docs = fieldnorm_ids.shape
        tf_ = None if term_field is None else np.ascontiguousarray(term_field, np.uint8)
        h = C.c_void_p()
        _check(lib().msvs_postings_create_fields(_p(post_off, C.c_int64), C.c_size_t(post_off.size - 1), _p(tf_, C.c_uint8),
                                                 _p(doc_ids, C.c_uint32), _p(tfs, C.c_uint32), _p(fieldnorm_ids, C.c_uint8),
                                                 C.c_size_t(self.num_fields), C.c_size_t(self.num_docs), C.byref(h)))
        self._h = h

    def set_alive(self, alive):
        """Resident lightweight-delete bitmap of the part (None clears)."""
        if alive is None:
            _check(lib().msvs_postings_set_alive(self._h, None, C.c_size_t(0)))
        else:
            bits = pack_bits(alive)
            _check(lib().msvs_postings_set_alive(self._h, _p(bits, C.c_uint64), C.c_size_t(len(alive))))

    def _batch_args(self, queries, dfs, groups, total_tokens):
        qoff = np.zeros(len(queries) + 1, np.uint32)
        qoff[1:] = np.cumsum([len(q) for q in queries])
        cat = lambda xs, dt: np.ascontiguousarray(np.concatenate([np.asarray(x, dt).ravel() for x in xs]) if len(xs) else [], dt)
        qterms, df = cat(queries, np.uint32), cat(dfs, np.uint64)
        qg = None if groups is None else cat(groups, np.uint32)
        tokens = np.ascontiguousarray(np.broadcast_to(np.asarray(total_tokens, np.uint64), (self.num_fields,)))
        return qoff, qterms, qg, df, tokens

    def bm25_search_batch(self, queries, dfs, total_docs, total_tokens, k, alive=None, groups=None, operator_or=True):
        """queries / dfs (/ groups): one sequence per query.  Returns lists of (rows, scores)."""
        nq = len(queries)
        qoff, qterms, qg, df, tokens = self._batch_args(queries, dfs, groups, total_tokens)
        bits, nbits = None, 0
        if alive is not None:
            nbits = len(alive)
            bits = pack_bits(alive)
        rows, scores, cnt = np.empty((nq, k), np.uint64), np.empty((nq, k), np.float32), np.zeros(nq, np.uint32)
        _check(lib().msvs_bm25_search_batch(self._h, C.c_size_t(nq), _p(qoff, C.c_uint32), _p(qterms, C.c_uint32),
                                            _p(qg, C.c_uint32), _p(df, C.c_uint64), C.c_uint64(int(total_docs)),
                                            _p(tokens, C.c_uint64), 1 if operator_or else 0, _p(bits, C.c_uint64),
                                            C.c_size_t(nbits), C.c_size_t(k), _p(rows, C.c_uint64), _p(scores, C.c_float),
                                            _p(cnt, C.c_uint32)))
        return [(rows[q, :cnt[q]].copy(), scores[q, :cnt[q]].copy()) for q in range(nq)]

    def prepare_batch(self, queries, dfs, total_tokens, groups=None):
        """The flat argument arrays of a batch, built once (a serving front end keeps them in this form anyway)."""
        return (len(queries),) + self._batch_args(queries, dfs, groups, total_tokens)

    def bm25_search_batch_device(self, queries, dfs, total_docs, total_tokens, k, d_row_ids, d_scores, stream=0,
                                 d_alive=0, nbits=0, groups=None, operator_or=True, prepared=None):
        """Stream-ordered: int64 ids (-1 = no hit) / f32 scores into DEVICE buffers [nq, k] given by address.
        prepared: the result of prepare_batch (queries / dfs / total_tokens / groups are then ignored)."""
        nq, qoff, qterms, qg, df, tokens = prepared if prepared is not None else self.prepare_batch(queries, dfs, total_tokens, groups)
        _check(lib().msvs_bm25_search_batch_device(self._h, C.c_size_t(nq), _p(qoff, C.c_uint32), _p(qterms, C.c_uint32),
                                                   _p(qg, C.c_uint32), _p(df, C.c_uint64), C.c_uint64(int(total_docs)),
                                                   _p(tokens, C.c_uint64), 1 if operator_or else 0, C.c_void_p(d_alive),
                                                   C.c_size_t(nbits), C.c_size_t(k), C.c_void_p(d_row_ids),
                                                   C.c_void_p(d_scores), C.c_void_p(stream)))

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.msvs_postings_free(self._h)
            self._h = None

    __del__ = close

    def bm25_search(self, qterms, df, total_docs, total_tokens, k, alive=None):
        qterms = np.ascontiguousarray(qterms, np.uint32)
        df = np.ascontiguousarray(df, np.uint64)
        bits, nbits = None, 0
        if alive is not None:
            nbits = len(alive)
            bits = pack_bits(alive)
        rows, scores = np.empty(k, np.uint64), np.empty(k, np.float32)
        n = C.c_size_t(0)
        _check(lib().msvs_bm25_search(self._h, _p(qterms, C.c_uint32), _p(df, C.c_uint64), C.c_size_t(qterms.size),
                                      C.c_uint64(int(total_docs)), C.c_uint64(int(total_tokens)), _p(bits, C.c_uint64),
                                      C.c_size_t(nbits), C.c_size_t(k), _p(rows, C.c_uint64), _p(scores, C.c_float),
                                      C.byref(n)))
        return rows[:n.value], scores[:n.value]
