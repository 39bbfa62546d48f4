"""Python model of the reference's brute-force block loop
(MergeTreeVSManager::vectorScanWithoutIndex, src/VectorIndex/Storages/MergeTreeVSManager.cpp:959-1535)
parameterised by the `search_wrapper` implementation, so the same driver replays the
goldens against the oracle (CPU tests) and against the product's host layer (GPU tests)."""
import numpy as np

FLT_MAX = np.finfo(np.float32).max
FLT_MIN = np.finfo(np.float32).tiny  # numeric_limits<float>::min(), the IP sentinel quirk (:1030-1033)

METRICS = {"L2": 0, "IP": 1, "Cosine": 2}


def brute_force_part(search_wrapper, vecs, empty, granularity, queries, k, metric, filt=None, row_exists=None):
    """vecs f32[n,d] with empty rows already FLT_MAX-padded; returns (final_id [nq,k], final_distance [nq,k])
    with part-local row offsets.  filt (bool[n] or None) = prewhere/where bitmap which already includes
    lightweight deletes; row_exists (bool[n] or None) = LWD mask for the no-filter path."""
    n, d = vecs.shape
    queries = np.ascontiguousarray(queries, np.float32).reshape(-1, d)
    nq = queries.shape[0]
    m = METRICS[metric]
    final_distance = np.full(nq * k, FLT_MIN if metric == "IP" else FLT_MAX, np.float32)
    final_id = np.full(nq * k, -1, np.int64)
    if filt is not None:
        # filter path (:1042-1330): one searchWrapper call per mark over the compacted passing, non-empty rows
        for lo in range(0, n, granularity):
            hi = min(n, lo + granularity)
            sel = [i for i in range(lo, hi) if filt[i] and not empty[i]]
            if hi - lo == 0:
                continue
            base = vecs[sel] if sel else np.zeros((0, d), np.float32)
            search_wrapper(queries, base, k, m, final_id, final_distance, num_rows_read=0,
                           actual_id_in_range=np.array(sel, np.uint64), row_exists=None, delete_id_num=0)
    else:
        # no-filter path (:1335-1497): dense block incl. FLT_MAX rows; LWD rows over-fetched then dropped
        read = 0
        while read < n:
            hi = min(n, read + granularity)
            base = vecs[read:hi]
            ex = None
            deleted = 0
            if row_exists is not None:
                ex = row_exists[read:hi]
                deleted = int((~ex).sum())
            search_wrapper(queries, base, k, m, final_id, final_distance, num_rows_read=read,
                           actual_id_in_range=None, row_exists=ex, delete_id_num=deleted)
            read = hi
    return final_id.reshape(nq, k), final_distance.reshape(nq, k)
