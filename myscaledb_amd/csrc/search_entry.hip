// search_entry.hip -- the search entry points of an index (include/msvs.h: msvs_index_search, msvs_index_search_device,
// msvs_combine_stats): the few-query path (one or two queries per call in two self-merging launches, latency_kernels.hpp), the
// host-pointer call with its staging, parameter string and exact rounds for large k, and the front end that combines many
// concurrent single-query callers into batches.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "index_internal.hpp"
#include "latency_kernels.hpp"

using namespace msvs;

// ------------------------------------------------------------------------------------------ few-query path

namespace
{
/// Buffers of the two-launch search (latency_kernels.hpp) of one (host thread, stream): grow-only, reused call after call.
struct LatCtx
{
    DevBuf<float> dq;
    DevBuf<int32_t> probes;
    DevBuf<float> probe_dis;
    DevBuf<uint32_t> cut;
    DevBuf<uint64_t> c_partial, partial;
    DevBuf<uint32_t> done;
    // stage 1 as the coarse quantiser of a small batch (coarse_few_launch): its own lists and per-query arrival counters
    DevBuf<uint64_t> cf_partial;
    DevBuf<uint32_t> cf_done;
    DevBuf<float> cf_dis;
    // host side (pinned, device-visible): queries in, results + completion word out
    unsigned char * pinned = nullptr;
    size_t pinned_bytes = 0;
    uint32_t seq = 0;
    // ... and of a small batch (index_search_host_call): its own block -- the few-query paths keep a completion word in theirs
    unsigned char * batch_pinned = nullptr;
    size_t batch_pinned_bytes = 0;
    // the few-query path's feedback: rows of the probed lists that survived the radius pruning in the LAST call of this context (a
    // pinned word stage 1 writes) and the index it was about -- sizes the next call's stage-2 grid (lat_launch)
    uint32_t * hint_rows = nullptr;
    const void * hint_ix = nullptr;
    static void grow(unsigned char *& p, size_t & have, size_t bytes)
    {
        if (bytes <= have)
            return;
        if (p)
            MSVS_HIP(hipHostFree(p));
        p = nullptr;
        have = 0;
        MSVS_HIP(hipHostMalloc(reinterpret_cast<void **>(&p), bytes, hipHostMallocCoherent)); // fine-grained whatever HIP_HOST_COHERENT says
        have = bytes;
        memset(p, 0, bytes);
    }
    void need_pinned(size_t bytes) { grow(pinned, pinned_bytes, bytes); }
    void need_batch_pinned(size_t bytes) { grow(batch_pinned, batch_pinned_bytes, bytes + bytes / 2); }
};

LatCtx & lat_ctx(hipStream_t stream)
{
    static thread_local std::map<std::pair<int, hipStream_t>, LatCtx> ctxs;
    int dev = 0;
    MSVS_HIP(hipGetDevice(&dev));
    return ctxs[{dev, stream}];
}

unsigned long long * g_lat_dbg = nullptr; // experiments: msvs_lat_debug()

struct LatShape
{
    uint32_t c_rows, c_blocks, items, grid_x;
    size_t lds1, lds2;
};

LatShape lat_shape(const msvs_index & ix, size_t nq, size_t k, size_t nprobe)
{
    LatShape s{};
    // stage 1: 32 centroids per block, more when all blocks' lists would not fit the LDS of a stage-2 block
    s.c_rows = (uint32_t)(32 * ceil_div(ix.nlist * nprobe, (size_t)32 * HEADS_CAP));
    s.c_blocks = (uint32_t)ceil_div(ix.nlist, (size_t)s.c_rows);
    // work items per query: the grid (items + nprobe) is exactly 2 blocks per CU over the whole call -- a CU streams
    // ~22 KB/us whatever runs on it, so one CU with a third block sets the time of the launch (27 us against 19)
    s.items = (uint32_t)std::max<size_t>(nprobe, 2 * (size_t)device_cu_count() / nq > nprobe ? 2 * (size_t)device_cu_count() / nq - nprobe : nprobe);
    if (options().lat_items >= 1)
        s.items = (uint32_t)std::max<double>((double)nprobe, options().lat_items);
    s.grid_x = s.items + (uint32_t)nprobe;
    s.lds1 = (size_t)ix.ld * 4 + std::max((size_t)5 * nprobe * 8, lat_merge_lds(s.c_blocks, (uint32_t)nprobe));
    s.lds2 = (size_t)ix.ld * 4 + std::max((size_t)5 * k * 8, lat_merge_lds(s.grid_x, (uint32_t)k));
    return s;
}

bool lat_eligible(const msvs_index & ix, size_t nq, size_t k, size_t nprobe)
{
    // beyond two queries per call the general path's batched kernels are as fast (201 vs 195 us at 4 queries)
    if (options().lat_path == 0 || ix.type != MSVS_INDEX_IVFFLAT || !ix.ready || ix.n == 0 || nq < 1
        || nq > std::min<size_t>(LAT_MAX_Q, options().lat_path >= 2 ? LAT_MAX_Q : 2) || k < 1
        || k > LAT_MAX_K)
        return false;
    nprobe = std::min<size_t>(std::max<size_t>(nprobe, 1), ix.nlist);
    if (nprobe > LAT_MAX_K)
        return false;
    const LatShape s = lat_shape(ix, nq, k, nprobe);
    return s.lds1 <= SCAN_LDS_BUDGET && s.lds2 <= SCAN_LDS_BUDGET;
}

/// Enqueue the two launches.  Q: nq scan-ready rows of ix.ld floats (device or pinned host memory).
void lat_launch(const msvs_index & ix, LatCtx & c, const float * Q, size_t nq, uint32_t k, size_t nprobe, const uint64_t * d_alive,
                size_t nbits, int64_t * out_ids, float * out_dis, uint32_t * flag, uint32_t seq, hipStream_t stream)
{
    nprobe = std::min<size_t>(std::max<size_t>(nprobe, 1), ix.nlist);
    const uint32_t ld4 = ix.ld / 4;
    LatParams p{};
    p.Q = reinterpret_cast<const float4 *>(Q);
    p.nq = (uint32_t)nq;
    p.ld4 = ld4;
    p.k = k;
    p.nprobe = (uint32_t)nprobe;
    p.nlist = (uint32_t)ix.nlist;
    p.C = reinterpret_cast<const float4 *>(ix.centroids.p);
    LatShape sh = lat_shape(ix, nq, k, nprobe);
    // Round 6: stage 2's grid follows the rows the LAST query of this context left after the radius pruning (on clustered data 32
    // probes become 1 - 3: ~100 items of 16 rows instead of 480 blocks that mostly have nothing to scan but still arrive on the
    // merge's counter -- 42.8 -> 39.0 us per call).  A hint: the cut adapts to whatever grid it gets (lat_make_cut), a query that
    // needs more than the hint foresaw scans longer items this once.
    if (!c.hint_rows)
    {
        MSVS_HIP(hipHostMalloc(reinterpret_cast<void **>(&c.hint_rows), 64, hipHostMallocCoherent));
        *c.hint_rows = 0;
    }
    if (options().lat_hint != 0 && options().lat_items < 1 && c.hint_ix == &ix)
    {
        const uint32_t rows_seen = *reinterpret_cast<volatile uint32_t *>(c.hint_rows);
        if (rows_seen)
        {
            const uint32_t want = (uint32_t)std::min<size_t>(sh.items, std::max<size_t>(64, round_up(ceil_div((size_t)rows_seen, (size_t)16), (size_t)32)));
            sh.items = std::max<uint32_t>(want, (uint32_t)nprobe);
            sh.grid_x = sh.items + (uint32_t)nprobe;
            sh.lds2 = (size_t)ix.ld * 4 + std::max((size_t)5 * k * 8, lat_merge_lds(sh.grid_x, (uint32_t)k));
        }
    }
    c.hint_ix = &ix;
    p.hint_rows = c.hint_rows;
    p.c_rows = sh.c_rows;
    p.c_blocks = sh.c_blocks;
    p.items = sh.items;
    p.reg_select = (int)options().lat_select; // 1: radix select, 3: bitwise search, 0: list merge
    const size_t grid_x = sh.grid_x;
    const size_t n_dq = LAT_MAX_Q * (size_t)ix.ld, n_cp = nq * (size_t)p.c_blocks * nprobe, n_part = nq * grid_x * k;
    const bool grow = c.dq.n < n_dq || c.c_partial.n < n_cp || c.partial.n < n_part || !c.done.p;
    if (grow)
    {
        MSVS_HIP(hipStreamSynchronize(stream)); // an earlier call on this stream may still use the old buffers
        if (c.dq.n < n_dq)
            c.dq.alloc(n_dq);
        if (!c.probes.p)
            c.probes.alloc(LAT_MAX_Q * LAT_MAX_K);
        if (!c.probe_dis.p)
            c.probe_dis.alloc(LAT_MAX_Q * LAT_MAX_K);
        if (!c.cut.p)
            c.cut.alloc(LAT_MAX_Q * (LAT_MAX_K + 3));
        if (c.c_partial.n < n_cp)
            c.c_partial.alloc(n_cp + n_cp / 2);
        if (c.partial.n < n_part)
            c.partial.alloc(n_part + n_part / 2);
        if (!c.done.p)
        {
            c.done.alloc(2);
            MSVS_HIP(hipMemset(c.done.p, 0, 8));
            MSVS_HIP(hipDeviceSynchronize());
        }
    }
    p.dq = reinterpret_cast<float4 *>(c.dq.p);
    p.c_partial = c.c_partial.p;
    p.probes = c.probes.p;
    p.probe_dis = c.probe_dis.p;
    p.cut = c.cut.p;
    // radius pruning of the list scan (latency_kernels.hpp: lat_cut): L2 indexes, unfiltered searches
    p.radius = options().h16_prune != 0 && options().lat_prune != 0 && ix.metric == MSVS_METRIC_L2 && !d_alive && ix.list_radius.p
            && ix.cnorm_max < 1e30f && ix.xnorm_max < 1e30f
        ? ix.list_radius.p
        : nullptr;
    p.cmax = ix.cnorm_max;
    p.xmax = ix.xnorm_max;
    p.c_canon = 1.05 * 32.0 * ldexp(1.0, -24);
    p.Y = reinterpret_cast<const float4 *>(ix.vecs.p);
    p.ids = ix.row_ids.p;
    p.list_off = ix.list_off.p;
    p.alive = d_alive;
    p.nbits = (uint32_t)std::min<size_t>(nbits, 0xffffffffu);
    p.partial = c.partial.p;
    p.out_ids = out_ids;
    p.out_dis = out_dis;
    p.cosine = ix.metric == MSVS_METRIC_COSINE;
    p.done = c.done.p;
    p.flag = flag;
    p.seq = seq;
    p.dbg = g_lat_dbg;
    const size_t lds1 = sh.lds1, lds2 = sh.lds2;
    const dim3 g1(p.c_blocks, (unsigned)nq), g2((unsigned)grid_x, (unsigned)nq);
    ProfileScope prof("lat_search", stream);
    if (ix.metric == MSVS_METRIC_L2)
    {
        hipLaunchKernelGGL((lat_coarse_kernel<M_L2>), g1, dim3(BLOCK), lds1, stream, p);
        hipLaunchKernelGGL((lat_scan_kernel<M_L2>), g2, dim3(BLOCK), lds2, stream, p);
    }
    else
    {
        hipLaunchKernelGGL((lat_coarse_kernel<M_IP>), g1, dim3(BLOCK), lds1, stream, p);
        hipLaunchKernelGGL((lat_scan_kernel<M_IP>), g2, dim3(BLOCK), lds2, stream, p);
    }
    MSVS_HIP(hipGetLastError());
}

}

/// Step 1 of the general IVFFLAT search for a SMALL batch (msvs_capi.hip: index_search_device_one): the canonical top-nprobe of the
/// centroids in ONE launch -- grid (centroid blocks, groups of 1 / 4 queries), the last block of a group (a per-group arrival
/// counter) selects its queries' probes: coarse_dense_kernel for tables of <= 2048 centroids, coarse_few_kernel beyond -- where the
/// batched kernels take a scan and a merge launch (22 + 14 us for 32 queries over 1024 centroids; this: one launch of ~20 us, most
/// of it launch, arrival and the selection).  Same canonical arithmetic and total order (the probes are the oracle's; they leave
/// in arbitrary order, which no consumer of the general path depends on; a sharded search's probe lists keep the batched form).
/// dq: nq scan-ready rows.  probe_dis: nullable.  -> false: not for this shape.
bool msvs::coarse_few_launch(const msvs_index & ix, const float * dq, size_t nq, size_t nprobe, int32_t * d_probes, float * d_probe_dis,
                             hipStream_t stream)
{
    if (options().coarse_few == 0 || nq < 1 || nq > std::min<size_t>(LAT_COARSE_MAX_Q, (size_t)options().coarse_few) || nprobe < 1
        || nprobe > LAT_MAX_K || nprobe > ix.nlist || options().lat_select == 0)
        return false;
    // queries per block: one while the table's re-reads are cheap (<= 16 queries: 41.7 against 49.7 us per 4-query search, 129
    // against 134 at 16), LAT_COARSE_T beyond (137.8 against 144.5 us at 32, 164 against 186 at 64)
    const int T = nq <= 16 ? 1 : LAT_COARSE_T;
    const size_t groups = ceil_div(nq, (size_t)T);
    // a table of at most 32 * WAVE centroids: every (query, centroid) key goes out, the selection reads them all
    // (coarse_dense_kernel: no top-k in the scanning blocks); a larger one: nprobe keys per block (coarse_few_kernel)
    const bool dense = ix.nlist <= 32 * WAVE && options().coarse_dense != 0;
    // centroids per block: 16 / 32, more when a query's lists would not fit the selection's registers or the launch would be
    // far beyond four blocks per CU
    size_t c_rows = dense ? 16 : 32 * ceil_div(ix.nlist * nprobe, (size_t)32 * 32 * WAVE);
    while (c_rows < 256 && ceil_div(ix.nlist, c_rows) * groups > (size_t)4 * device_cu_count())
        c_rows *= 2;
    const size_t c_blocks = ceil_div(ix.nlist, c_rows);
    const size_t lds = (size_t)T * ix.ld * 4 + (dense ? 0 : (size_t)T * 5 * nprobe * 8);
    if (lds > SCAN_LDS_BUDGET || (!dense && c_blocks * nprobe > 32 * WAVE))
        return false;
    LatCtx & c = lat_ctx(stream);
    const size_t n_cp = dense ? nq * ix.nlist : nq * c_blocks * nprobe;
    if (c.cf_partial.n < n_cp || !c.cf_done.p)
    {
        MSVS_HIP(hipStreamSynchronize(stream)); // an earlier call on this stream may still use the old buffers
        if (c.cf_partial.n < n_cp)
            c.cf_partial.alloc(LAT_COARSE_MAX_Q * (size_t)32 * WAVE);
        if (!c.cf_done.p)
        {
            c.cf_done.alloc(LAT_COARSE_MAX_Q);
            c.cf_dis.alloc(LAT_COARSE_MAX_Q * LAT_MAX_K);
            MSVS_HIP(hipMemset(c.cf_done.p, 0, LAT_COARSE_MAX_Q * 4));
            MSVS_HIP(hipDeviceSynchronize());
        }
    }
    LatParams p{};
    p.Q = reinterpret_cast<const float4 *>(dq);
    p.nq = (uint32_t)nq;
    p.ld4 = ix.ld / 4;
    p.nprobe = (uint32_t)nprobe;
    p.nlist = (uint32_t)ix.nlist;
    p.C = reinterpret_cast<const float4 *>(ix.centroids.p);
    p.c_rows = (uint32_t)c_rows;
    p.c_blocks = (uint32_t)c_blocks;
    p.reg_select = (int)options().lat_select;
    p.c_partial = c.cf_partial.p;
    p.probes = d_probes;
    p.probe_dis = d_probe_dis ? d_probe_dis : c.cf_dis.p;
    p.done_q = c.cf_done.p;
    const dim3 g1(p.c_blocks, (unsigned)groups);
    ProfileScope prof("coarse_few", stream);
    if (dense)
    {
        if (ix.metric == MSVS_METRIC_L2)
        {
            if (T == 1)
                hipLaunchKernelGGL((coarse_dense_kernel<M_L2, 1>), g1, dim3(BLOCK), lds, stream, p);
            else
                hipLaunchKernelGGL((coarse_dense_kernel<M_L2, LAT_COARSE_T>), g1, dim3(BLOCK), lds, stream, p);
        }
        else if (T == 1)
            hipLaunchKernelGGL((coarse_dense_kernel<M_IP, 1>), g1, dim3(BLOCK), lds, stream, p);
        else
            hipLaunchKernelGGL((coarse_dense_kernel<M_IP, LAT_COARSE_T>), g1, dim3(BLOCK), lds, stream, p);
    }
    else if (ix.metric == MSVS_METRIC_L2)
    {
        if (T == 1)
            hipLaunchKernelGGL((coarse_few_kernel<M_L2, 1>), g1, dim3(BLOCK), lds, stream, p);
        else
            hipLaunchKernelGGL((coarse_few_kernel<M_L2, LAT_COARSE_T>), g1, dim3(BLOCK), lds, stream, p);
    }
    else if (T == 1)
        hipLaunchKernelGGL((coarse_few_kernel<M_IP, 1>), g1, dim3(BLOCK), lds, stream, p);
    else
        hipLaunchKernelGGL((coarse_few_kernel<M_IP, LAT_COARSE_T>), g1, dim3(BLOCK), lds, stream, p);
    MSVS_HIP(hipGetLastError());
    return true;
}

namespace
{
/// VectorDataset::normalize() of one row on the host: the arithmetic of normalize_rows_kernel (strictly sequential f32
/// sum of squares, IEEE sqrt and divide), so a query prepared here equals one prepared on the device bit for bit.
void normalize_row_host(float * p, uint32_t d)
{
    volatile float sum = 0.f;
    for (uint32_t j = 0; j < d; j++)
    {
        volatile float sq = p[j] * p[j];
        sum = sum + sq;
    }
    if (sum < 1.1920928955078125e-7f)
        return;
    const float s = sqrtf(sum);
    for (uint32_t j = 0; j < d; j++)
        p[j] = p[j] / s;
}

/// Host-pointer search of a few queries: queries go in through pinned memory the kernels read directly, results and a
/// completion word come back the same way (no memcpy calls, no stream synchronisation: the host thread spins on the
/// word).  d_alive: the effective filter, already on the device and ordered on `stream`.
void lat_search_host(const msvs_index & ix, const float * queries, size_t nq, uint32_t k, size_t nprobe, const uint64_t * d_alive,
                     size_t nbits, int64_t * ids, float * dis, hipStream_t stream)
{
    LatCtx & c = lat_ctx(stream);
    const size_t ld = ix.ld, o_ids = round_up(LAT_MAX_Q * ld * 4, (size_t)256), o_dis = o_ids + LAT_MAX_Q * LAT_MAX_K * 8,
                 o_flag = o_dis + LAT_MAX_Q * LAT_MAX_K * 4;
    c.need_pinned(o_flag + 256);
    float * hq = reinterpret_cast<float *>(c.pinned);
    for (size_t q = 0; q < nq; q++)
    {
        float * row = hq + q * ld;
        memcpy(row, queries + q * ix.dim, ix.dim * 4);
        for (size_t j = ix.dim; j < ld; j++)
            row[j] = 0.f;
        if (ix.metric == MSVS_METRIC_COSINE)
            normalize_row_host(row, (uint32_t)ix.dim);
    }
    volatile uint32_t * flag = reinterpret_cast<volatile uint32_t *>(c.pinned + o_flag);
    const uint32_t seq = ++c.seq ? c.seq : ++c.seq; // never 0: the word starts at 0
    int64_t * h_ids = reinterpret_cast<int64_t *>(c.pinned + o_ids);
    float * h_dis = reinterpret_cast<float *>(c.pinned + o_dis);
    lat_launch(ix, c, hq, nq, k, nprobe, d_alive, nbits, h_ids, h_dis, const_cast<uint32_t *>(flag), seq, stream);
    // acquire: the copies of the results below are ordered after the word (a volatile read alone orders nothing for the compiler)
    for (uint64_t spins = 1; __atomic_load_n(const_cast<const uint32_t *>(flag), __ATOMIC_ACQUIRE) != seq; spins++)
    {
        __builtin_ia32_pause();
        if ((spins & 0xfff) == 0) // a failed launch never sets the word: look at the stream now and then
        {
            const hipError_t e = hipStreamQuery(stream);
            if (e == hipSuccess)
                break;
            if (e != hipErrorNotReady)
                fail(MSVS_ERR_DEVICE, "few-query search: %s", hipGetErrorString(e));
        }
    }
    if (*flag != seq)
        MSVS_HIP(hipStreamSynchronize(stream));
    memcpy(ids, h_ids, nq * k * 8);
    memcpy(dis, h_dis, nq * k * 4);
}
}

namespace
{
/// See index_search_host_call.  The device-level search reads the queries from pinned memory (dense, ix.dim floats per row) and
/// writes ids / distances to pinned memory; when the table pass used the signal the results are complete at its return.
void flat_few_search_host(const msvs_index & ix, const float * queries, size_t nq, uint32_t k, int64_t * ids, float * dis, hipStream_t stream)
{
    LatCtx & c = lat_ctx(stream);
    const size_t o_ids = round_up(16 * ix.dim * 4, (size_t)256), o_dis = o_ids + 16 * 40 * 8, o_flag = o_dis + 16 * 40 * 4;
    c.need_pinned(std::max<size_t>(o_flag + 256, c.pinned_bytes));
    float * hq = reinterpret_cast<float *>(c.pinned);
    memcpy(hq, queries, nq * ix.dim * 4);
    int64_t * h_ids = reinterpret_cast<int64_t *>(c.pinned + o_ids);
    float * h_dis = reinterpret_cast<float *>(c.pinned + o_dis);
    uint32_t * flag = reinterpret_cast<uint32_t *>(c.pinned + o_flag);
    HostSignal & hs = host_signal();
    hs.flag = flag;
    hs.nfail = flag + 16;
    hs.reset(); // (the slot's position depends on the call's shape: whatever an earlier, larger call left there must not read as this call's word)
    hs.seq = ++c.seq ? c.seq : ++c.seq; // never 0: the word starts at 0
    hs.armed = true;
    hs.used = false;
    try
    {
        index_search_device(ix, hq, nq, k, 1, nullptr, 0, h_ids, h_dis, stream);
    }
    catch (...)
    {
        hs.armed = false;
        throw;
    }
    hs.armed = false;
    if (!hs.used) // another path served the call (a small table, an option): its results are in flight
        MSVS_HIP(hipStreamSynchronize(stream));
    memcpy(ids, h_ids, nq * k * 8);
    memcpy(dis, h_dis, nq * k * 4);
}
}

extern "C" int msvs_index_search_device(const msvs_index_t * ix, const float * d_queries, size_t nq, int k, int nprobe,
                                        const uint64_t * d_alive_bits, size_t nbits, int64_t * d_ids, float * d_dis,
                                        void * hip_stream)
{
    return guarded([&] {
        if (!ix || (nq && (!d_queries || !d_ids || !d_dis)) || k < 0)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null index/buffer or negative k");
        const auto meta = ix->get_meta();
        size_t eff_bits = nbits;
        const uint64_t * eff = effective_filter(*ix, meta.get(), d_alive_bits, nbits, &eff_bits, as_stream(hip_stream));
        if (k > 0 && lat_eligible(*ix, nq, (size_t)k, (size_t)std::max(nprobe, 0)) && ix->ld == ix->dim && ix->metric != MSVS_METRIC_COSINE)
            // a few scan-ready queries: two launches (latency_kernels.hpp)
            lat_launch(*ix, lat_ctx(as_stream(hip_stream)), d_queries, nq, (uint32_t)k, (size_t)std::max(nprobe, 0), eff, eff_bits, d_ids,
                       d_dis, nullptr, 0, as_stream(hip_stream));
        else
            index_search_device(*ix, d_queries, nq, (uint32_t)k, (size_t)std::max(nprobe, 0), eff, eff_bits, d_ids, d_dis,
                                as_stream(hip_stream));
        apply_row_ids_map(meta.get(), d_ids, nq * (size_t)k, as_stream(hip_stream));
    });
}


namespace msvs
{
int index_search_host_call(const msvs_index_t * ix, const float * queries, size_t nq, int k, const char * params,
                                  const uint64_t * alive_bits, size_t nbits, int64_t * ids, float * dis)
{
    return guarded([&] {
        if (!ix || (nq && (!queries || !ids || !dis)) || k < 0)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null index/buffer or negative k");
        if (!ix->ready)
            fail(MSVS_ERR_NOT_READY, "index is not ready");
        if (nq == 0 || k == 0)
            return;
        DeviceGuard on_device(ix->device);
        if ((size_t)k > MSVS_MAX_K_ROUNDS)
            fail(MSVS_ERR_UNSUPPORTED_K, "k = %d exceeds the limit %d", k, MSVS_MAX_K_ROUNDS);
        auto p = parse_params(params);
        for (const auto & kv : p)
            if (kv.first != "nprobe")
                fail(MSVS_ERR_INVALID_ARGUMENT, "unknown search parameter `%s`", kv.first.c_str());
        long nprobe = param_int(p, "nprobe", 1);
        if (nprobe < 1)
            fail(MSVS_ERR_INVALID_ARGUMENT, "nprobe must be >= 1");
        hipStream_t stream = thread_stream();
        // a filter that is PRESENT with zero valid bits means "no row passes" (not "no filter"): it still travels as one
        // zero word with nbits = 0, and every id fails the `id < nbits` test
        const bool filtered = alive_bits != nullptr;
        const size_t words = filtered ? std::max<size_t>(1, ceil_div(nbits, 64)) : 0;
        // host staging lives in its own arena (scratch_for() belongs to the device-level search underneath)
        Scratch & stg = staging_for(stream);
        stg.reserve(nq * ix->dim * 4 + nq * (size_t)k * 12 + words * 8 + 4096, stream);
        const DevView<float> dq{stg.take<float>(nq * ix->dim)};
        const DevView<int64_t> d_ids{stg.take<int64_t>(nq * (size_t)k)};
        const DevView<float> d_dis{stg.take<float>(nq * (size_t)k)};
        const DevView<uint64_t> d_alive{words ? stg.take<uint64_t>(words) : nullptr};
        if (filtered)
        {
            MSVS_HIP(hipMemsetAsync(d_alive.p, 0, words * 8, stream));
            if (nbits)
                MSVS_HIP(hipMemcpyAsync(d_alive.p, alive_bits, ceil_div(nbits, 64) * 8, hipMemcpyHostToDevice, stream));
        }
        const auto meta = ix->get_meta();
        size_t eff_bits = nbits;
        const uint64_t * eff = effective_filter(*ix, meta.get(), words ? d_alive.p : nullptr, nbits, &eff_bits, stream);
        const bool eff_filtered = eff != nullptr;
        if (lat_eligible(*ix, nq, (size_t)k, (size_t)nprobe) && !(meta && meta->row_ids_n))
        {
            // a few queries: two launches, queries and results through pinned memory (no copies, no stream sync)
            lat_search_host(*ix, queries, nq, (uint32_t)k, (size_t)nprobe, eff, eff_bits, ids, dis, stream);
            return;
        }
        if (ix->type == MSVS_INDEX_FLAT && !eff_filtered && nq < 16 && (size_t)k <= 40 && ix->shadow_ready && options().flat_few != 0
            && options().flat_host_signal != 0 && !(meta && meta->row_ids_n))
        {
            // a few queries over a FLAT index (VIWithDataPart.cpp:922-926 with IndexType::FLAT: one query per call): queries in and
            // results out through pinned memory, a completion word instead of copies + synchronisation, the canonical fallback
            // launched only when a certificate failed (HostSignal, index_internal.hpp)
            flat_few_search_host(*ix, queries, nq, (uint32_t)k, ids, dis, stream);
            return;
        }
        if (ix->type == MSVS_INDEX_IVFFLAT && !filtered && !eff_filtered && (size_t)k <= MSVS_MAX_K && nq <= (size_t)options().host_pinned
            && !(meta && meta->row_ids_n))
        {
            // a small batch (a combined batch of concurrent single-query callers; a few rows of a batch_distance call): the queries
            // go up from pinned memory and the result kernels write ids / distances straight into pinned memory -- the copy from
            // pageable memory and the two copies back cost ~35 of the call's ~185 us at 32 queries
            LatCtx & c = lat_ctx(stream);
            const size_t o_ids = round_up(nq * ix->dim * 4, (size_t)256), o_dis = o_ids + round_up(nq * (size_t)k * 8, (size_t)256);
            const size_t o_flag = round_up(o_dis + nq * (size_t)k * 4, (size_t)256);
            if (o_flag + 256 > c.batch_pinned_bytes)
                c.need_batch_pinned(o_flag + 256);
            memcpy(c.batch_pinned, queries, nq * ix->dim * 4);
            int64_t * h_ids = reinterpret_cast<int64_t *>(c.batch_pinned + o_ids);
            float * h_dis = reinterpret_cast<float *>(c.batch_pinned + o_dis);
            fetch_from_pinned(dq.p, c.batch_pinned, nq * ix->dim * 4, stream); // (a copy kernel: no barrier packets around it)
            // the shadow list scan ends with a completion word instead of its normally-empty fallback launches (HostSignal)
            HostSignal & hs = host_signal();
            uint32_t * flag = reinterpret_cast<uint32_t *>(c.batch_pinned + o_flag);
            hs.flag = flag;
            hs.nfail = flag + 16;
            hs.reset(); // (see flat_few_search_host: the slot may hold an earlier call's ids)
            hs.seq = ++c.seq ? c.seq : ++c.seq; // never 0: the word starts at 0
            hs.armed = options().host_signal_batch != 0 && nq <= 256; // (one pass of the device-level search: index_search_device splits beyond)
            hs.used = false;
            try
            {
                index_search_device(*ix, dq.p, nq, (uint32_t)k, (size_t)nprobe, nullptr, 0, h_ids, h_dis, stream);
            }
            catch (...)
            {
                hs.armed = false;
                throw;
            }
            hs.armed = false;
            if (!hs.used) // (another path than the shadow list scan served the batch: its results are in flight)
                MSVS_HIP(hipStreamSynchronize(stream));
            memcpy(ids, h_ids, nq * (size_t)k * 8);
            memcpy(dis, h_dis, nq * (size_t)k * 4);
            return;
        }
        MSVS_HIP(hipMemcpyAsync(dq.p, queries, nq * ix->dim * 4, hipMemcpyHostToDevice, stream));
        if ((size_t)k <= MSVS_MAX_K && filtered)
        {
            // the strategy (bit test / compacted view) goes by how many rows the caller's bitmap lets through
            uint64_t alive_count = 0;
            for (size_t w = 0; w < ceil_div(nbits, (size_t)64); w++)
                alive_count += (uint64_t)__builtin_popcountll(alive_bits[w]);
            index_search_filtered(*ix, dq.p, nq, (uint32_t)k, (size_t)nprobe, eff, eff_bits, alive_count, d_ids.p, d_dis.p, stream);
        }
        else if ((size_t)k <= MSVS_MAX_K)
            index_search_device(*ix, dq.p, nq, (uint32_t)k, (size_t)nprobe, eff, eff_bits, d_ids.p, d_dis.p, stream);
        else
        {
            // k beyond one wavefront top-k pass: rounds of MSVS_MAX_K per query, each round excluding the rows already
            // returned through a private copy of the filter bitmap (exact: round r returns ranks 256r .. 256r+255)
            const size_t idspace = std::max<size_t>(eff_filtered ? eff_bits : 0, (size_t)ix->max_id + 1);
            const size_t bw = ceil_div(idspace, 64);
            DevBuf<uint64_t> bm(bw);
            for (size_t q = 0; q < nq; q++)
            {
                if (eff_filtered)
                {
                    MSVS_HIP(hipMemsetAsync(bm.p, 0, bw * 8, stream));
                    MSVS_HIP(hipMemcpyAsync(bm.p, eff, std::max<size_t>(1, ceil_div(eff_bits, (size_t)64)) * 8,
                                            hipMemcpyDeviceToDevice, stream));
                }
                else
                    MSVS_HIP(hipMemsetAsync(bm.p, 0xFF, bw * 8, stream));
                for (size_t done = 0; done < (size_t)k; done += MSVS_MAX_K)
                {
                    const uint32_t kr = (uint32_t)std::min<size_t>(MSVS_MAX_K, (size_t)k - done);
                    int64_t * oi = d_ids.p + q * (size_t)k + done;
                    index_search_device(*ix, dq.p + q * ix->dim, 1, kr, (size_t)nprobe, bm.p, eff_filtered ? eff_bits : idspace,
                                        oi, d_dis.p + q * (size_t)k + done, stream);
                    hipLaunchKernelGGL(clear_bits_kernel, dim3(1), dim3(256), 0, stream, bm.p, oi, kr);
                    MSVS_HIP(hipGetLastError());
                }
            }
        }
        apply_row_ids_map(meta.get(), d_ids.p, nq * (size_t)k, stream);
        MSVS_HIP(hipMemcpyAsync(ids, d_ids.p, nq * (size_t)k * 8, hipMemcpyDeviceToHost, stream));
        MSVS_HIP(hipMemcpyAsync(dis, d_dis.p, nq * (size_t)k * 4, hipMemcpyDeviceToHost, stream));
        MSVS_HIP(hipStreamSynchronize(stream));
    });
}
}

// ------------------------------------------------------------------------------------------ combining concurrent callers
//
// The reference's host calls VectorIndex::search from up to ScanThreadLimiter-many threads, one query each
// (MergeTreeVSManager.cpp:973).  One query is a whole-GPU job of ~45 us here, so beyond a handful of concurrent callers the
// calls only queue behind each other on the device (64 threads: 26 k QPS), while ONE batched search of 64 queries takes
// 0.2 ms.  So: up to `combine` (8) single calls run directly, each on its thread's stream, as long as nobody waits; callers
// beyond that wait in a queue, and the next search to finish while no batch is in flight collects EVERY compatible waiter's
// query (same k, same parameter string, no filter) into one batch for the index's WORKER thread, which runs it, wakes its
// callers and collects the next batch itself while callers keep arriving.  No timer, no extra latency for a lone caller;
// results are the same bits either way (every path is exact).  msvs_combine_stats counts the batches.
// Round 5 (tools/r5_threads.py; 64 callers, batches of ~32): the batches ran on the first waiter's thread before -- every
// batch on another thread with its own stream, scratch arenas and pinned block (cold, or not allocated yet), behind a
// finisher that woke 31 callers under the lock the new leader needed: 307 us per batch cycle around a 174 us search,
// 104 k QPS.  One warm worker thread + per-request wake-ups spread over four of the callers: 188 us per cycle, 166 k QPS
// (128 callers: 134 k -> 269 k).
namespace
{
struct CombineReq
{
    const float * q;
    size_t nq;
    int k;
    std::string params;
    int64_t * ids;
    float * dis;
    int status = 0;
    std::string err;
    std::vector<CombineReq *> batch;
    unsigned long long t_handed = 0; // (experiments) when a finisher made this request the next leader
    // 0 waiting, 2 served.  Every request waits on its OWN mutex / condition variable (the combiner's mutex guards the queue only):
    // whoever wakes the callers of a batch holds no lock anybody else needs (with one mutex for everything the next batch sat behind
    // 31 wake-ups: ~130 us of a ~300 us cycle at 64 callers).
    std::atomic<int> state{0};
    std::vector<CombineReq *> to_wake; // a served request may be handed more of its batch to wake (the worker wakes a few heads only)
    std::mutex m;
    std::condition_variable cv;
    void wake(int s)
    {
        std::lock_guard<std::mutex> l(m); // (the waiter passes this mutex before it returns: `this` outlives the call)
        state.store(s, std::memory_order_release);
        cv.notify_one();
    }
    int wait()
    {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return state.load(std::memory_order_acquire) != 0; });
        return state.load(std::memory_order_acquire);
    }
};
struct Combiner
{
    std::mutex mu;
    std::deque<CombineReq *> queue;
    int active = 0;  // searches running (single calls and batches)
    int batches = 0; // ... of which batches of several callers
    // The batches run on ONE worker thread per index: one stream, one set of scratch arenas and pinned blocks, all of them warm.
    // (Round 4 made the first waiter the leader of the next batch: with 64 callers every batch ran on another thread, each with
    // its own stream, arenas and pinned block to allocate and fault in -- 40 to 170 us per batch beside the search itself.)
    const msvs_index * ix = nullptr;
    std::thread worker;
    std::mutex wmu;
    std::condition_variable wcv;
    std::deque<CombineReq *> handed; // collected batches (their first requests) the worker runs next (guarded by wmu)
    std::atomic<int> n_handed{0};
    std::atomic<bool> stop{false};
    CombineReq * take() // wmu held
    {
        if (handed.empty())
            return nullptr;
        CombineReq * r = handed.front();
        handed.pop_front();
        n_handed.fetch_sub(1, std::memory_order_relaxed);
        return r;
    }
    void submit(CombineReq * lead); // a finisher hands the batch over
    void worker_main();
    ~Combiner()
    {
        if (worker.joinable())
        {
            {
                std::lock_guard<std::mutex> l(wmu);
                stop.store(true);
                wcv.notify_one();
            }
            worker.join();
        }
    }
};
std::mutex g_comb_mu;
std::unordered_map<const msvs_index *, std::shared_ptr<Combiner>> g_comb;
std::atomic<unsigned long long> g_comb_calls{0}, g_comb_batches{0}, g_comb_batched{0};
// experiments (msvs_debug_combine_times): ns of batches of several callers -- hand-over (the previous finisher's decision -> this leader
// running), gather, the search call, distribution
std::atomic<unsigned long long> g_comb_ns[4];
inline unsigned long long comb_now() { return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
constexpr size_t COMBINE_MAX_QUERIES = 1024;

std::shared_ptr<Combiner> combiner_of(const msvs_index * ix)
{
    std::lock_guard<std::mutex> lk(g_comb_mu);
    auto & c = g_comb[ix];
    if (!c)
    {
        c = std::make_shared<Combiner>();
        c->ix = ix;
    }
    return c;
}
}
namespace msvs
{
void combiner_forget(const msvs_index * ix)
{
    std::lock_guard<std::mutex> lk(g_comb_mu);
    g_comb.erase(ix);
}
}
namespace
{

/// Runs the leader's batch: one request = the plain call into its own buffers; several = one gathered search.
void combine_run(const msvs_index * ix, CombineReq & lead)
{
    auto & b = lead.batch;
    if (b.size() == 1)
    {
        lead.status = index_search_host_call(ix, lead.q, lead.nq, lead.k, lead.params.c_str(), nullptr, 0, lead.ids, lead.dis);
        if (lead.status)
            lead.err = msvs_last_error();
        return;
    }
    size_t total = 0;
    for (auto * r : b)
        total += r->nq;
    const size_t d = ix->dim, k = (size_t)lead.k;
    const unsigned long long t0 = comb_now();
    if (lead.t_handed)
        g_comb_ns[0].fetch_add(t0 - lead.t_handed, std::memory_order_relaxed);
    static thread_local std::vector<float> qbuf, dbuf;
    static thread_local std::vector<int64_t> ibuf;
    qbuf.resize(total * d);
    ibuf.resize(total * k);
    dbuf.resize(total * k);
    size_t at = 0;
    for (auto * r : b)
    {
        memcpy(qbuf.data() + at * d, r->q, r->nq * d * 4);
        at += r->nq;
    }
    const unsigned long long t1 = comb_now();
    const int rc = index_search_host_call(ix, qbuf.data(), total, lead.k, lead.params.c_str(), nullptr, 0, ibuf.data(), dbuf.data());
    const unsigned long long t2 = comb_now();
    const std::string err = rc ? msvs_last_error() : "";
    at = 0;
    for (auto * r : b)
    {
        if (!rc)
        {
            memcpy(r->ids, ibuf.data() + at * k, r->nq * k * 8);
            memcpy(r->dis, dbuf.data() + at * k, r->nq * k * 4);
        }
        r->status = rc;
        r->err = err;
        at += r->nq;
    }
    g_comb_batches.fetch_add(1, std::memory_order_relaxed);
    g_comb_batched.fetch_add(total, std::memory_order_relaxed);
    g_comb_ns[1].fetch_add(t1 - t0, std::memory_order_relaxed);
    g_comb_ns[2].fetch_add(t2 - t1, std::memory_order_relaxed);
    g_comb_ns[3].fetch_add(comb_now() - t2, std::memory_order_relaxed);
}

/// `lead` takes every compatible waiter's queries with it (same k, same parameter string; waiters with another k / parameter
/// string follow in a later batch).  Called with the combiner's mutex held.
void combine_collect(Combiner & c, CombineReq * lead)
{
    lead->batch.assign(1, lead);
    size_t total = lead->nq;
    for (auto it = c.queue.begin(); it != c.queue.end() && total < COMBINE_MAX_QUERIES;)
        if ((*it)->k == lead->k && (*it)->params == lead->params && total + (*it)->nq <= COMBINE_MAX_QUERIES)
        {
            total += (*it)->nq;
            lead->batch.push_back(*it);
            it = c.queue.erase(it);
        }
        else
            ++it;
    if (lead->batch.size() > 1)
        c.batches++;
}

void combine_run_guarded(const msvs_index * ix, CombineReq & lead)
{
    try
    {
        combine_run(ix, lead); // the C entry underneath translates its own exceptions; what is left is the gather's allocation
    }
    catch (...)
    {
        for (auto * r : lead.batch)
        {
            r->status = MSVS_ERR_OUT_OF_MEMORY;
            r->err = "host allocation failed while combining concurrent searches";
        }
    }
}

/// `done` has run: the books, and the next batch if callers are waiting (-> its first request, collected; else nullptr).
CombineReq * combine_finish(Combiner & c, const CombineReq & done)
{
    const int max_batches = std::max(1, (int)options().combine_batches);
    std::lock_guard<std::mutex> lk(c.mu);
    c.active--;
    if (done.batch.size() > 1)
        c.batches--;
    if (c.queue.empty() || c.batches >= max_batches)
        return nullptr;
    c.active++;
    CombineReq * next = c.queue.front();
    c.queue.pop_front();
    combine_collect(c, next);
    next->t_handed = comb_now();
    return next;
}

void Combiner::submit(CombineReq * lead)
{
    std::lock_guard<std::mutex> l(wmu);
    if (!worker.joinable())
        worker = std::thread([this] { worker_main(); });
    handed.push_back(lead);
    n_handed.fetch_add(1, std::memory_order_release);
    wcv.notify_one();
}

void Combiner::worker_main()
{
    (void)hipSetDevice(ix->device);
    for (;;)
    {
        // the next batch: polled for a while after the last one (a futex wake-up is ~50 us of idle device), then slept for
        CombineReq * lead = nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        const auto spin = std::chrono::microseconds((long)options().combine_spin);
        for (uint32_t i = 1; n_handed.load(std::memory_order_acquire) == 0 && !stop.load(std::memory_order_relaxed); i++)
        {
            __builtin_ia32_pause();
            if ((i & 0xff) == 0 && std::chrono::steady_clock::now() - t0 > spin)
                break;
        }
        {
            std::unique_lock<std::mutex> l(wmu);
            wcv.wait(l, [&] { return !handed.empty() || stop.load(); });
            lead = take();
            if (!lead)
                return;
        }
        while (lead)
        {
            combine_run_guarded(ix, *lead);
            CombineReq * next = combine_finish(*this, *lead);
            // the callers of this batch: the worker wakes four of them and each of those its share of the rest -- 32 wake-ups are
            // ~100 us of system calls the next batch would wait behind.  (A woken request is gone: the list is taken first.)
            std::vector<CombineReq *> members;
            members.swap(lead->batch);
            const size_t heads = std::min<size_t>(4, members.size());
            for (size_t i = heads; i < members.size(); i++)
                members[i % heads]->to_wake.push_back(members[i]);
            for (size_t i = 0; i < heads; i++)
                members[i]->wake(2);
            lead = next;
        }
    }
}

int combined_search(const msvs_index * ix, const float * queries, size_t nq, int k, const char * params, int64_t * ids, float * dis)
{
    const int max_direct = (int)options().combine;
    auto comb = combiner_of(ix); // keeps the combiner alive across a concurrent msvs_index_free (which is a caller bug anyway)
    Combiner & c = *comb;
    CombineReq me;
    me.q = queries;
    me.nq = nq;
    me.k = k;
    me.params = params ? params : "";
    me.ids = ids;
    me.dis = dis;
    g_comb_calls.fetch_add(1, std::memory_order_relaxed);
    std::unique_lock<std::mutex> lk(c.mu);
    // direct while nobody waits and no batch is in flight (a batch uses the whole device well; single calls next to it
    // would only slow it down and keep the next batch small)
    const int max_batches = std::max(1, (int)options().combine_batches);
    if (c.active == 0 || (c.active < max_direct && c.queue.empty() && c.batches == 0))
    {
        c.active++;
        me.batch.assign(1, &me);
    }
    else if (c.batches >= 1 && c.batches < max_batches && !c.queue.empty())
    {
        // a batch is running and callers are waiting already: this one leads them NOW (a second batch in flight keeps the device
        // busy through the first one's completion, hand-over and wake-ups; whoever arrives meanwhile waits for the next finisher)
        c.active++;
        combine_collect(c, &me);
    }
    else
    {
        c.queue.push_back(&me);
        lk.unlock();
        me.wait(); // (served by the worker's batch)
        for (auto * r : me.to_wake)
            r->wake(2);
        if (me.status)
            set_last_error(me.err);
        return me.status;
    }
    lk.unlock();
    combine_run_guarded(ix, me);
    CombineReq * next = combine_finish(c, me);
    if (next)
        c.submit(next);
    for (auto * r : me.batch) // (a second batch in flight, led by this caller: option combine_batches)
        if (r != &me)
            r->wake(2);
    if (me.status)
        set_last_error(me.err);
    return me.status;
}
}

extern "C" int msvs_index_search(const msvs_index_t * ix, const float * queries, size_t nq, int k, const char * params,
                                 const uint64_t * alive_bits, size_t nbits, int64_t * ids, float * dis)
{
    // few unfiltered queries on a ready index: through the combiner (anything else, and every argument error, directly)
    if (options().combine >= 1 && ix && ix->ready && nq >= 1 && nq <= 4 && !alive_bits && queries && ids && dis && k >= 1
        && (size_t)k <= MSVS_MAX_K)
        return combined_search(ix, queries, nq, k, params, ids, dis);
    return index_search_host_call(ix, queries, nq, k, params, alive_bits, nbits, ids, dis);
}

/// calls that went through the combiner, batches of more than one caller, queries served by such batches
extern "C" __attribute__((visibility("default"))) int msvs_debug_combine_times(uint64_t * out4)
{
    for (int i = 0; i < 4; i++)
        out4[i] = g_comb_ns[i].load();
    return 0;
}

extern "C" int msvs_combine_stats(uint64_t * calls, uint64_t * batches, uint64_t * batched_queries)
{
    if (calls)
        *calls = g_comb_calls.load();
    if (batches)
        *batches = g_comb_batches.load();
    if (batched_queries)
        *batched_queries = g_comb_batched.load();
    return MSVS_OK;
}


/// Experiments only (not in msvs.h): wall-clock stamps (100 MHz) of the last blocks of the two few-query launches.
extern "C" __attribute__((visibility("default"))) int msvs_lat_debug(unsigned long long * out16)
{
    return guarded([&] {
        if (!g_lat_dbg)
        {
            MSVS_HIP(hipMalloc(&g_lat_dbg, 16 * 8));
            MSVS_HIP(hipMemset(g_lat_dbg, 0, 16 * 8));
        }
        MSVS_HIP(hipDeviceSynchronize());
        if (out16)
            MSVS_HIP(hipMemcpy(out16, g_lat_dbg, 16 * 8, hipMemcpyDeviceToHost));
    });
}
