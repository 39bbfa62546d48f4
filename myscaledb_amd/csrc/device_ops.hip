// device_ops.hip -- kernel instantiation + launch dispatch for the scan / merge kernels.
#include "device_ops.hpp"

#include <atomic>
#include <map>
#include <mutex>

namespace msvs
{

static thread_local std::string g_last_error;

void set_last_error(const std::string & m) { g_last_error = m; }
const char * last_error_cstr() { return g_last_error.c_str(); }

void Scratch::reserve(size_t bytes, hipStream_t stream)
{
    used = 0;
    // bounded: an arena that has not needed a quarter of its capacity for 64 operations in a row (a full window after the large one) gives the rest back
    // (one unusually large batch must not pin hundreds of MB per client thread for the life of the process)
    recent_max = std::max(recent_max, bytes);
    if (++ops >= 64)
    {
        if (buf.n > ((size_t)64 << 20) && recent_max * 4 < buf.n && bytes <= recent_max * 2)
        {
            MSVS_HIP(hipStreamSynchronize(stream));
            buf.alloc(recent_max * 2);
        }
        ops = 0;
        recent_max = 0;
    }
    if (bytes <= buf.n)
        return;
    if (buf.p)
        MSVS_HIP(hipStreamSynchronize(stream)); // earlier work on this stream may still read the old arena
    size_t cap = bytes + bytes / 2;
    buf.alloc(cap);
}

hipStream_t thread_stream()
{
    static thread_local std::map<int, hipStream_t> streams;
    int dev = 0;
    MSVS_HIP(hipGetDevice(&dev));
    auto it = streams.find(dev);
    if (it != streams.end())
        return it->second;
    hipStream_t s = nullptr;
    MSVS_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    streams[dev] = s;
    return s;
}

namespace
{
/// Every arena of the calling host thread, keyed by (device, stream).
struct ThreadArenas
{
    std::map<std::pair<int, hipStream_t>, Scratch> scratch, staging, aux, shard, view, route;
};
thread_local ThreadArenas t_arenas;

Scratch & arena_in(std::map<std::pair<int, hipStream_t>, Scratch> & m, hipStream_t stream)
{
    int dev = 0;
    MSVS_HIP(hipGetDevice(&dev));
    return m[{dev, stream}];
}
}

Scratch & scratch_for(hipStream_t stream) { return arena_in(t_arenas.scratch, stream); }
Scratch & aux_for(hipStream_t stream) { return arena_in(t_arenas.aux, stream); }
Scratch & shard_for(hipStream_t stream) { return arena_in(t_arenas.shard, stream); }
Scratch & staging_for(hipStream_t stream) { return arena_in(t_arenas.staging, stream); }
Scratch & view_for(hipStream_t stream) { return arena_in(t_arenas.view, stream); }
Scratch & route_for(hipStream_t stream) { return arena_in(t_arenas.route, stream); }

size_t release_thread_arenas()
{
    size_t freed = 0;
    (void)hipDeviceSynchronize(); // nothing enqueued may still use them
    for (auto * m : {&t_arenas.scratch, &t_arenas.staging, &t_arenas.aux, &t_arenas.shard, &t_arenas.view, &t_arenas.route})
    {
        for (auto & kv : *m)
            freed += kv.second.buf.n;
        m->clear();
    }
    return freed;
}

typedef uint32_t fetch_u32x4 __attribute__((ext_vector_type(4)));
static __global__ void fetch_pinned_kernel(fetch_u32x4 * dst, const fetch_u32x4 * src, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = __builtin_nontemporal_load(&src[i]);
}

void fetch_from_pinned(void * d_dst, const void * h_pinned_src, size_t bytes, hipStream_t stream)
{
    if (bytes == 0)
        return;
    const size_t n16 = ceil_div(bytes, (size_t)16);
    if (options().pinned_fetch == 0 || bytes > (size_t)options().pinned_fetch_max || ((uintptr_t)d_dst & 15) || ((uintptr_t)h_pinned_src & 15))
    {
        MSVS_HIP(hipMemcpyAsync(d_dst, h_pinned_src, bytes, hipMemcpyHostToDevice, stream));
        return;
    }
    const unsigned grid = (unsigned)std::min<size_t>(64, ceil_div(n16, (size_t)256));
    hipLaunchKernelGGL(fetch_pinned_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<fetch_u32x4 *>(d_dst),
                       reinterpret_cast<const fetch_u32x4 *>(h_pinned_src), n16);
    MSVS_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ profiling

namespace
{
struct Sample
{
    const char * name;
    hipEvent_t start, stop;
};
std::mutex g_prof_mu;
std::vector<Sample> g_samples;
std::atomic<bool> g_prof_on{false};
}

void profile_enable(bool on) { g_prof_on.store(on); }

ProfileScope::ProfileScope(const char * n, hipStream_t s) : name(n), stream(s)
{
    if (!g_prof_on.load(std::memory_order_relaxed))
        return;
    MSVS_HIP(hipEventCreate(&start));
    MSVS_HIP(hipEventRecord(start, stream));
}

ProfileScope::~ProfileScope()
{
    if (!start)
        return;
    hipEvent_t stop = nullptr;
    if (hipEventCreate(&stop) != hipSuccess || hipEventRecord(stop, stream) != hipSuccess)
        return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_samples.push_back({name, start, stop});
}

void profile_get(const char * name, uint64_t * calls, double * total_ms)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    uint64_t c = 0;
    double t = 0;
    for (auto & s : g_samples)
    {
        if (strcmp(s.name, name) != 0)
            continue;
        MSVS_HIP(hipEventSynchronize(s.stop));
        float ms = 0;
        MSVS_HIP(hipEventElapsedTime(&ms, s.start, s.stop));
        c++;
        t += ms;
    }
    if (calls)
        *calls = c;
    if (total_ms)
        *total_ms = t;
}

void profile_reset()
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto & s : g_samples)
    {
        (void)hipEventDestroy(s.start);
        (void)hipEventDestroy(s.stop);
    }
    g_samples.clear();
}

FlatPlan plan_flat(size_t n_rows, size_t nq, uint32_t ld4, uint32_t k)
{
    FlatPlan p;
    p.T = nq <= 1 ? 1 : (nq <= 2 ? 2 : (nq <= 4 ? 4 : 8));
    // the query tile and the block's merge lists live in LDS: halve the tile until they fit the default dynamic LDS
    // (64 KiB; one query of 8192 dimensions with k = 256 takes 42 KiB)
    while (p.T > 1 && scan_lds_bytes(p.T, ld4, k) > SCAN_LDS_BUDGET)
        p.T /= 2;
    p.n_qtiles = (uint32_t)ceil_div(nq ? nq : 1, p.T);
    // enough blocks to fill 256 CUs a few times over, but at least 16 rows (one block iteration) each
    size_t want = 2048 / p.n_qtiles;
    if (want < 1)
        want = 1;
    size_t max_blocks = ceil_div(n_rows ? n_rows : 1, 16);
    size_t nb = want < max_blocks ? want : max_blocks;
    size_t rpb = round_up(ceil_div(n_rows ? n_rows : 1, nb), 16);
    p.rows_per_block = (uint32_t)rpb;
    p.n_blocks = (uint32_t)ceil_div(n_rows ? n_rows : 1, rpb);
    return p;
}

template <int METRIC, int T>
static void flat_dispatch_r(const FlatPlan & plan, const ScanParams & a, hipStream_t stream)
{
    dim3 grid(plan.n_blocks, plan.n_qtiles);
    size_t lds = scan_lds_bytes(T, a.ld4, a.k);
    switch (r_for_k(a.k))
    {
        case 1:
            hipLaunchKernelGGL((flat_scan_kernel<METRIC, T, 1>), grid, dim3(BLOCK), lds, stream, a);
            break;
        case 2:
            hipLaunchKernelGGL((flat_scan_kernel<METRIC, T, 2>), grid, dim3(BLOCK), lds, stream, a);
            break;
        default:
            hipLaunchKernelGGL((flat_scan_kernel<METRIC, T, 4>), grid, dim3(BLOCK), lds, stream, a);
            break;
    }
}

template <int METRIC>
static void flat_dispatch_t(const FlatPlan & plan, const ScanParams & a, hipStream_t stream)
{
    switch (plan.T)
    {
        case 1:
            flat_dispatch_r<METRIC, 1>(plan, a, stream);
            break;
        case 2:
            flat_dispatch_r<METRIC, 2>(plan, a, stream);
            break;
        case 4:
            flat_dispatch_r<METRIC, 4>(plan, a, stream);
            break;
        default:
            flat_dispatch_r<METRIC, 8>(plan, a, stream);
            break;
    }
}

void launch_flat_scan(int metric, const FlatPlan & plan, ScanParams a, hipStream_t stream)
{
    a.rows_per_block = plan.rows_per_block;
    a.n_blocks = plan.n_blocks;
    if (scan_lds_bytes(plan.T, a.ld4, a.k) > 160 * 1024)
        fail(MSVS_ERR_INVALID_ARGUMENT, "dimension %u too large for the LDS query tile", a.ld4 * 4);
    ProfileScope prof("flat_scan", stream);
    if (metric == M_IP)
        flat_dispatch_t<M_IP>(plan, a, stream);
    else
        flat_dispatch_t<M_L2>(plan, a, stream);
    MSVS_HIP(hipGetLastError());
}

template <int METRIC>
static void merge_dispatch(const MergeParams & a, uint32_t nq, hipStream_t stream)
{
    size_t lds = merge_lds_bytes(a.k, 0);
    switch (r_for_k(a.k))
    {
        case 1:
            hipLaunchKernelGGL((merge_kernel<METRIC, 1>), dim3(nq), dim3(BLOCK), lds, stream, a);
            break;
        case 2:
            hipLaunchKernelGGL((merge_kernel<METRIC, 2>), dim3(nq), dim3(BLOCK), lds, stream, a);
            break;
        default:
            hipLaunchKernelGGL((merge_kernel<METRIC, 4>), dim3(nq), dim3(BLOCK), lds, stream, a);
            break;
    }
}

void launch_merge(int metric, MergeParams a, uint32_t nq, hipStream_t stream)
{
    if (nq == 0)
        return;
    ProfileScope prof("merge", stream);
    if (metric == M_IP)
        merge_dispatch<M_IP>(a, nq, stream);
    else
        merge_dispatch<M_L2>(a, nq, stream);
    MSVS_HIP(hipGetLastError());
}

template <int METRIC>
static void ivf_dispatch(const ScanParams & a, hipStream_t stream)
{
    dim3 grid(a.seg_max, a.nprobe, a.nq);
    size_t lds = scan_lds_bytes(1, a.ld4, a.k);
    switch (r_for_k(a.k))
    {
        case 1:
            hipLaunchKernelGGL((ivf_scan_kernel<METRIC, 1>), grid, dim3(BLOCK), lds, stream, a);
            break;
        case 2:
            hipLaunchKernelGGL((ivf_scan_kernel<METRIC, 2>), grid, dim3(BLOCK), lds, stream, a);
            break;
        default:
            hipLaunchKernelGGL((ivf_scan_kernel<METRIC, 4>), grid, dim3(BLOCK), lds, stream, a);
            break;
    }
}

void launch_ivf_scan(int metric, ScanParams a, hipStream_t stream)
{
    if (a.nq == 0 || a.nprobe == 0 || a.seg_max == 0)
        return;
    ProfileScope prof("ivf_scan", stream);
    if (metric == M_IP)
        ivf_dispatch<M_IP>(a, stream);
    else
        ivf_dispatch<M_L2>(a, stream);
    MSVS_HIP(hipGetLastError());
}

bool ivf_plan_fused(const IvfPlanParams & p)
{
    return p.n_pairs <= PLAN_FUSED_PAIRS && p.nlist <= PLAN_LDS_LISTS && options().plan_lds != 0 && options().plan_fused != 0;
}

void launch_ivf_plan(const IvfPlanParams & p, hipStream_t stream)
{
    if (p.n_pairs == 0)
        return;
    ProfileScope prof("ivf_plan", stream);
    if (ivf_plan_fused(p))
        hipLaunchKernelGGL(ivf_plan_scan_kernel<true>, dim3(1), dim3(1024), 0, stream, p);
    else if (p.nlist <= PLAN_LDS_LISTS && options().plan_lds != 0)
    {
        unsigned g = (unsigned)ceil_div(p.n_pairs, PLAN_CHUNK);
        hipLaunchKernelGGL(ivf_hist_lds_kernel, dim3(g), dim3(256), 0, stream, p);
        hipLaunchKernelGGL(ivf_plan_scan_kernel<false>, dim3(1), dim3(1024), 0, stream, p);
        hipLaunchKernelGGL(ivf_scatter_lds_kernel, dim3(g), dim3(256), 0, stream, p);
    }
    else
    {
        unsigned g = (unsigned)ceil_div(p.n_pairs, 256);
        hipLaunchKernelGGL(ivf_hist_kernel, dim3(g), dim3(256), 0, stream, p);
        hipLaunchKernelGGL(ivf_plan_scan_kernel<false>, dim3(1), dim3(1024), 0, stream, p);
        hipLaunchKernelGGL(ivf_scatter_kernel, dim3(g), dim3(256), 0, stream, p);
    }
    MSVS_HIP(hipGetLastError());
}

void launch_ivf_plan_rescan(const IvfPlanParams & p, hipStream_t stream)
{
    if (p.n_pairs == 0)
        return;
    hipLaunchKernelGGL(ivf_plan_scan_kernel<false>, dim3(1), dim3(1024), 0, stream, p);
    MSVS_HIP(hipGetLastError());
}

template <int METRIC, int T>
static void ivf_batched_dispatch_r(uint32_t grid, const ScanParams & a, hipStream_t stream)
{
    size_t lds = scan_lds_bytes(T, a.ld4, a.k);
    switch (r_for_k(a.k))
    {
        case 1:
            hipLaunchKernelGGL((ivf_batched_scan_kernel<METRIC, T, 1>), dim3(grid), dim3(BLOCK), lds, stream, a);
            break;
        case 2:
            hipLaunchKernelGGL((ivf_batched_scan_kernel<METRIC, T, 2>), dim3(grid), dim3(BLOCK), lds, stream, a);
            break;
        default:
            hipLaunchKernelGGL((ivf_batched_scan_kernel<METRIC, T, 4>), dim3(grid), dim3(BLOCK), lds, stream, a);
            break;
    }
}

template <int METRIC>
static void ivf_batched_dispatch_t(uint32_t T, uint32_t grid, const ScanParams & a, hipStream_t stream)
{
    switch (T)
    {
        case 2:
            ivf_batched_dispatch_r<METRIC, 2>(grid, a, stream);
            break;
        case 4:
            ivf_batched_dispatch_r<METRIC, 4>(grid, a, stream);
            break;
        default:
            ivf_batched_dispatch_r<METRIC, 8>(grid, a, stream);
            break;
    }
}

void launch_ivf_batched_scan(int metric, uint32_t T, uint32_t grid, ScanParams a, hipStream_t stream)
{
    if (grid == 0)
        return;
    if (scan_lds_bytes(T, a.ld4, a.k) > 160 * 1024)
        fail(MSVS_ERR_INVALID_ARGUMENT, "dimension %u too large for the LDS query tile", a.ld4 * 4);
    ProfileScope prof("ivf_scan", stream);
    if (metric == M_IP)
        ivf_batched_dispatch_t<M_IP>(T, grid, a, stream);
    else
        ivf_batched_dispatch_t<M_L2>(T, grid, a, stream);
    MSVS_HIP(hipGetLastError());
}

template <int METRIC>
static void ivf_merge_dispatch(const IvfMergeParams & a, uint32_t nq, hipStream_t stream)
{
    size_t lds = merge_lds_bytes(a.k, a.nprobe);
    switch (r_for_k(a.k))
    {
        case 1:
            hipLaunchKernelGGL((ivf_merge_kernel<METRIC, 1>), dim3(nq), dim3(BLOCK), lds, stream, a);
            break;
        case 2:
            hipLaunchKernelGGL((ivf_merge_kernel<METRIC, 2>), dim3(nq), dim3(BLOCK), lds, stream, a);
            break;
        default:
            hipLaunchKernelGGL((ivf_merge_kernel<METRIC, 4>), dim3(nq), dim3(BLOCK), lds, stream, a);
            break;
    }
}

void launch_ivf_merge(int metric, IvfMergeParams a, uint32_t nq, hipStream_t stream)
{
    if (nq == 0)
        return;
    ProfileScope prof("merge", stream);
    if (metric == M_IP)
        ivf_merge_dispatch<M_IP>(a, nq, stream);
    else
        ivf_merge_dispatch<M_L2>(a, nq, stream);
    MSVS_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ matrix-core pass

void launch_row_sqnorm(const float * X, float * out, size_t n, uint32_t ld4, uint32_t * max_bits, hipStream_t stream)
{
    if (n == 0)
        return;
    hipLaunchKernelGGL(row_sqnorm16_kernel, dim3((unsigned)ceil_div(n, 16)), dim3(BLOCK), 0, stream,
                       reinterpret_cast<const float4 *>(X), out, n, ld4, max_bits);
    MSVS_HIP(hipGetLastError());
}

void launch_split_queries(const float * Q, uint32_t nq, uint32_t ld4, void * out, hipStream_t stream)
{
    if (nq == 0)
        return;
    const uint32_t nk = (ld4 + 7) / 8;
    const size_t n = (size_t)nq * nk * 8;
    hipLaunchKernelGGL(split_queries_kernel, dim3((unsigned)ceil_div(n, (size_t)256)), dim3(256), 0, stream,
                       reinterpret_cast<const float4 *>(Q), nq, ld4, nk, reinterpret_cast<uint4 *>(out));
    MSVS_HIP(hipGetLastError());
}

template <int METRIC, int NQG>
static void mfma_scan_dispatch(bool main_phase, uint32_t grid, const ScanParams & a, hipStream_t stream)
{
    if (main_phase)
        hipLaunchKernelGGL((ivf_mfma_scan_big_kernel<METRIC, NQG, 1>), dim3(grid), dim3(NQG * BLOCK), 0, stream, a);
    else
        hipLaunchKernelGGL((ivf_mfma_scan_big_kernel<METRIC, NQG, 0>), dim3(grid), dim3(NQG * BLOCK), 0, stream, a);
}

void launch_ivf_mfma_scan(int metric, uint32_t nqg, uint32_t grid, ScanParams a, hipStream_t stream,
                          const char * profile_name, bool main_phase)
{
    if (grid == 0)
        return;
    ProfileScope prof(profile_name, stream);
    if (nqg == 2)
    {
        if (metric == M_IP)
            mfma_scan_dispatch<M_IP, 2>(main_phase, grid, a, stream);
        else
            mfma_scan_dispatch<M_L2, 2>(main_phase, grid, a, stream);
    }
    else if (metric == M_IP)
        mfma_scan_dispatch<M_IP, 1>(main_phase, grid, a, stream);
    else
        mfma_scan_dispatch<M_L2, 1>(main_phase, grid, a, stream);
    MSVS_HIP(hipGetLastError());
}

void launch_single_list_plan(uint32_t nq, uint32_t row_begin, uint32_t row_end, uint32_t rows_per_block, uint32_t tq,
                             uint32_t * pairs, int32_t * probes0, int64_t * list_off, uint32_t * pair_off,
                             uint32_t * work_off, hipStream_t stream)
{
    hipLaunchKernelGGL(single_list_plan_kernel, dim3((unsigned)ceil_div(std::max<uint32_t>(nq, 1), 256u)), dim3(256), 0,
                       stream, nq, row_begin, row_end, rows_per_block, tq, pairs, probes0, list_off, pair_off, work_off);
    MSVS_HIP(hipGetLastError());
}

void launch_sample_cut(const uint64_t * cand, uint32_t kc, uint32_t m, uint32_t nq, uint32_t * qthr, hipStream_t stream)
{
    if (nq == 0)
        return;
    hipLaunchKernelGGL(sample_cut_kernel, dim3((unsigned)ceil_div(nq, 256u)), dim3(256), 0, stream, cand, kc, m, nq,
                       qthr);
    MSVS_HIP(hipGetLastError());
}

void launch_cand_select(const uint64_t * buf, const uint32_t * qcnt, const uint32_t * qthr, uint32_t cap, uint32_t nq,
                        uint32_t kc, uint64_t * out, uint64_t * bound, hipStream_t stream)
{
    if (nq == 0)
        return;
    ProfileScope prof("merge", stream);
    if (cap <= CAND_SELECT_WAVE_CAP && kc <= 64 && options().wave_select != 0)
        hipLaunchKernelGGL(cand_select_wave_kernel, dim3((nq + 3) / 4), dim3(BLOCK), 0, stream, buf, qcnt, qthr, cap, nq, kc, out,
                           bound, options().wave_select == 3 ? 0 : 1, options().rerank_early != 0 ? 1 : 0);
    else if (kc <= 64 && !(nq <= 16 && options().wave_select != 0)) // (a few queries with ~2000 candidates each: the block-wide radix select, 19 -> ~6 us)
        hipLaunchKernelGGL(cand_select_kernel<1>, dim3(nq), dim3(BLOCK), 0, stream, buf, qcnt, qthr, cap, nq, kc, out, bound);
    else if (options().wave_select != 0)
        hipLaunchKernelGGL(cand_select_block_kernel, dim3(nq), dim3(BLOCK), 0, stream, buf, qcnt, qthr, cap, nq, kc, out, bound);
    else
        hipLaunchKernelGGL(cand_select_kernel<4>, dim3(nq), dim3(BLOCK), 0, stream, buf, qcnt, qthr, cap, nq, kc, out, bound);
    MSVS_HIP(hipGetLastError());
}

void launch_ivf_rerank(int metric, RerankParams a, uint32_t nq, hipStream_t stream)
{
    if (nq == 0)
        return;
    if (a.kc > 256 || a.k > a.kc)
        fail(MSVS_ERR_INVALID_ARGUMENT, "re-rank of %u candidates for k = %u", a.kc, a.k);
    const size_t lds = (size_t)a.ld4 * 16 + (a.kc <= 64 ? 64 : 256) * 8
        + (a.fuse_second ? (2 * RA_KMAX + RA_CHUNK) * 8 + RA_CHUNK * 4 + 16 : 0); // (+ the second chance's arrays: rerank_all_query)
    if (lds > 64 * 1024)
        fail(MSVS_ERR_INVALID_ARGUMENT, "dimension %u too large for the re-rank block", a.ld4 * 4);
    ProfileScope prof("rerank", stream);
    const bool ip = metric == M_IP;
    if (a.kc > 64) // 256 candidates (40 < k <= 128)
    {
        if (ip)
            hipLaunchKernelGGL((ivf_rerank_kernel<M_IP, 16, 4>), dim3(nq), dim3(256), lds, stream, a);
        else
            hipLaunchKernelGGL((ivf_rerank_kernel<M_L2, 16, 4>), dim3(nq), dim3(256), lds, stream, a);
    }
    else if (options().rerank_groups != 32)
    {
        if (ip)
            hipLaunchKernelGGL((ivf_rerank_kernel<M_IP, 16, 1>), dim3(nq), dim3(256), lds, stream, a);
        else
            hipLaunchKernelGGL((ivf_rerank_kernel<M_L2, 16, 1>), dim3(nq), dim3(256), lds, stream, a);
    }
    else if (ip)
        hipLaunchKernelGGL((ivf_rerank_kernel<M_IP, 32, 1>), dim3(nq), dim3(512), lds, stream, a);
    else
        hipLaunchKernelGGL((ivf_rerank_kernel<M_L2, 32, 1>), dim3(nq), dim3(512), lds, stream, a);
    MSVS_HIP(hipGetLastError());
}

void launch_ivf_rerank_all(int metric, const RerankParams & a, const RerankAllParams & b, uint32_t nq, hipStream_t stream)
{
    if (nq == 0)
        return;
    if (a.k > RA_KMAX)
        fail(MSVS_ERR_INVALID_ARGUMENT, "second-chance re-rank for k = %u", a.k);
    const size_t lds = (size_t)a.ld4 * 16 + (2 * RA_KMAX + RA_CHUNK) * 8 + RA_CHUNK * 4 + 16;
    ProfileScope prof("rerank", stream);
    const uint32_t grid = std::min<uint32_t>(nq, 2048); // usually nobody is on the list: every block exits at once
    if (metric == M_IP)
        hipLaunchKernelGGL((ivf_rerank_all_kernel<M_IP>), dim3(grid), dim3(256), lds, stream, a, b);
    else
        hipLaunchKernelGGL((ivf_rerank_all_kernel<M_L2>), dim3(grid), dim3(256), lds, stream, a, b);
    MSVS_HIP(hipGetLastError());
}

template <int METRIC>
static void ivf_subset_dispatch(const ScanParams & a, uint32_t z, hipStream_t stream)
{
    dim3 grid(a.seg_max, a.nprobe, z);
    size_t lds = scan_lds_bytes(1, a.ld4, a.k);
    switch (r_for_k(a.k))
    {
        case 1:
            hipLaunchKernelGGL((ivf_scan_subset_kernel<METRIC, 1>), grid, dim3(BLOCK), lds, stream, a);
            break;
        case 2:
            hipLaunchKernelGGL((ivf_scan_subset_kernel<METRIC, 2>), grid, dim3(BLOCK), lds, stream, a);
            break;
        default:
            hipLaunchKernelGGL((ivf_scan_subset_kernel<METRIC, 4>), grid, dim3(BLOCK), lds, stream, a);
            break;
    }
}

void launch_ivf_scan_subset(int metric, ScanParams a, uint32_t z, hipStream_t stream)
{
    if (z == 0 || a.nprobe == 0 || a.seg_max == 0)
        return;
    ProfileScope prof("fallback_scan", stream);
    if (metric == M_IP)
        ivf_subset_dispatch<M_IP>(a, z, stream);
    else
        ivf_subset_dispatch<M_L2>(a, z, stream);
    MSVS_HIP(hipGetLastError());
}

template <int METRIC>
static void ivf_merge_subset_dispatch(const IvfMergeParams & a, uint32_t blocks, hipStream_t stream)
{
    size_t lds = merge_lds_bytes(a.k, a.nprobe);
    switch (r_for_k(a.k))
    {
        case 1:
            hipLaunchKernelGGL((ivf_merge_subset_kernel<METRIC, 1>), dim3(blocks), dim3(BLOCK), lds, stream, a);
            break;
        case 2:
            hipLaunchKernelGGL((ivf_merge_subset_kernel<METRIC, 2>), dim3(blocks), dim3(BLOCK), lds, stream, a);
            break;
        default:
            hipLaunchKernelGGL((ivf_merge_subset_kernel<METRIC, 4>), dim3(blocks), dim3(BLOCK), lds, stream, a);
            break;
    }
}

void launch_ivf_merge_subset(int metric, IvfMergeParams a, uint32_t blocks, hipStream_t stream)
{
    if (blocks == 0)
        return;
    ProfileScope prof("fallback_merge", stream);
    if (metric == M_IP)
        ivf_merge_subset_dispatch<M_IP>(a, blocks, stream);
    else
        ivf_merge_subset_dispatch<M_L2>(a, blocks, stream);
    MSVS_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ options

namespace
{
struct OptionField
{
    const char * name;
    double Options::*field;
};
const OptionField g_option_fields[] = {
    {"ivf_pass", &Options::ivf_pass},       {"ivf_mfma", &Options::ivf_pass} /* round-1 name */,
    {"ivf_h16", &Options::ivf_h16},         {"coarse_mfma", &Options::coarse_mfma},
    {"coarse_h16", &Options::coarse_h16},   {"wave_select", &Options::wave_select},
    {"plan_lds", &Options::plan_lds},       {"fb_segs", &Options::fb_segs},
    {"h16_kc", &Options::h16_kc},           {"coarse_kc", &Options::coarse_kc},
    {"bm25_fine_sample", &Options::bm25_fine_sample}, {"lat_select", &Options::lat_select},
    {"rerank_stats", &Options::rerank_stats},   {"rerank_early", &Options::rerank_early},
    {"rerank_groups", &Options::rerank_groups}, {"combine", &Options::combine},           {"h16_k128", &Options::h16_k128},
    {"h16_target", &Options::h16_target},       
    {"coarse_h16_min_q", &Options::coarse_h16_min_q}, 
    {"combine_batches", &Options::combine_batches},
    {"flat_mfma", &Options::flat_mfma},     {"ivf_nqg", &Options::ivf_nqg},
    {"ivf_rpb", &Options::ivf_rpb},         {"ivf_grid", &Options::ivf_grid},
    {"ivf_t", &Options::ivf_t},             {"ivf_xcd", &Options::ivf_xcd},
    {"cand_cap", &Options::cand_cap},       {"ivf_eps_scale", &Options::ivf_eps_scale},
               {"h16_grid", &Options::h16_grid},
    {"h16_min_pairs", &Options::h16_min_pairs}, {"h16_ncb", &Options::h16_ncb},
    {"h16_nocut", &Options::h16_nocut},     {"fb_cap", &Options::fb_cap},           {"rerank_second", &Options::rerank_second}, {"rerank_fused", &Options::rerank_fused},
    {"lat_path", &Options::lat_path},         {"filter_compact_below", &Options::filter_compact_below},
    {"bm25_wave", &Options::bm25_wave},     {"rerank_hint", &Options::rerank_hint}, {"coarse_band", &Options::coarse_band}, {"coarse_tail", &Options::coarse_tail}, {"merge_small", &Options::merge_small}, {"route_streams", &Options::route_streams}, {"coarse_slow_inline", &Options::coarse_slow_inline}, {"coarse_slow_window", &Options::coarse_slow_window}, {"coarse_gemm_tq", &Options::coarse_gemm_tq}, {"coarse_gemm_tc", &Options::coarse_gemm_tc}, {"h16_prune", &Options::h16_prune}, {"h16_feedback", &Options::h16_feedback}, {"h16_group_appends", &Options::h16_group_appends}, {"h16_preprune", &Options::h16_preprune}, {"lat_prune", &Options::lat_prune}, {"lat_items", &Options::lat_items}, {"lat_hint", &Options::lat_hint},     {"bm25_posting", &Options::bm25_posting}, {"bm25_sub_docs", &Options::bm25_sub_docs}, {"bm25_dbg", &Options::bm25_dbg},
    {"h16_rho", &Options::h16_rho}, {"h16_segs", &Options::h16_segs}, {"h16_stamps", &Options::h16_stamps},   {"flat_h16", &Options::flat_h16},     {"flat_segb", &Options::flat_segb}, {"flat_rot", &Options::flat_rot}, {"flat_lazy_flush", &Options::flat_lazy_flush},   {"flat_ncb", &Options::flat_ncb},
    {"bm25_emit", &Options::bm25_emit},     {"bm25_cand_cap", &Options::bm25_cand_cap},
    {"flat_few", &Options::flat_few},
    {"bm25_rec", &Options::bm25_rec},       {"bm25_slots", &Options::bm25_slots},
    {"bm25_cutk", &Options::bm25_cutk},     {"bm25_bounds8", &Options::bm25_bounds8}, {"bm25_tables_ride", &Options::bm25_tables_ride}, {"bm25_lean", &Options::bm25_lean}, {"bm25_skip", &Options::bm25_skip}, {"bm25_select2", &Options::bm25_select2}, {"bm25_items_per_wave", &Options::bm25_items_per_wave}, {"flat_host_signal", &Options::flat_host_signal}, {"flat_sample_few", &Options::flat_sample_few}, {"plan_fused", &Options::plan_fused}, {"coarse_dense", &Options::coarse_dense}, {"host_signal_batch", &Options::host_signal_batch}, {"combine_spin", &Options::combine_spin}, {"host_pinned", &Options::host_pinned}, {"coarse_few", &Options::coarse_few}, {"pinned_fetch", &Options::pinned_fetch}, {"pinned_fetch_max", &Options::pinned_fetch_max},
    {"route_self_rccl", &Options::route_self_rccl},
};
Options g_options;
std::once_flag g_options_once;
std::mutex g_options_mu;

void options_from_env()
{
    for (const auto & f : g_option_fields)
    {
        std::string env = "MSVS_";
        for (const char * c = f.name; *c; ++c)
            env.push_back((char)toupper((unsigned char)*c));
        if (const char * v = getenv(env.c_str()))
            if (*v)
                g_options.*(f.field) = atof(v);
    }
}
}

const Options & options()
{
    std::call_once(g_options_once, options_from_env);
    return g_options;
}

bool set_option(const char * name, const char * value)
{
    (void)options();
    if (!name)
        return false;
    std::string n;
    for (const char * c = name; *c; ++c)
        n.push_back((char)tolower((unsigned char)*c));
    if (n.rfind("msvs_", 0) == 0)
        n = n.substr(5);
    std::lock_guard<std::mutex> lk(g_options_mu);
    for (const auto & f : g_option_fields)
        if (n == f.name)
        {
            g_options.*(f.field) = value && *value ? atof(value) : Options().*(f.field);
            return true;
        }
    return false;
}

// ------------------------------------------------------------------------------------------ params

std::map<std::string, std::string> parse_params(const char * s)
{
    std::map<std::string, std::string> m;
    if (!s)
        return m;
    std::string cur_k, cur_v;
    bool in_v = false;
    auto flush = [&]() {
        auto trim = [](std::string x) {
            size_t a = x.find_first_not_of(" \t\r\n\"'{}");
            size_t b = x.find_last_not_of(" \t\r\n\"'{}");
            return a == std::string::npos ? std::string() : x.substr(a, b - a + 1);
        };
        std::string k = trim(cur_k), v = trim(cur_v);
        if (!k.empty())
            m[k] = v;
        cur_k.clear();
        cur_v.clear();
        in_v = false;
    };
    for (const char * p = s; *p; ++p)
    {
        char c = *p;
        if (c == ',' || c == ';')
            flush();
        else if ((c == '=' || c == ':') && !in_v)
            in_v = true;
        else
            (in_v ? cur_v : cur_k).push_back(c);
    }
    flush();
    return m;
}

long param_int(const std::map<std::string, std::string> & m, const char * key, long dflt)
{
    auto it = m.find(key);
    if (it == m.end())
        return dflt;
    char * end = nullptr;
    long v = strtol(it->second.c_str(), &end, 10);
    if (end == it->second.c_str() || *end != '\0')
        fail(MSVS_ERR_INVALID_ARGUMENT, "parameter `%s` value should be int, got `%s`", key, it->second.c_str());
    return v;
}

}
