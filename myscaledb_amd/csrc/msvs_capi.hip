// msvs_capi.hip -- the C-ABI of libmsvs.so (include/msvs.h): seam A2 (brute force), seam A1 (index object),
// top-k merge.  Everything computes on the GPU; there is no CPU fallback.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <list>
#include <map>
#include <string>
#include <vector>
#include <cmath>
#include <mutex>
#include <cstdio>
#include <memory>
#include <numeric>
#include <random>

#include "device_ops.hpp"
#include "index_internal.hpp"
#include "ivf_build_kernels.hpp"
#include "h16_scan_kernels.hpp"
#include "io_stream.hpp"
#include "latency_kernels.hpp"
#include "filter_kernels.hpp"

namespace msvs
{
const char * last_error_cstr();



/// Copy n rows of d floats (host or device) into a device buffer with row stride ld (zero padded).
void upload_rows(float * dst, const float * src, size_t n, uint32_t d, uint32_t ld, int mem, hipStream_t stream)
{
    if (n == 0)
        return;
    if (ld != d)
        MSVS_HIP(hipMemsetAsync(dst, 0, n * ld * sizeof(float), stream));
    MSVS_HIP(hipMemcpy2DAsync(dst, ld * sizeof(float), src, d * sizeof(float), d * sizeof(float), n,
                              mem == MSVS_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
}




/// Exhaustive top-k of device queries (nq x ld) against device rows (n x ld) -> device ids/dis.
/// `scr` must have been reserved by the caller for flat_scratch_bytes().
size_t flat_scratch_bytes(size_t n, size_t nq, uint32_t k, uint32_t ld)
{
    FlatPlan p = plan_flat(n, nq, ld / 4, k);
    return flat_partial_keys(p, nq, k) * 8 + 1024;
}


void flat_search_device(Scratch & scr, int metric, const float * d_rows, const uint32_t * d_row_ids, size_t n, uint32_t ld,
                        const float * d_q, size_t nq, uint32_t k, const uint64_t * d_alive, size_t nbits, MergeParams out,
                        hipStream_t stream, const SearchView * view)
{
    if (view)
        n = std::max<size_t>(1, std::min(n, view->n_upper));
    FlatPlan p = plan_flat(n, nq, ld / 4, k);
    uint64_t * partial = scr.take<uint64_t>(flat_partial_keys(p, nq, k));
    ScanParams a{};
    a.Y = reinterpret_cast<const float4 *>(d_rows);
    a.ids = d_row_ids;
    a.alive = d_alive;
    a.nbits = (uint32_t)std::min<size_t>(nbits, 0xffffffffu);
    a.Q = reinterpret_cast<const float4 *>(d_q);
    a.partial = partial;
    a.ld4 = ld / 4;
    a.k = k;
    a.nq = (uint32_t)nq;
    a.n_rows = (uint32_t)n;
    if (view)
    {
        a.rowmap = view->rowmap;
        a.n_rows_dev = view->n_rows;
    }
    launch_flat_scan(scan_metric(metric), p, a, stream);
    out.partial = partial;
    out.n_lists = p.n_blocks;
    out.k = k;
    launch_merge(scan_metric(metric), out, (uint32_t)nq, stream);
}

}

using namespace msvs;

// =========================================================================================== basics

extern "C" const char * msvs_last_error(void) { return last_error_cstr(); }
extern "C" const char * msvs_version(void) { return "msvs 0.1 (gfx950)"; }

extern "C" int msvs_device_count(int * count)
{
    return guarded([&] {
        if (!count)
            fail(MSVS_ERR_INVALID_ARGUMENT, "count is null");
        MSVS_HIP(hipGetDeviceCount(count));
    });
}

extern "C" int msvs_set_device(int ordinal)
{
    return guarded([&] { MSVS_HIP(hipSetDevice(ordinal)); });
}

extern "C" int msvs_device_synchronize(void)
{
    return guarded([&] { MSVS_HIP(hipDeviceSynchronize()); });
}

// =========================================================================================== seam A1

static __global__ void and_bits_kernel(const uint64_t * a, size_t na, const uint64_t * b, size_t nb, uint64_t * out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = (i < na ? a[i] : 0ull) & (i < nb ? b[i] : 0ull);
}

/// getRealBitmap (src/VectorIndex/Utils/VIUtils.cpp:479-498): a filter over the rows of the merged (decoupled) part ->
/// a filter over the labels of this source part's index.  out must be zeroed.
static __global__ void real_bitmap_kernel(const uint64_t * filter_new, size_t nbits_new, const uint64_t * inv_ids,
                                          const uint8_t * inv_src, uint32_t own_id, size_t n_new, size_t total_vec,
                                          unsigned long long * out)
{
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_new || r >= nbits_new)
        return;
    if (((filter_new[r >> 6] >> (r & 63)) & 1) && inv_src[r] == own_id)
    {
        const uint64_t old = inv_ids[r];
        if (old < total_vec)
            atomicOr(out + (old >> 6), 1ull << (old & 63));
    }
}

/// transferToNewRowIds (VIWithDataPart.cpp:56-67): label -> row_ids_map[label] for every non-empty result slot.
static __global__ void remap_ids_kernel(int64_t * ids, size_t n, const uint64_t * map, size_t map_n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && ids[i] >= 0 && (uint64_t)ids[i] < map_n)
        ids[i] = (int64_t)map[ids[i]];
}

namespace msvs
{
/// The filter a search really runs with: (per-search filter, converted to label space for a decoupled part) AND the resident
/// delete bitmap.  Returns the device pointer (nullptr = no filter) and its valid bits; scratch from aux_for(stream).
const uint64_t * effective_filter(const msvs_index & ix, const msvs_index::Meta * meta, const uint64_t * d_alive,
                                  size_t nbits, size_t * eff_nbits, hipStream_t stream)
{
    *eff_nbits = nbits;
    if (!meta)
        return d_alive;
    const bool convert = d_alive && meta->inv_n;
    const bool deleted = meta->delete_alive.p != nullptr;
    if (!convert && !deleted)
        return d_alive;
    const size_t total = (size_t)ix.max_id + 1;
    const size_t words = ceil_div(total, (size_t)64) + 1;
    Scratch & aux = aux_for(stream);
    aux.reserve(2 * words * 8 + 1024, stream);
    const uint64_t * cur = d_alive;
    size_t cur_bits = nbits;
    if (convert)
    {
        uint64_t * real = aux.take<uint64_t>(words);
        MSVS_HIP(hipMemsetAsync(real, 0, words * 8, stream));
        const size_t n = std::min(meta->inv_n, nbits);
        if (n)
            hipLaunchKernelGGL(real_bitmap_kernel, dim3((unsigned)ceil_div(n, (size_t)256)), dim3(256), 0, stream, d_alive, nbits,
                               meta->inv_row_ids.p, meta->inv_sources.p, meta->own_id, meta->inv_n, total,
                               reinterpret_cast<unsigned long long *>(real));
        cur = real;
        cur_bits = total;
    }
    if (deleted)
    {
        if (!cur) // no per-search filter: the delete bitmap alone
        {
            *eff_nbits = meta->delete_nbits;
            return meta->delete_alive.p;
        }
        uint64_t * both = aux.take<uint64_t>(words);
        const size_t bits = std::min(cur_bits, meta->delete_nbits);
        hipLaunchKernelGGL(and_bits_kernel, dim3((unsigned)ceil_div(words, (size_t)256)), dim3(256), 0, stream, cur,
                           ceil_div(cur_bits, (size_t)64), meta->delete_alive.p, ceil_div(meta->delete_nbits, (size_t)64), both,
                           words);
        cur = both;
        cur_bits = bits;
    }
    MSVS_HIP(hipGetLastError());
    *eff_nbits = cur_bits;
    return cur;
}

void apply_row_ids_map(const msvs_index::Meta * meta, int64_t * d_ids, size_t n, hipStream_t stream)
{
    if (!meta || !meta->row_ids_n || !n)
        return;
    hipLaunchKernelGGL(remap_ids_kernel, dim3((unsigned)ceil_div(n, (size_t)256)), dim3(256), 0, stream, d_ids, n,
                       meta->row_ids_map.p, meta->row_ids_n);
    MSVS_HIP(hipGetLastError());
}
}

/// Process-wide counters of the candidate passes (device side): [0] = result queries whose certificate failed, [1] = the
/// same for the coarse quantiser's passes (probe lists).
static unsigned long long * prefilter_fail_counter()
{
    static std::map<int, unsigned long long *> per_device;
    static std::mutex mu;
    int dev = 0;
    MSVS_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    auto it = per_device.find(dev);
    if (it != per_device.end())
        return it->second;
    unsigned long long * p = nullptr;
    MSVS_HIP(hipMalloc(&p, 128));
    MSVS_HIP(hipMemset(p, 0, 128));
    per_device[dev] = p;
    return p;
}
static std::atomic<unsigned long long> g_prefilter_queries{0}, g_coarse_queries{0};

namespace msvs
{

/// How an IVF search of nq queries is decomposed.
struct IvfSearchPlan
{
    uint32_t T;       // 1 = one query per block (ivf_scan_kernel); 2/4/8 = list-batched query tiles;
                      // 128 / 256 (BG_TQ * nqg) = matrix-core candidate pass + canonical re-rank (mfma_scan_kernels.hpp)
    uint32_t rpb;     // rows per work item
    uint32_t seg_max; // segments of the longest list
    uint32_t grid;    // batched: fixed grid size
    // matrix-core pass only
    uint32_t kc;       // candidates kept per query
    uint32_t rpb1;     // rows per block / segments of the canonical fallback scan (one query per block)
    uint32_t seg_max1;
    uint32_t fb_slots; // block slots (grid z) of the fallback
    uint32_t nqg;      // 128-query groups per workgroup of the candidate pass (1 or 2)
    bool h16;          // the candidate pass runs over the fp16 shadow (h16_scan_kernels.hpp)
    uint32_t h_mth;    // ... the number of rows per query its cut aims to leave below it (h16_sample_thr_kernel)
    uint32_t h_cap;    // ... and its candidate-buffer capacity per query
    uint32_t h_ncb;    // ... and its query tile: 32 * h_ncb queries resident in LDS
    bool mfma() const { return nqg != 0; }
};

static IvfSearchPlan plan_ivf(const msvs_index & ix, size_t nq, size_t nprobe, uint32_t k, bool allow_pass = true)
{
    IvfSearchPlan p{};
    const size_t pairs = nq * nprobe;
    const size_t nlist = std::max<size_t>(ix.nlist, 1);
    const size_t avg = std::max<size_t>(1, ix.n / nlist);
    // Matrix-core candidate pass: pays once ~4 queries share a list pass (the canonical scan is VALU-bound there);
    // needs finite, sane row norms for its error bound and k small enough for a 64-entry candidate list.
    {
        const int mode = allow_pass ? (int)options().ivf_pass : 0; // experiment knob: 0 = never, 2 = whenever eligible
        // row positions travel in the low word of the candidate keys: < 2^32 rows; NaN norms compare false
        // the shadow pass keeps a whole query tile in LDS: at least one column block of 32 queries must fit
        const bool h16 = ix.shadow_ready && options().ivf_h16 != 0 && h16_lds_bytes(1, ix.h_nch) <= 160 * 1024;
        // k <= 40: 64 candidates; the shadow pass also serves 40 < k <= 128 with 256 (hybrid searches take a vector top-100)
        const bool eligible = (k <= 40 || (h16 && k <= 128 && options().h16_k128 != 0)) && ix.xnorm.p && ix.xnorm_max < 1e30f
            && ix.n <= 0xfffffff0ull;
        const double min_pairs = h16 ? options().h16_min_pairs : 4.0;
        if (mode != 0 && eligible && ((double)pairs >= min_pairs * (double)nlist || mode >= 2))
        {
            p.h16 = h16;
            if (h16)
            {
                // queries per tile: what the average list is probed by, up to what LDS holds (and 4 accumulators)
                // (twice the average: probes concentrate on the popular lists; measured: 64-query tiles beat 32-query
                // ones from 8 pairs per list on)
                uint32_t ncb = (uint32_t)std::min<size_t>(4, std::max<size_t>(2, ceil_div(2 * ceil_div(pairs, nlist), (size_t)32)));
                if (options().h16_ncb >= 1)
                    ncb = (uint32_t)std::min(4.0, options().h16_ncb);
                while (ncb > 1 && h16_lds_bytes(ncb, ix.h_nch) > 160 * 1024)
                    ncb--;
                p.h_ncb = ncb;
            }
            {
                // sample = block 0 (<= 32 rows) of every probed list; the cut of a query = its m-th best sample row with
                // m = target * (sample rows) / (probed rows), i.e. about `target` rows of everything it probes lie below the
                // cut (h16_sample_thr_kernel).  target = 25 k: m ~ 9 on lists of ~1000 rows; the certificate fails when
                // fewer than k rows do (>= m of the k + m best rows fell into the sample: ~1e-6 there, and the floor m = 4
                // is only reached when the sample is a small fraction of the probed rows)
                // ... k > 40 (256 candidates re-ranked): 10 k, at least what k = 40 gets -- the sample's rank noise is relative
                // Round 4: 15 k where the wavefront cut kernel runs (nprobe <= 64).  Every appended row costs the main launch (a staged
                // record + its share of an atomic: 250 -> 150 rows per query took 27 us off the 0.34 ms launch of the bench step),
                // and a failed certificate has become cheap there -- the second chance, then a canonical scan of the few lists the
                // pruning left (4 of 32 768 queries on sigma-0.3 blobs).  Every list probed by every query (nprobe 256) keeps
                // 25 k: 66 instead of 11 fallbacks over 256 lists each cost 40 % of the step.
                p.h_mth = options().h16_nocut != 0 ? 0u
                                                    : (uint32_t)(k <= 40 ? std::max<size_t>(64, (nprobe <= 64 ? 15 : 25) * (size_t)k)
                                                                         : std::max<size_t>(1000, 10 * (size_t)k)); // the target
                if (options().h16_target >= 1 && options().h16_nocut == 0) // experiment knob
                    p.h_mth = (uint32_t)options().h16_target;
                // capacity: 8 x the target, and 2.5 x what the floor m = 4 leaves below the cut when every probed list is as
                // long as the longest one (sample fraction 32 / max_list_len)
                size_t cap = std::min<size_t>(std::max<size_t>(std::max<size_t>(1024, round_up(8 * (size_t)p.h_mth, 256)),
                                                               round_up(10 * ceil_div(ix.max_list_len, (size_t)H_ROWS), 256)),
                                              16384);
                if (options().cand_cap >= 1)
                    cap = std::max<size_t>(64, (size_t)options().cand_cap);
                // never more than the rows a query can meet
                p.h_cap = (uint32_t)std::min<size_t>(cap, std::max<size_t>(64, nprobe * round_up(ix.max_list_len, H_ROWS)));
            }
            // 256-query tiles (nqg = 2: one workgroup of 8 wavefronts per CU, the rows read once per 256 probing
            // queries) measured SLOWER than two independent 128-query workgroups per CU at every batch size
            // (4096 q/step: 1.92 vs 1.70 ms, 16384: 6.01 vs 5.89 ms): kept as a knob only
            p.nqg = options().ivf_nqg == 2 ? 2 : 1;
            p.T = BG_TQ * p.nqg;
            p.kc = k <= 12 ? 32 : k <= 40 ? 64 : 256;
            if (p.h16 && options().h16_kc >= k) // experiment knob: candidates per query of the shadow pass
                p.kc = (uint32_t)std::min<double>(k <= 40 ? 64 : 256, options().h16_kc);
            // work item = 1 slice of a list for a tile of <= 128 queries, a grid of 4096 blocks: with the selection cheap,
            // the finest granularity balances best (2 slices / 2048 blocks: +5-8 % step time at 1024 .. 16384 q/step)
            p.rpb = BG_ROWS;
            if (options().ivf_rpb >= BG_ROWS)
                p.rpb = (uint32_t)round_up((size_t)options().ivf_rpb, (size_t)BG_ROWS);
            p.grid = 4096 / p.nqg;
            if (options().ivf_grid >= 1)
                p.grid = (uint32_t)options().ivf_grid;
            // the candidate lists are per 128-row SLICE (<= 16 keys each), whatever the work-item size
            p.seg_max = (uint32_t)std::max<size_t>(1, ceil_div(ix.max_list_len, (size_t)BG_ROWS));
            // the canonical fallback (a handful of queries per 10 000) scans a probed list with FOUR blocks: its partial
            // lists are nq * nprobe * 4 * k keys whatever the longest list is (segments of 256 rows made that 84 x
            // larger on a skewed index, for a buffer that is reserved on every search; one block per list made a single
            // failing query cost 0.8 ms = 10 % of the bench step on average)
            // ... and SIXTEEN when their partial lists stay small (nprobe * 16 * k keys <= the merge's LDS fast path):
            // one failing query of the bench step then costs ~25 us instead of ~100 (it is the tail of the whole step)
            size_t segs = nprobe * 16 * (size_t)k <= 6144 ? 16 : 4;
            if (options().fb_segs >= 1)
                segs = (size_t)options().fb_segs;
            p.rpb1 = (uint32_t)round_up(std::max<size_t>(ceil_div(ix.max_list_len, segs), 16), 16);
            p.seg_max1 = (uint32_t)std::max<size_t>(1, ceil_div(std::max<size_t>(ix.max_list_len, 1), (size_t)p.rpb1));
            p.fb_slots = (uint32_t)std::min<size_t>(nq, 8);
            return p;
        }
    }
    // Query tiles only pay when several queries of the batch probe the same list.
    // (measured on MI355X, 1M x 768, nlist 1024, nprobe 32: T=4 beats T=8 up to ~8 pairs per list because its 128
    // VGPRs allow 4 waves/SIMD against 2; see profiles/r01_ivf_tuning_sweep.txt)
    p.T = pairs >= 16 * nlist ? 8 : (pairs >= 2 * nlist ? 4 : (pairs >= nlist ? 2 : 1));
    while (p.T > 1 && scan_lds_bytes(p.T, ix.ld / 4, k) > SCAN_LDS_BUDGET) // see plan_flat
        p.T /= 2;
    const size_t tiles = std::max<size_t>(1, pairs / p.T);
    if (p.T == 1)
    {
        // one (query, list, segment) per block: many small blocks, ~8192 of them
        size_t rpb = round_up(std::max<size_t>(16, avg * pairs / 8192), 16);
        // keep a query's partial lists (~1.5 * nprobe * avg / rpb of them) within the LDS budget of the merge block
        size_t rpb_min = round_up(std::max<size_t>(64, nprobe * avg * 3 / 2 / 400), 16);
        p.rpb = (uint32_t)std::min<size_t>(std::max<size_t>(rpb, rpb_min), 256);
    }
    else
    {
        // work items of up to 512 rows walked by a fixed grid of 4096 blocks (4 rounds at 4 blocks/CU)
        size_t rpb = round_up(std::max<size_t>(64, avg * tiles / 4096), 16);
        p.rpb = (uint32_t)std::min<size_t>(rpb, p.T == 8 ? 1024 : 512);
    }
    p.grid = 4096;
    // tuning knobs for experiments (never needed in production): MSVS_IVF_T / MSVS_IVF_RPB / MSVS_IVF_GRID
    {
        const int t = (int)options().ivf_t, r = (int)options().ivf_rpb, g = (int)options().ivf_grid;
        if (t == 1 || t == 2 || t == 4 || t == 8)
            p.T = (uint32_t)t;
        if (r >= 16)
            p.rpb = (uint32_t)round_up((size_t)r, 16);
        if (g >= 1)
            p.grid = (uint32_t)g;
    }
    p.seg_max = (uint32_t)std::max<size_t>(1, ceil_div(ix.max_list_len, p.rpb));
    p.grid = (uint32_t)std::min<size_t>(p.grid, std::max<size_t>(1, tiles * ceil_div(avg, p.rpb) * 2));
    return p;
}

static void h16_flat_dispatch(int metric, uint32_t ncb, int shape, uint32_t grid, size_t lds, const H16Params & a,
                              const H16FlatParams & f, hipStream_t stream);
static size_t table_pass_scratch(size_t n, size_t nq, uint32_t k);
static size_t fallback_cap(size_t nq, size_t nprobe, size_t seg_max, uint32_t k);

static size_t big_cand_cap(size_t nprobe, size_t slices_max)
{
    size_t limit = 16384;
    if (options().cand_cap >= 1) // experiment / test knob
        limit = std::max<size_t>(64, (size_t)options().cand_cap);
    return std::min<size_t>(nprobe * slices_max * BG_SLICE_K, limit);
}

/// allow_pass: false for a search over a compacted view (it runs the canonical plan, whose partial lists can be LARGER than the
/// candidate pass's buffers: an inner-product index with one giant list has many segments per probe -- round 4's sparse-filter
/// case overflowed an arena reserved for the candidate-pass plan)
static size_t index_search_scratch(const msvs_index & ix, size_t nq, uint32_t k, size_t nprobe, bool allow_pass = true)
{
    size_t b = nq * (size_t)ix.ld * 4 + 2 * (nq * (size_t)(ix.ld + 32) * 4 + 4096) + 4096; // queries (+ split copies)
    if (ix.type == MSVS_INDEX_FLAT)
        return b + flat_scratch_bytes(ix.n, nq, k, ix.ld) + table_pass_scratch(ix.n, nq, std::min<uint32_t>(k, 40))
            + (ix.shadow_ready ? nq * ((size_t)ix.h_nch * 128 + 24 + (size_t)H_FLAT_SAMPLE_BLK * H_ROWS * 4) + 9216 : 0); // shadow pass: query images, sample words
    IvfSearchPlan p = plan_ivf(ix, nq, nprobe, k, allow_pass);
    size_t need = b + flat_scratch_bytes(ix.nlist, nq, (uint32_t)nprobe, ix.ld)
        + table_pass_scratch(ix.nlist, nq, (uint32_t)std::min<size_t>(nprobe, 40))
        + nq * nprobe * 4
        + (5 * ix.nlist + 16 + nq * nprobe) * 4 + 32768
        + (4 * ix.nlist + 18) * 4 + 1024 + 1024 // the list scan's counters taken up front, the padded query images
        + 4 * nq * nprobe * 4 + nq * 4 + 2 * (ix.nlist + 1) * 4 + 8192 // probe pruning: surviving probes (two stages), the second plan
        + (ix.c_shadow_ready ? nq * ((size_t)ix.h_nch * 128 + 24 + round_up(ix.nlist, (size_t)H_ROWS) * 4 + ceil_div(ix.nlist, (size_t)H_ROWS) * 4) + 9216 : 0);
    if (p.mfma())
        need += nq * (p.h16 ? (size_t)p.h_cap : big_cand_cap(nprobe, p.seg_max)) * 8
            + nq * (size_t)p.kc * 8 + nq * 32 + 8192
            + fallback_cap(nq, nprobe, p.seg_max1, k) * nprobe * (size_t)p.seg_max1 * k * 8
            + (p.h16 ? nq * ((size_t)ix.h_nch * 128 + 8 + 4 + 4) + 2048 + nq * nprobe * H_ROWS * 4 + 4 * ix.nlist + 4096 : 0);
    else
        need += nq * nprobe * (size_t)p.seg_max * k * 8;
    return need;
}

/// Error model of the split-bf16 candidate pass (mfma_scan_kernels.hpp header), times the test knob MSVS_IVF_EPS_SCALE.
static void set_error_model(RerankParams & rp, size_t dim)
{
    const double scale = 1.05 * options().ivf_eps_scale, dd = (double)dim; // knob: inflate eps to force the fallback
    rp.c_dot = scale * (3.1 * ldexp(1.0, -16) + 3.05 * dd * ldexp(1.0, -23));
    rp.c_norm = scale * (dd + 8.0) * ldexp(1.0, -24);
    rp.c_canon = scale * 32.0 * ldexp(1.0, -24);
}

/// A plain row table searched through the matrix-core candidate pass: the table is one "list" every query probes
/// (single_list_plan_kernel); 16 candidates per (query, 128-row slice) -> 32 / 64 per query -> canonical re-rank ->
/// exact top-k with the same certificate / canonical fallback as the list scan (mfma_scan_kernels.hpp).
/// Used for the coarse quantiser (table = centroids, result = probe lists) and for FLAT indexes (table = all rows).
/// The queries of a search as the fp16 shadow passes want them (h16_prep_queries_kernel): the coarse pass over the centroid
/// shadow and the list scan use the same images, scale and norms -- prepared once per search.
struct H16Queries
{
    uint4 * qh = nullptr;
    float2 * qinfo = nullptr;
    float * qnorm = nullptr;
    float * qrho = nullptr; // measured rounding error of every query's image (set_error_model_h16)
    uint32_t * counters = nullptr; // in: the list scan's counters, to be zeroed along the way (cleared flag: qh != nullptr)
    uint32_t n_counters = 0;
    const uint32_t * coarse_words = nullptr; // out: the coarse pass's approximate distance word of every (query, centroid) ...
    uint32_t coarse_npad = 0;                // ... [nq][coarse_npad]: what the probe pruning of the list scan reads
    const uint32_t * probe_words = nullptr;  // or, a sharded search: the words of the given probes, [nq][nprobe] (ProbeWords::given)
    const float * probe_dis = nullptr;       // or, a small batch: canonical distances of the probes from the canonical coarse scan
    float * upre = nullptr;                  // the pre-pruning's bound per query (h16_preprune_kernel), for the list scan's second stage
    bool prepruned = false;                  // the probe lists went through the pre-pruning (h16_preprune_kernel)
};

struct TablePass
{
    const float * rows;    // n x ld
    const uint32_t * ids;  // nullable: id of row r (else r)
    const float * norms;   // |row|^2
    float norm_max;
    size_t n;
    const uint64_t * alive; // nullable filter bitmap over ids
    size_t nbits;
    uint32_t k;
    int32_t * out_probes; // either the probe lists ...
    int64_t * out_ids;    // ... or (ids, distances)
    float * out_dis;
    int cosine;
    const char * prof_name;
    bool flat_h16 = false; // the table is a FLAT index with an fp16 shadow: candidates through h16_flat_kernel
    float h16_rho = -1.f;  // measured rounding error of the table's shadow (set_error_model_h16; < 0: worst case)
    bool h16 = false; // the table is the centroid table and its fp16 shadow is usable: scan through h16_sample_kernel
    H16Queries * h16_out = nullptr; // h16: where the pass leaves the queries' fp16 images for the list scan that follows
};

static uint32_t table_fallback_rpb(size_t n) { return (uint32_t)round_up(std::max<size_t>(256, ceil_div(n, (size_t)256)), 16); }

static size_t table_pass_scratch(size_t n, size_t nq, uint32_t k)
{
    const size_t nslices = ceil_div(std::max<size_t>(n, 1), (size_t)BG_ROWS);
    const size_t segs = ceil_div(std::max<size_t>(n, 1), (size_t)table_fallback_rpb(n));
    return nq * (big_cand_cap(1, nslices) * 8 + 64 * 8 + 96 + 4 /* the coarse tail's slow queue */) + fallback_cap(nq, 1, segs, k) * segs * k * 8 + 65536;
}

/// Worth it once the (query tile) x (128-row slice) grid can occupy the chip (64 work items measured no better than
/// the canonical scan on the 1024-centroid table, 256 items 2x better; the tiles shrink to 32 queries to get there)
/// and enough queries share each pass over the rows for the canonical scan to be VALU-bound (>= 16).
static bool table_pass_eligible(size_t n, const float * norms, float norm_max, size_t nq, uint32_t k, double knob)
{
    const int mode = (int)knob; // experiment knob: 0 = never, 2 = whenever possible
    const size_t items = ceil_div(nq, (size_t)32) * ceil_div(n, (size_t)BG_ROWS); // at the smallest tile (32 queries)
    return mode != 0 && norms && norm_max < 1e30f /* false for NaN */ && k <= 40 && n >= 256 && n <= 0xfffffff0ull
        && ((items >= 256 && nq >= 16) || mode == 2);
}

/// The coarse quantiser through the centroid shadow needs no 128-query tiles: it pays from ~200 queries on (its ten
/// launches cost ~85 us whatever the batch; the canonical centroid scan + merge of 64 / 256 / 512 queries 54 / 95 / 120 us).
static bool coarse_shadow_eligible(const msvs_index & ix, size_t nq, size_t nprobe)
{
    return options().coarse_mfma != 0 && options().coarse_h16 != 0 && ix.c_shadow_ready && ix.cnorm.p && ix.cnorm_max < 1e30f
        && nprobe <= 40 && ix.nlist >= 256 && options().coarse_h16_min_q >= 1 && nq >= (size_t)options().coarse_h16_min_q
        && nq * round_up(ix.nlist, (size_t)H_ROWS) * 4 <= ((size_t)128 << 20);
}

/// The canonical fallback of a candidate pass: the queries on the device-side fail list are scanned and merged in ROUNDS of
/// at most `cap` (their partial lists are indexed by the rank in the round), so the buffers are sized for `cap` queries
/// instead of all nq -- normally nobody is on the list and every launch exits at once.
static size_t fallback_cap(size_t nq, size_t nprobe, size_t seg_max, uint32_t k)
{
    const size_t per_query = std::max<size_t>(1, nprobe * seg_max * k * 8);
    if (options().fb_cap >= 1) // test knob: several rounds on small batches
        return std::min(nq, (size_t)options().fb_cap);
    return std::min(nq, std::max<size_t>(64, ((size_t)256 << 20) / per_query));
}

static void run_fallback_rounds(int metric, ScanParams c, IvfMergeParams fm, size_t nq, size_t cap, uint32_t slots, hipStream_t stream)
{
    for (size_t base = 0; base < nq; base += cap)
    {
        c.slot_base = fm.slot_base = (uint32_t)base;
        c.slot_cap = fm.slot_cap = (uint32_t)cap;
        launch_ivf_scan_subset(metric, c, slots, stream);
        launch_ivf_merge_subset(metric, fm, slots, stream);
    }
}

static void set_error_model_h16(RerankParams & rp, size_t dim, float rho_table = -1.f, const float * qrho = nullptr);

HostSignal & host_signal()
{
    static thread_local HostSignal s;
    return s;
}

static __global__ void host_signal_kernel(const uint32_t * nfail, uint32_t * h_nfail, uint32_t * h_flag, uint32_t seq)
{
    *h_nfail = *nfail;
    __builtin_amdgcn_s_waitcnt(0);
    __hip_atomic_store(h_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); // the re-rank's result stores ended with its launch
}

/// The armed signal of a host-pointer call (index_internal.hpp: HostSignal): the certificate-failure count goes to the host behind the
/// launches enqueued so far, the host waits for the word.  -> true: nobody failed -- the results are complete and in place.
static bool host_signal_round(HostSignal & hs, const uint32_t * nfail, hipStream_t stream)
{
    hs.armed = false;
    hs.used = true;
    hipLaunchKernelGGL(host_signal_kernel, dim3(1), dim3(1), 0, stream, nfail, hs.nfail, hs.flag, hs.seq);
    MSVS_HIP(hipGetLastError());
    for (uint64_t spins = 1; __atomic_load_n(hs.flag, __ATOMIC_ACQUIRE) != hs.seq; spins++)
    {
        __builtin_ia32_pause();
        if ((spins & 0xfff) == 0) // a failed launch never sets the word: look at the stream now and then
        {
            const hipError_t e = hipStreamQuery(stream);
            if (e == hipSuccess)
                break;
            if (e != hipErrorNotReady)
                fail(MSVS_ERR_DEVICE, "host-pointer search: %s", hipGetErrorString(e));
        }
    }
    if (__atomic_load_n(hs.flag, __ATOMIC_ACQUIRE) != hs.seq)
    {
        // the stream ran dry without the word: the signal kernel did not run (its launch failed after hipGetLastError looked) -- the
        // failure count in pinned memory is not this call's: the results cannot be trusted
        MSVS_HIP(hipStreamSynchronize(stream));
        if (__atomic_load_n(hs.flag, __ATOMIC_ACQUIRE) != hs.seq)
            fail(MSVS_ERR_DEVICE, "host-pointer search: the completion word was never written (signal launch lost)");
    }
    return __atomic_load_n(hs.nfail, __ATOMIC_ACQUIRE) == 0;
}

/// Second half of a table pass, whatever produced the candidates: canonical re-rank + certificate, then the canonical scan
/// of the table for the queries on the fail list.
static RerankParams table_rerank_params(const msvs_index & ix, const TablePass & t, const ScanParams & a, const float * qnorm,
                                        const uint64_t * cand, const uint64_t * bound, uint32_t kc, uint32_t * failq, uint32_t * nfail,
                                        bool h16, const float * qrho);
static void table_pass_fallback(int m, const TablePass & t, const ScanParams & a, size_t nq, uint32_t * failq, uint32_t * nfail,
                                uint64_t * partial1, size_t fb_cap, uint32_t rpb1, uint32_t seg_max1, const int32_t * probes0,
                                const int64_t * list_off, hipStream_t stream);

/// slowq / nslow (nullable): the coarse quantiser's selection + band ran in coarse_tail_kernel; the re-rank launch serves only the
/// queries it queued.
static void table_pass_tail(const msvs_index & ix, int m, const TablePass & t, const ScanParams & a, size_t nq, const float * qnorm,
                            const uint64_t * cand, const uint64_t * bound, uint32_t kc, uint32_t * failq, uint32_t * nfail,
                            uint64_t * partial1, size_t fb_cap, uint32_t rpb1, uint32_t seg_max1, const int32_t * probes0,
                            const int64_t * list_off, bool h16, const float * qrho, hipStream_t stream, const uint32_t * slowq = nullptr,
                            const uint32_t * nslow = nullptr)
{
    RerankParams rp = table_rerank_params(ix, t, a, qnorm, cand, bound, kc, failq, nfail, h16, qrho);
    rp.qmap = slowq;
    rp.qcount = nslow;
    if (t.out_probes)
        g_coarse_queries.fetch_add(nq, std::memory_order_relaxed);
    launch_ivf_rerank(scan_metric(m), rp, (uint32_t)nq, stream);
    table_pass_fallback(m, t, a, nq, failq, nfail, partial1, fb_cap, rpb1, seg_max1, probes0, list_off, stream);
}

static RerankParams table_rerank_params(const msvs_index & ix, const TablePass & t, const ScanParams & a, const float * qnorm,
                                        const uint64_t * cand, const uint64_t * bound, uint32_t kc, uint32_t * failq, uint32_t * nfail,
                                        bool h16, const float * qrho)
{
    RerankParams rp{};
    rp.Y = a.Y;
    rp.ids = t.ids;
    rp.Q = a.Q;
    rp.qnorm = qnorm;
    rp.cand = cand;
    rp.bound = bound;
    rp.kc = kc;
    rp.k = t.k;
    rp.ld4 = a.ld4;
    rp.out_probes = t.out_probes;
    rp.out_ids = t.out_ids;
    rp.out_dis = t.out_dis;
    rp.cosine = t.cosine;
    if (h16)
        set_error_model_h16(rp, ix.dim, t.h16_rho, qrho);
    else
        set_error_model(rp, ix.dim);
    rp.xmax = t.norm_max;
    rp.failq = failq;
    rp.nfail = nfail;
    rp.early_exit = options().rerank_early != 0 && !t.out_probes ? 1 : 0; // result passes only: the centroid table gains nothing
    rp.band = t.out_probes && h16 && options().coarse_band != 0 ? 1 : 0;
    rp.stat_fail = prefilter_fail_counter() + (t.out_probes ? 1 : 0); // msvs_prefilter_stats / msvs_coarse_stats
    rp.stat_skip = options().rerank_stats != 0 ? prefilter_fail_counter() + (t.out_probes ? 4 : 2) : nullptr;
    return rp;
}

static void table_pass_fallback(int m, const TablePass & t, const ScanParams & a, size_t nq, uint32_t * failq, uint32_t * nfail,
                                uint64_t * partial1, size_t fb_cap, uint32_t rpb1, uint32_t seg_max1, const int32_t * probes0,
                                const int64_t * list_off, hipStream_t stream)
{
    // queries without a certificate: canonical scan of the table
    ScanParams c = a;
    c.k = t.k;
    c.partial = partial1;
    c.rows_per_block = rpb1;
    c.seg_max = seg_max1;
    c.qmap = failq;
    c.qcount = nfail;
    const uint32_t slots = (uint32_t)std::min<size_t>(nq, 8);
    IvfMergeParams fm{};
    fm.partial = partial1;
    fm.probes = probes0;
    fm.list_off = list_off;
    fm.nprobe = 1;
    fm.seg_max = seg_max1;
    fm.rows_per_block = rpb1;
    fm.k = t.k;
    fm.out_probes = t.out_probes;
    fm.out_ids = t.out_ids;
    fm.out_dis = t.out_dis;
    fm.cosine = t.cosine;
    fm.qmap = failq;
    fm.qcount = nfail;
    HostSignal & hs = host_signal();
    if (hs.armed && t.flat_h16 && !t.out_probes)
    {
        // host-pointer call of a few queries: tell the host, let it decide about the fallback
        if (host_signal_round(hs, nfail, stream))
            return;
        run_fallback_rounds(scan_metric(m), c, fm, nq, fb_cap, slots, stream);
        MSVS_HIP(hipStreamSynchronize(stream));
        return;
    }
    run_fallback_rounds(scan_metric(m), c, fm, nq, fb_cap, slots, stream);
}

/// Tests (msvs_debug_coarse_words): where this host thread's last centroid-shadow pass left its words (valid until the thread's
/// next search on that stream).
struct CoarseLast
{
    const uint32_t * words = nullptr;
    uint32_t nq = 0, npad = 0;
    hipStream_t stream = nullptr;
};
static thread_local CoarseLast g_coarse_last;

static void table_candidate_pass(const msvs_index & ix, Scratch & scr, int m, const float * dq, size_t nq,
                                 const TablePass & t, hipStream_t stream)
{
    const uint32_t ld = ix.ld, nrows = (uint32_t)t.n;
    uint32_t kc = t.k <= 12 ? 32 : 64;
    if (t.h16 && options().coarse_kc >= t.k) // experiment knob: candidates per query of the centroid-shadow pass
        kc = (uint32_t)std::min<double>(64, options().coarse_kc);
    const uint32_t nqg = options().ivf_nqg == 2 ? 2 : 1; // 256-query tiles: see plan_ivf
    // queries per tile: the full 128 when that still gives the chip >= 512 work items, else 64 or 32 (the rows are then
    // re-read by more tiles -- cheap for a table that lives in L2, like the centroids)
    uint32_t tq = BG_TQ * nqg;
    while (tq > 32 && ceil_div(nq, (size_t)tq) * ceil_div(t.n, (size_t)BG_ROWS) < 512)
        tq /= 2;
    const uint32_t nslices = (uint32_t)ceil_div(t.n, (size_t)BG_ROWS);
    const uint32_t cap = (uint32_t)big_cand_cap(1, nslices);
    uint32_t * pairs = scr.take<uint32_t>(nq);
    int32_t * probes0 = scr.take<int32_t>(nq);
    int64_t * list_off = scr.take<int64_t>(6);  // whole table, sample, rest
    uint32_t * small = scr.take<uint32_t>(9);   // pair_off[2], nfail, work_off[2] x 3
    uint32_t * qstate = scr.take<uint32_t>(2 * nq);
    float * qnorm = scr.take<float>(nq);
    uint64_t * candbuf = scr.take<uint64_t>(nq * (size_t)cap);
    uint64_t * cand = scr.take<uint64_t>(nq * (size_t)kc);
    uint64_t * bound = scr.take<uint64_t>(nq);
    uint32_t * failq = scr.take<uint32_t>(nq);
    const uint32_t rpb1 = table_fallback_rpb(t.n), seg_max1 = (uint32_t)ceil_div(t.n, (size_t)rpb1);
    const size_t fb_cap = fallback_cap(nq, 1, seg_max1, t.k);
    uint64_t * partial1 = scr.take<uint64_t>(fb_cap * (size_t)seg_max1 * t.k);
    uint32_t * nfail = small + 2;
    const uint32_t rpb = BG_ROWS; // work item = 1 slice (see plan_ivf)
    if (!t.h16 && !t.flat_h16) // (the shadow passes' query preparation kernel does the rest)
    {
        MSVS_HIP(hipMemsetAsync(small, 0, 9 * sizeof(uint32_t), stream));
        MSVS_HIP(hipMemsetAsync(qstate, 0xFF, nq * sizeof(uint32_t), stream));
        MSVS_HIP(hipMemsetAsync(qstate + nq, 0, nq * sizeof(uint32_t), stream));
        // plan 0: the whole table (also what the fallback scans)
        launch_single_list_plan((uint32_t)nq, 0, nrows, rpb, tq, pairs, probes0, list_off, small, small + 3, stream);
    }
    float4 * qsplit = nullptr;
    if (!t.h16 && !t.flat_h16)
    {
        launch_row_sqnorm(dq, qnorm, nq, ld / 4, nullptr, stream);
        qsplit = scr.take<float4>(nq * (size_t)ceil_div((size_t)ld / 4, (size_t)8) * 8);
        launch_split_queries(dq, (uint32_t)nq, ld / 4, qsplit, stream);
    }
    // A table too long for "16 keys per slice" to fit the candidate buffers is searched in two phases: a sample first,
    // whose m-th best candidate becomes the query's cut for the rest (sample_cut_kernel)
    const bool two_phase = (size_t)nslices * BG_SLICE_K > cap;
    const uint32_t sample = two_phase
        ? (uint32_t)std::min<size_t>(t.n / 2, round_up(std::max<size_t>(32768, t.n / 32), (size_t)rpb))
        : 0;
    if (two_phase && !t.h16 && !t.flat_h16) // (the shadow passes sample their own way)
    {
        launch_single_list_plan((uint32_t)nq, 0, sample, rpb, tq, pairs, probes0, list_off + 2, small, small + 5,
                                stream);
        launch_single_list_plan((uint32_t)nq, sample, nrows, rpb, tq, pairs, probes0, list_off + 4, small, small + 7,
                                stream);
    }
    ScanParams a{};
    a.Y = reinterpret_cast<const float4 *>(t.rows);
    a.ids = t.ids;
    a.alive = t.alive;
    a.nbits = (uint32_t)std::min<size_t>(t.nbits, 0xffffffffu);
    a.Q = reinterpret_cast<const float4 *>(dq);
    a.partial = candbuf;
    a.ld4 = ld / 4;
    a.k = kc;
    a.nq = (uint32_t)nq;
    a.rows_per_block = rpb;
    a.probes = probes0;
    a.list_off = list_off;
    a.nprobe = 1;
    a.seg_max = nslices;
    a.pairs = pairs;
    a.pair_off = small;
    a.work_off = small + 3;
    a.nlist = 1;
    a.xcd_order = 1;
    a.qnorm = qnorm;
    a.xnorm = t.norms;
    a.qthr = qstate;
    a.qcnt = qstate + nq;
    a.cand_cap = cap;
    a.tile_q = tq;
    a.Qsplit = qsplit;
    const size_t tiles = ceil_div(nq, (size_t)tq);
    ProfileScope prof(t.prof_name, stream);
    if (t.h16)
    {
        // every approximate distance of the batch through the centroid shadow, then the kc best per query
        const uint32_t G = (uint32_t)ceil_div(t.n, (size_t)H_ROWS), n_pad = G * H_ROWS;
        uint4 * qh = scr.take<uint4>(nq * (size_t)ix.h_nch * 8 + 64); // + what the register-tile list scan may read past the end
        float2 * qinfo = scr.take<float2>(nq);
        float * qrho = scr.take<float>(nq); // measured rounding error of every image (set_error_model_h16)
        float * qn16 = qnorm; // computed by the preparation kernel itself
        uint32_t * sample = scr.take<uint32_t>(nq * (size_t)n_pad);
        uint32_t * cpairs = scr.take<uint32_t>(nq * (size_t)G);
        uint32_t * cpoff = scr.take<uint32_t>(G + 1);
        uint32_t * cwoff = scr.take<uint32_t>(G + 1);
        uint32_t * failq2c = t.out_probes ? scr.take<uint32_t>(nq) : nullptr; // queries coarse_tail_kernel leaves to the block-per-query re-rank ...
        uint32_t * nslow = t.out_probes ? scr.take<uint32_t>(1) : nullptr;    // ... and their count (cleared by the preparation kernel)
        // one launch: norms, fp16 images, the one-list plan of this pass (what its fallback scans), its fail counter, and the
        // zeroed counters of the list scan that follows
        H16PrepAux aux{};
        aux.pairs = pairs;
        aux.probes0 = probes0;
        aux.list_off = list_off;
        aux.pair_off = small;
        aux.work_off = small + 3;
        aux.nfail = nfail;
        aux.row_end = nrows;
        aux.rows_per_block = rpb;
        aux.tq = tq;
        if (t.h16_out && t.h16_out->counters)
        {
            aux.zero[0] = t.h16_out->counters;
            aux.nzero[0] = t.h16_out->n_counters;
        }
        if (nslow)
        {
            aux.zero[1] = nslow;
            aux.nzero[1] = 1;
        }
        hipLaunchKernelGGL(h16_prep_queries_kernel, dim3((unsigned)ceil_div(nq, (size_t)4)), dim3(256), 0, stream, dq, (uint32_t)nq, ld,
                           ix.h_nch, ix.h_inv_scale, m == MSVS_METRIC_L2 ? 0 : 1, qh, qinfo, qn16, 1, aux, qrho);
        if (t.h16_out)
        {
            t.h16_out->qrho = qrho;
            t.h16_out->qh = qh;
            t.h16_out->qinfo = qinfo;
            t.h16_out->qnorm = qn16;
            if (options().coarse_h16 != 2) // (the dedicated kernel writes every word)
            {
                t.h16_out->coarse_words = sample;
                t.h16_out->coarse_npad = n_pad;
                g_coarse_last = CoarseLast{sample, (uint32_t)nq, n_pad, stream};
            }
        }
        if (options().coarse_h16 != 2)
        {
            // dedicated kernel: 2 x 2 blocks per wavefront, every word of `sample` written
            const uint32_t items = (uint32_t)(ceil_div((size_t)G, (size_t)2) * ceil_div(ceil_div(nq, (size_t)32), (size_t)2));
            const uint32_t cgrid = (uint32_t)std::min<size_t>(ceil_div((size_t)items, (size_t)4), (size_t)device_cu_count() * 2);
            if (options().coarse_h16 != 3 && nq >= 256)
            {
                // round 6: 128 x 128 workgroup tiles, both operands through LDS (coarse_gemm_kernel)
                // (smaller tiles than 128 x 128: several workgroups per CU in different phases -- the launch is its prologue, epilogue
                // and stores, not its loop)
                const int nt = options().coarse_gemm_tq == 128 ? 2 : 1, ns = options().coarse_gemm_tc == 128 ? 2 : 1;
                const dim3 ggrid((unsigned)ceil_div(nq, (size_t)(64 * nt)), (unsigned)ceil_div((size_t)G, (size_t)(2 * ns)));
                auto go = [&](auto kern) {
                    hipLaunchKernelGGL(kern, ggrid, dim3(256), 0, stream, ix.c_shadow.p, ix.h_nch, qh, qinfo, t.norms, (uint32_t)t.n, (uint32_t)nq,
                                       sample);
                };
                if (scan_metric(m) == M_IP)
                    nt == 2 ? (ns == 2 ? go(coarse_gemm_kernel<M_IP, 2, 2>) : go(coarse_gemm_kernel<M_IP, 2, 1>))
                            : (ns == 2 ? go(coarse_gemm_kernel<M_IP, 1, 2>) : go(coarse_gemm_kernel<M_IP, 1, 1>));
                else
                    nt == 2 ? (ns == 2 ? go(coarse_gemm_kernel<M_L2, 2, 2>) : go(coarse_gemm_kernel<M_L2, 2, 1>))
                            : (ns == 2 ? go(coarse_gemm_kernel<M_L2, 1, 2>) : go(coarse_gemm_kernel<M_L2, 1, 1>));
            }
            else if (scan_metric(m) == M_IP)
                hipLaunchKernelGGL((coarse_h16_kernel<M_IP>), dim3(cgrid), dim3(BLOCK), 0, stream, ix.c_shadow.p, ix.h_nch, qh, qinfo,
                                   t.norms, (uint32_t)t.n, (uint32_t)nq, sample);
            else
                hipLaunchKernelGGL((coarse_h16_kernel<M_L2>), dim3(cgrid), dim3(BLOCK), 0, stream, ix.c_shadow.p, ix.h_nch, qh, qinfo,
                                   t.norms, (uint32_t)t.n, (uint32_t)nq, sample);
        }
        else
        {
            // coarse_h16 = 2: h16_sample_kernel on the trivial plan (one 32 x 32 tile per wavefront)
            MSVS_HIP(hipMemsetAsync(sample, 0xFF, nq * (size_t)n_pad * sizeof(uint32_t), stream));
            hipLaunchKernelGGL(coarse_plan_kernel, dim3((unsigned)ceil_div(std::max<size_t>(nq * (size_t)G, G + 1), (size_t)256)), dim3(256),
                               0, stream, (uint32_t)nq, G, cpairs, cpoff, cwoff);
            H16Params h{};
            h.H = ix.c_shadow.p;
            h.hoff = ix.c_hoff.p;
            h.nks = ix.h_nks;
            h.nch = ix.h_nch;
            h.Qh = qh;
            h.qinfo = qinfo;
            h.xnorm = t.norms;
            h.list_off = ix.c_list_off.p;
            h.pairs = cpairs;
            h.pair_off = cpoff;
            h.work_off = cwoff;
            h.nlist = G;
            h.nprobe = G;
            h.sample_out = sample;
            const uint32_t sgrid = device_cu_count() * 8;
            if (scan_metric(m) == M_IP)
                hipLaunchKernelGGL((h16_sample_kernel<M_IP, 1>), dim3(sgrid), dim3(BLOCK), 0, stream, h);
            else
                hipLaunchKernelGGL((h16_sample_kernel<M_L2, 1>), dim3(sgrid), dim3(BLOCK), 0, stream, h);
        }
        // round 6: selection + band re-rank of the probe lists in ONE launch, a wavefront per query (coarse_tail_kernel); the block-per-query
        // re-rank then serves only the queries whose band could not be formed
        const bool tail = t.out_probes && options().coarse_tail != 0 && options().coarse_band != 0 && options().wave_select == 1
            && options().rerank_stats == 0 && n_pad <= 32 * WAVE && kc <= WAVE && kc > t.k && nslow != nullptr;
        // ... and those, too, by their own wavefront while the index has not had one for a while (coarse_exact_wave): the queue's chain is
        // three launches that find nothing to do on every other search (~15 us of the 4096-query step).  Which form runs is decided on
        // a stamp the kernel leaves in pinned memory (the sequence number of the last search of this index that met such a query; read
        // without synchronisation: a hint -- both forms are exact, the inline one is slow when MANY queries need it)
        bool slow_inline = false;
        uint32_t * slow_stamp = nullptr;
        uint32_t seq = 0;
        if (tail && ix.plan_fb.pairs)
        {
            slow_stamp = ix.plan_fb.pairs + 1;
            seq = ix.plan_fb.seq.fetch_add(1, std::memory_order_relaxed) + 1;
            const uint32_t last = *reinterpret_cast<volatile uint32_t *>(slow_stamp), seen = *reinterpret_cast<volatile uint32_t *>(slow_stamp + 1);
            const uint32_t base = (uint32_t)options().coarse_slow_window;
            uint32_t window = std::max(ix.plan_fb.window.load(std::memory_order_relaxed), base);
            if (seen && base != 0 && last != 0 && last == ix.plan_fb.last_inline.load(std::memory_order_relaxed)
                && ix.plan_fb.bumped.exchange(last, std::memory_order_relaxed) != last)
            {
                // the last inline attempt met such a query again: this index has them for good -- try less often (data whose centroid
                // distances sit closer together than the error bound keeps a few hundred of them per batch)
                window = (uint32_t)std::min<uint64_t>((uint64_t)window * 4, 1u << 20);
                ix.plan_fb.window.store(window, std::memory_order_relaxed);
            }
            slow_inline = options().coarse_slow_inline != 0 && (seen == 0 || (uint32_t)(seq - last) > window);
            if (slow_inline && seen) // (a retry: the first search of an index is not one)
                ix.plan_fb.last_inline.store(seq, std::memory_order_relaxed);
        }
        if (tail)
        {
            const RerankParams rp = table_rerank_params(ix, t, a, qn16, cand, bound, kc, failq, nfail, true, qrho);
            const dim3 tgrid((unsigned)ceil_div(nq, (size_t)(BLOCK / WAVE)));
            if (scan_metric(m) == M_IP)
                hipLaunchKernelGGL((coarse_tail_kernel<M_IP>), tgrid, dim3(BLOCK), 0, stream, sample, (uint32_t)nq, n_pad, rp, cand, bound, failq2c, nslow,
                                   (uint32_t)t.n, slow_stamp, seq, slow_inline ? 1 : 0);
            else
                hipLaunchKernelGGL((coarse_tail_kernel<M_L2>), tgrid, dim3(BLOCK), 0, stream, sample, (uint32_t)nq, n_pad, rp, cand, bound, failq2c, nslow,
                                   (uint32_t)t.n, slow_stamp, seq, slow_inline ? 1 : 0);
        }
        else
            hipLaunchKernelGGL(coarse_select_kernel, dim3((unsigned)ceil_div(nq, (size_t)4)), dim3(BLOCK), 0, stream, sample, (uint32_t)nq,
                               n_pad, kc, cand, bound, (int)options().wave_select);
        MSVS_HIP(hipGetLastError());
        if (slow_inline) // (every query left the tail launch with its probes)
        {
            g_coarse_queries.fetch_add(nq, std::memory_order_relaxed);
            return;
        }
        table_pass_tail(ix, m, t, a, nq, qn16, cand, bound, kc, failq, nfail, partial1, fb_cap, rpb1, seg_max1, probes0, list_off, true, qrho,
                        stream, tail ? failq2c : nullptr, tail ? nslow : nullptr);
        return;
    }
    if (t.flat_h16)
    {
        // the FLAT table's fp16 shadow (h16_scan_kernels.hpp, "FLAT tables"): sample -> cut -> exhaustive shadow scan
        const uint32_t nblk = (uint32_t)ceil_div(t.n, (size_t)H_ROWS);
        const uint32_t gs = std::min<uint32_t>(nblk, H_FLAT_SAMPLE_BLK), blk_stride = std::max<uint32_t>(1, nblk / gs);
        const uint32_t n_pad = gs * H_ROWS;
        uint4 * qh = scr.take<uint4>(nq * (size_t)ix.h_nch * 8 + 64);
        float2 * qinfo = scr.take<float2>(nq);
        float * qrho = scr.take<float>(nq);
        uint32_t * sample = scr.take<uint32_t>(nq * (size_t)n_pad);
        uint32_t * sched = scr.take<uint32_t>(9); // 8 item cursors of the scan + the sample's ticket
        H16PrepAux aux{};
        aux.pairs = pairs;
        aux.probes0 = probes0;
        aux.list_off = list_off;
        aux.pair_off = small;
        aux.work_off = small + 3;
        aux.nfail = nfail;
        aux.row_end = nrows;
        aux.rows_per_block = rpb;
        aux.tq = tq;
        aux.zero[0] = sched;
        aux.nzero[0] = 9;
        hipLaunchKernelGGL(h16_prep_queries_kernel, dim3((unsigned)ceil_div(nq, (size_t)4)), dim3(256), 0, stream, dq, (uint32_t)nq, ld,
                           ix.h_nch, ix.h_inv_scale, m == MSVS_METRIC_L2 ? 0 : 1, qh, qinfo, qnorm, 1, aux, qrho);
        // ~target rows of the whole table below the cut: the m-th smallest of the S sample rows, m = target S / n (the sample is a
        // 1 / blk_stride share of the table), at least 4 (the certificate fails when fewer than k rows lie below the cut)
        const size_t target = std::max<size_t>(64, 25 * (size_t)t.k);
        const uint32_t mth = (uint32_t)std::min<size_t>(64, std::max<size_t>(4, ceil_div(target * n_pad, std::max<size_t>(t.n, 1))));
        const bool few = nq <= 32 && options().flat_sample_few != 0; // one query tile: sample and cut in one launch
        if (few)
        {
            if (scan_metric(m) == M_IP)
                hipLaunchKernelGGL((flat_sample_few_kernel<M_IP>), dim3(gs), dim3(BLOCK), 0, stream, ix.shadow.p, ix.h_nch, qh, qinfo, t.norms, gs,
                                   (uint32_t)nq, sample, blk_stride, nrows, t.ids, t.alive, a.nbits, mth, qstate, qstate + nq, sched + 8);
            else
                hipLaunchKernelGGL((flat_sample_few_kernel<M_L2>), dim3(gs), dim3(BLOCK), 0, stream, ix.shadow.p, ix.h_nch, qh, qinfo, t.norms, gs,
                                   (uint32_t)nq, sample, blk_stride, nrows, t.ids, t.alive, a.nbits, mth, qstate, qstate + nq, sched + 8);
        }
        else
        {
            const uint32_t items = (uint32_t)(ceil_div((size_t)gs, (size_t)2) * ceil_div(ceil_div(nq, (size_t)32), (size_t)2));
            const uint32_t cgrid = (uint32_t)std::min<size_t>(ceil_div((size_t)items, (size_t)4), (size_t)device_cu_count() * 2);
            if (scan_metric(m) == M_IP)
                hipLaunchKernelGGL((coarse_h16_kernel<M_IP>), dim3(cgrid), dim3(BLOCK), 0, stream, ix.shadow.p, ix.h_nch, qh, qinfo, t.norms,
                                   n_pad, (uint32_t)nq, sample, blk_stride, nrows, t.ids, t.alive, a.nbits);
            else
                hipLaunchKernelGGL((coarse_h16_kernel<M_L2>), dim3(cgrid), dim3(BLOCK), 0, stream, ix.shadow.p, ix.h_nch, qh, qinfo, t.norms,
                                   n_pad, (uint32_t)nq, sample, blk_stride, nrows, t.ids, t.alive, a.nbits);
        }
        if (!few)
            hipLaunchKernelGGL(flat_cut_kernel, dim3((unsigned)ceil_div(nq, (size_t)4)), dim3(BLOCK), 0, stream, sample, (uint32_t)nq, n_pad, mth,
                               qstate, qstate + nq);
        H16Params h{};
        h.H = ix.shadow.p;
        h.nks = ix.h_nks;
        h.nch = ix.h_nch;
        h.Qh = qh;
        h.qinfo = qinfo;
        h.xnorm = t.norms;
        h.ids = t.ids;
        h.alive = t.alive;
        h.nbits = a.nbits;
        h.qthr = qstate;
        h.qcnt = qstate + nq;
        h.partial = candbuf;
        h.cand_cap = cap;
        h.sched = sched;
        h.group_appends = (uint32_t)options().h16_group_appends;
        uint32_t ncb = (uint32_t)std::min<size_t>(3, ceil_div(nq, (size_t)32));
        if (options().flat_ncb >= 1)
            ncb = (uint32_t)std::min(3.0, options().flat_ncb);
        while (ncb > 1 && h16_lds_bytes(ncb, ix.h_nch) > 160 * 1024)
            ncb--;
        H16FlatParams f{};
        f.nq = (uint32_t)nq;
        f.nblk = nblk;
        f.n_rows = nrows;
        f.ntiles = (uint32_t)ceil_div(nq, (size_t)32 * ncb);
        // segments of 64 blocks (3 MB of shadow at d = 768) for a single tile (it stays in LDS from item to item: 489 items for 256
        // CUs at 1M rows), 128 / 256 blocks with several / many tiles: the tiles of a segment are consecutive items, the CUs of an
        // XCD walk it together and share it through their L2 whatever its size, and the tile load (147 KB per item) weighs less
        // (measured at 4096 queries: 64 / 128 / 256 / 512 / 1024 blocks -> 573 / 587 / 602 / 580 / 552 k QPS)
        f.segb = f.ntiles >= 8 ? 4 * H_FLAT_SEGB : f.ntiles >= 2 ? 2 * H_FLAT_SEGB : H_FLAT_SEGB;
        // ... but a short table is cut finer: at least ~4 items per CU, down to one block (pair) per wavefront
        const uint32_t want_items = 4 * device_cu_count();
        if ((size_t)ceil_div((size_t)nblk, (size_t)f.segb) * f.ntiles < want_items)
            f.segb = (uint32_t)std::max<size_t>((size_t)H_NW * (f.ntiles >= 2 ? 2 : 1), (size_t)nblk * f.ntiles / want_items);
        if (options().flat_segb >= 1)
            f.segb = (uint32_t)options().flat_segb;
        f.nseg = (uint32_t)ceil_div((size_t)nblk, (size_t)f.segb);
        f.rot = f.ntiles >= 2 ? (uint32_t)options().flat_rot : 0;
        h.lazy_flush = options().flat_lazy_flush != 0 && f.ntiles >= 2 ? 1 : 0;
        const size_t lds = h16_lds_bytes(ncb, ix.h_nch);
        const uint32_t per_cu = (uint32_t)std::min<size_t>(2, std::max<size_t>(1, (160 * 1024) / lds));
        const uint32_t fgrid = options().h16_grid >= 1 ? (uint32_t)options().h16_grid : device_cu_count() * per_cu;
        // two row blocks per wavefront whenever a tile has two column blocks or more (every query fragment read from LDS feeds two MFMAs:
        // a single 64-query tile 0.488 -> 0.437 ms per step over 1M x 768; a 32-query tile gains nothing)
        const int shape = options().flat_h16 == 3 ? 1 : options().flat_h16 == 6 ? 6 : (f.ntiles >= 2 || ncb >= 2 ? 2 : 1); // (3 / 6: experiments)
        h16_flat_dispatch(scan_metric(m), ncb, shape, fgrid, lds, h, f, stream);
        MSVS_HIP(hipGetLastError());
        launch_cand_select(candbuf, h.qcnt, h.qthr, cap, (uint32_t)nq, kc, cand, bound, stream);
        table_pass_tail(ix, m, t, a, nq, qnorm, cand, bound, kc, failq, nfail, partial1, fb_cap, rpb1, seg_max1, probes0, list_off, true, qrho,
                        stream);
        return;
    }
    if (two_phase)
    {
        ScanParams sa = a;
        sa.list_off = list_off + 2;
        sa.work_off = small + 5;
        launch_ivf_mfma_scan(scan_metric(m), nqg,
                             (uint32_t)std::min<size_t>(tiles * ceil_div((size_t)sample, (size_t)rpb), 4096 / nqg), sa, stream,
                             "table_scan", false);
        launch_cand_select(candbuf, a.qcnt, a.qthr, cap, (uint32_t)nq, kc, cand, bound, stream);
        // the m-th best sample candidate leaves ~m * n / sample rows of the table below the cut; the query fails its
        // certificate when fewer than k + 1 of them do, i.e. when >= m of the table's k + m best rows fell into the
        // sample: m = 6 at a 1/32 sample makes that ~1e-5 for k = 10 (m = 3 measured 1.6 % fallbacks)
        const uint32_t mth = (uint32_t)std::min<size_t>(kc, std::max<size_t>(6, 2 + ceil_div((size_t)8 * t.k * sample, t.n)));
        launch_sample_cut(cand, kc, mth, (uint32_t)nq, a.qthr, stream);
        sa.list_off = list_off + 4;
        sa.work_off = small + 7;
        launch_ivf_mfma_scan(scan_metric(m), nqg,
                             (uint32_t)std::min<size_t>(tiles * ceil_div(t.n - sample, (size_t)rpb), 4096 / nqg), sa, stream,
                             "table_scan", false);
    }
    else
        launch_ivf_mfma_scan(scan_metric(m), nqg,
                             (uint32_t)std::min<size_t>(tiles * ceil_div(t.n, (size_t)rpb), 4096 / nqg), a, stream,
                             "table_scan", false);
    launch_cand_select(candbuf, a.qcnt, a.qthr, cap, (uint32_t)nq, kc, cand, bound, stream);
    table_pass_tail(ix, m, t, a, nq, qnorm, cand, bound, kc, failq, nfail, partial1, fb_cap, rpb1, seg_max1, probes0, list_off, false, nullptr,
                    stream);
}

/// Error model of the fp16 shadow pass (h16_scan_kernels.hpp), u = 2^-11:
///   stored values: |fp16(x s) / s - x| <= u |x| + 2^-38 max|x| (rounding to nearest; the second term covers fp16
///   subnormals, the scale s puts max|x| in [2^13, 2^14)); same for the query with its own scale, so
///   |<x', q'> - <x, q>| <= (2u + u^2) |x||q| + 2.01 sqrt(d) 2^-38 |x|max |q| + d 2^-76 |x|max |q|;
///   products of two fp16 are exact in f32; the MFMA accumulates n = 16 ceil(d/16) of them in f32 in an order and with a
///   per-addition rounding we do not rely on: <= n 2^-23 sum|x'_i q'_i| <= 1.01 n 2^-23 |x||q|
///   (tests/test_gpu_parity.py::test_mfma_accumulation_error_bound_on_hardware measures it);
/// c_norm / c_canon as in set_error_model (the norms and the canonical distance do not change).
/// Round 4: the rounding term is MEASURED where it can be.  With dx = x' - x, dq = q' - q:
///   |<x', q'> - <x, q>| = |<dx, q> + <x, dq> + <dx, dq>| <= (rho_x + rho_q + rho_x rho_q) |x||q|,  rho_x = max over the table's rows
///   of |dx| / |x| (h16_rho_kernel at build: `rho_table`), rho_q = |dq| / |q| of the query's image (h16_prep_queries_kernel: `qrho`,
///   added per query by rerank_eps / the cut kernels).  Rounding errors are uniform in their interval and relative to the element's
///   binade, so rho ~ 0.43 u over a row of hundreds of elements where the worst case is u (+ the subnormal terms, which the
///   measurement contains): eps drops to ~0.45 of the worst case and stays a proof.  rho_table < 0 or no qrho: the worst case.
static void set_error_model_h16(RerankParams & rp, size_t dim, float rho_table, const float * qrho)
{
    const double scale = 1.05 * options().ivf_eps_scale, dd = (double)round_up(dim, H_CHUNK);
    const double rho_worst = ldexp(1.0, -11) + 2.01 * sqrt(dd) * ldexp(1.0, -38); // u |v| + the fp16 subnormal quantum per element
    const bool measured = options().h16_rho != 0;
    const double rho_x = measured && rho_table >= 0.f ? (double)rho_table : rho_worst;
    rp.qrho = measured ? qrho : nullptr;
    rp.qrho_scale = scale * (1.0 + rho_x);
    rp.c_dot = scale * (rho_x + (rp.qrho ? 0.0 : rho_worst * (1.0 + rho_x)) + 1.01 * dd * ldexp(1.0, -23));
    rp.c_norm = scale * ((double)dim + 8.0) * ldexp(1.0, -24);
    rp.c_canon = scale * 32.0 * ldexp(1.0, -24);
}

/// The same model for the kernels that bound sample / centroid words (H16Prune): the rows' table and the centroid table each
/// with its own measured rounding error.
/// remote_words: the coarse words were computed by ANOTHER rank's centroid shadow and query images (ProbeWords::given): its measured
/// rounding errors are not known here -- the worst-case model covers them.  foreign_rows: the bound speaks about rows of other ranks
/// (the routed search's pre-pruning over the whole index): the worst-case row error instead of this shard's measured one.
static void set_prune_error_model(H16Prune & pr, const msvs_index & ix, const float * qrho, bool remote_words = false, bool foreign_rows = false)
{
    RerankParams ex{}, ec{};
    set_error_model_h16(ex, ix.dim, foreign_rows ? -1.f : ix.h_rho, qrho); // (the query images of the row scan are this rank's own)
    set_error_model_h16(ec, ix.dim, remote_words ? -1.f : ix.c_rho, remote_words ? nullptr : qrho);
    pr.c_dot = ex.c_dot;
    pr.c_dot_c = ec.c_dot;
    pr.c_norm = ex.c_norm;
    pr.c_canon = ex.c_canon;
    pr.qrho = ex.qrho;
    pr.qrho_scale = ex.qrho_scale;
    pr.qrho_scale_c = ec.qrho_scale;
}

template <int METRIC, int NCB>
static void h16_launch(uint32_t grid, size_t lds, const H16Params & a, hipStream_t stream)
{
    // more than 64 KiB of dynamic LDS needs the attribute raised once per kernel
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&h16_scan_kernel<METRIC, NCB>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL((h16_scan_kernel<METRIC, NCB>), dim3(grid), dim3(64 * H_NW), lds, stream, a);
}

template <int METRIC>
static void h16_dispatch(uint32_t ncb, uint32_t grid, size_t lds, const H16Params & a, hipStream_t stream)
{
    switch (ncb)
    {
        case 1: h16_launch<METRIC, 1>(grid, lds, a, stream); break;
        case 2: h16_launch<METRIC, 2>(grid, lds, a, stream); break;
        case 3: h16_launch<METRIC, 3>(grid, lds, a, stream); break;
        default: h16_launch<METRIC, 4>(grid, lds, a, stream); break;
    }
}

template <int METRIC, int NCB, int NRB, int RING>
static void h16_flat_launch(uint32_t grid, size_t lds, const H16Params & a, const H16FlatParams & f, hipStream_t stream)
{
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&h16_flat_kernel<METRIC, NCB, NRB, RING>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL((h16_flat_kernel<METRIC, NCB, NRB, RING>), dim3(grid), dim3(64 * H_NW), lds, stream, a, f);
}

/// shape: 2 = two row blocks per wavefront, 3-slot ring (batches of several tiles); 1 = one row block, the list scan's 4-slot ring
/// (a single tile: the table streams out of HBM once); 6 = one row block, 6-slot ring (experiment)
template <int METRIC>
static void h16_flat_dispatch_m(uint32_t ncb, int shape, uint32_t grid, size_t lds, const H16Params & a, const H16FlatParams & f,
                                hipStream_t stream)
{
    if (shape == 2 && ncb >= 3)
        h16_flat_launch<METRIC, 3, 2, 3>(grid, lds, a, f, stream);
    else if (shape == 2 && ncb == 2)
        h16_flat_launch<METRIC, 2, 2, 3>(grid, lds, a, f, stream);
    else if (shape == 6 && ncb >= 3)
        h16_flat_launch<METRIC, 3, 1, 6>(grid, lds, a, f, stream);
    else if (shape == 6 && ncb == 2)
        h16_flat_launch<METRIC, 2, 1, 6>(grid, lds, a, f, stream);
    else if (shape == 6)
        h16_flat_launch<METRIC, 1, 1, 6>(grid, lds, a, f, stream);
    else if (ncb >= 3)
        h16_flat_launch<METRIC, 3, 1, H_RING>(grid, lds, a, f, stream);
    else if (ncb == 2)
        h16_flat_launch<METRIC, 2, 1, H_RING>(grid, lds, a, f, stream);
    else
        h16_flat_launch<METRIC, 1, 1, H_RING>(grid, lds, a, f, stream);
}

static void h16_flat_dispatch(int metric, uint32_t ncb, int shape, uint32_t grid, size_t lds, const H16Params & a,
                              const H16FlatParams & f, hipStream_t stream)
{
    ProfileScope prof("flat_shadow_scan", stream);
    if (metric == M_IP)
        h16_flat_dispatch_m<M_IP>(ncb, shape, grid, lds, a, f, stream);
    else
        h16_flat_dispatch_m<M_L2>(ncb, shape, grid, lds, a, f, stream);
}

/// Experiments (option h16_stamps): the per-item wall-clock stamps of the last main launch (H16Params::stamps).
static DevBuf<uint64_t> & g_h16_stamps = *new DevBuf<uint64_t>(); // leaked on purpose: no hipFree after the runtime is gone
static uint32_t g_h16_stamp_grid = 0;

uint32_t device_cu_count()
{
    static std::mutex mu;
    static std::map<int, uint32_t> cus;
    int dev = 0;
    MSVS_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    auto it = cus.find(dev);
    if (it != cus.end())
        return it->second;
    int n = 0;
    MSVS_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    return cus[dev] = (uint32_t)std::max(n, 1);
}

/// What the last shadow pass of this host thread left behind (tests: msvs_debug_h16_keys reads the kernel's OWN approximate
/// keys out of the scratch arena; valid until the thread's next search on that stream).
struct H16Last
{
    const uint64_t * partial = nullptr;
    const uint32_t * qcnt = nullptr;
    uint32_t cap = 0;
    size_t nq = 0;
    hipStream_t stream = nullptr;
};
static thread_local H16Last g_h16_last;

/// List scan of a batch over the fp16 shadow: sample launch -> cut -> main launch -> candidate select -> canonical
/// re-rank + certificate -> canonical fallback for the queries without one (h16_scan_kernels.hpp).
static void h16_list_scan(const msvs_index & ix, Scratch & scr, int m, const float * dq, size_t nq, uint32_t k,
                          size_t nprobe, const IvfSearchPlan & pl, const int32_t * d_probes, const uint64_t * d_alive,
                          size_t nbits, uint64_t * partial, int64_t * d_ids, float * d_dis, hipStream_t stream,
                          const H16Queries & prepared)
{
    const uint32_t ld = ix.ld;
    // work item = (list, tile of 32 * h_ncb probing queries); main launch: rows [list_off + 32, end), sample launch: block 0
    const int32_t * plan_probes = d_probes; // (pre-pruned by the caller where that applies: index_search_device_one)
    IvfPlanParams pp{};
    pp.probes = plan_probes;
    pp.list_off = ix.list_mid32.p;
    pp.list_end = ix.list_off.p + 1;
    pp.whole_off = ix.list_off.p;
    pp.n_pairs = (uint32_t)(nq * nprobe);
    pp.nlist = (uint32_t)ix.nlist;
    pp.rows_per_block = 0x7fffffffu; // one segment per non-empty row range
    pp.T = 32 * pl.h_ncb;
    const size_t n_counters = 4 * ix.nlist + 2 + 16; // two plans (cnt, fill each), nfail, 16 queue cursors, nfail2
    const bool zeroed = prepared.qh && prepared.counters && prepared.n_counters >= n_counters;
    uint32_t * counters = zeroed ? prepared.counters : scr.take<uint32_t>(n_counters);
    pp.cnt = counters;
    pp.fill = counters + ix.nlist;
    uint32_t * nfail = counters + 2 * ix.nlist;
    uint32_t * sched = nfail + 1; // 8 work-queue cursors per launch
    uint32_t * nfail2 = sched + 16; // queries still without a certificate after the second chance
    pp.pair_off = scr.take<uint32_t>(ix.nlist + 1);
    pp.work_off = scr.take<uint32_t>(ix.nlist + 1);
    pp.pairs = scr.take<uint32_t>(nq * nprobe);
    // row segments of the main launch, sized on the device for ~4 items per workgroup of its grid (option h16_segs = 0: one item per (list, tile))
    // (only when the launch could run short of items: at least min(nq, nlist) lists keep a pair on any but degenerate batches, and a
    // launch of >= 4 items per CU gains nothing from the extra pass of the plan kernel: + 3 us per plan, 1 % of the 4096-query step)
    const bool want_segs = options().h16_segs == 2 || (options().h16_segs != 0 && std::min(nq, ix.nlist) < (size_t)4 * device_cu_count());
    uint32_t * seg_words = want_segs ? scr.take<uint32_t>(2) : nullptr;
    pp.seg_out = seg_words;
    pp.seg_target_items = 4 * device_cu_count();
    // the queries' fp16 images: the coarse pass over the centroid shadow left them behind, or they are made here -- in FRONT of the plan
    // when that launch can clear the counters on its way (a search with given probes: the searching side of a routed step; a memset
    // between two kernels is a barrier packet, ~8 us of idle device)
    float * qnorm = prepared.qh ? prepared.qnorm : scr.take<float>(nq);
    uint4 * qh = prepared.qh ? prepared.qh : scr.take<uint4>(nq * (size_t)ix.h_nch * 8 + 64); // + what the register kernel may read past the last image
    float2 * qinfo = prepared.qh ? prepared.qinfo : scr.take<float2>(nq);
    float * qrho = prepared.qh ? prepared.qrho : scr.take<float>(nq);
    bool prep_done = prepared.qh != nullptr;
    auto prep = [&](const H16PrepAux & aux) {
        ProfileScope prof("ivf_prep", stream);
        hipLaunchKernelGGL(h16_prep_queries_kernel, dim3((unsigned)ceil_div(nq, (size_t)4)), dim3(256), 0, stream, dq,
                           (uint32_t)nq, ld, ix.h_nch, ix.h_inv_scale, m == MSVS_METRIC_L2 ? 0 : 1, qh, qinfo, qnorm, 1, aux, qrho);
        prep_done = true;
    };
    if (!zeroed)
    {
        if (ivf_plan_fused(pp)) // (a small batch: the one-launch plan needs no zeroed counts and clears nfail / the cursors / nfail2 itself)
        {
            pp.zero = nfail;
            pp.nzero = 18;
        }
        else if (!prep_done)
        {
            H16PrepAux aux{};
            aux.zero[0] = counters;
            aux.nzero[0] = (uint32_t)n_counters;
            prep(aux);
        }
        else
            MSVS_HIP(hipMemsetAsync(counters, 0, n_counters * sizeof(uint32_t), stream));
    }
    // ... and the sample launch's partition of the same pairs: block 0 of every probed list, tiles of 32 queries (small
    // workgroups) -- one scan launch computes both
    pp.list_off2 = ix.list_off.p;
    pp.list_end2 = ix.list_mid32.p;
    pp.T2 = 32; // one column block of 32 queries per sample item
    pp.work_off2 = scr.take<uint32_t>(ix.nlist + 1);
    // (measurement: rows the two launches read -- the second pruning's plan, when there is one, is the main launch's)
    // ... or when the lists are LONG (a pair of >= 2.5 MB: a second plan's 18 us are a fraction of one list's scan -- batches of 64
    // over 10M+ rows; on the 1.5 MB lists of the 1M-row config the pre-pruning has done what can be done by then)
    const double pair_bytes = (double)ix.n / (double)std::max<size_t>(ix.nlist, 1) * (2.0 * (double)ix.dim + 8.0);
    // Round 6: the pairs that MATTER are those the pre-pruning left, and the host does not know their number -- but the last search of
    // this index over the same batch shape does: its plan kernel left the count in a pinned word (msvs_index::PlanFeedback; no
    // synchronisation, stale by one search, a hint only).  On SURVEY 8d's sigma-0.3 blobs 31 of 32 probes are gone before the sample
    // launch: ~1.05 pairs per query, nothing left for a second stage to find -- its plan (three launches, ~17 us of the 4096-query
    // step) and the bound arithmetic of the cut kernel are skipped until a batch comes back with more pairs.
    double eff_pairs = (double)nq * (double)nprobe;
    uint32_t * fb_out = nullptr;
    if (prepared.prepruned && ix.plan_fb.pairs && options().h16_prune != 2 && options().h16_feedback != 0)
    {
        const uint32_t seen = *reinterpret_cast<volatile uint32_t *>(ix.plan_fb.pairs);
        // (the same nprobe; another batch size scales the count: the searching side of a routed step sees a few queries more or less
        // every time)
        if (seen != 0xFFFFFFFFu && ix.plan_fb.nq != 0 && ix.plan_fb.nprobe == (uint32_t)nprobe)
            eff_pairs = std::min(eff_pairs, 1.25 * (double)seen * ((double)nq / (double)ix.plan_fb.nq) + 64.0);
        ix.plan_fb.nq = (uint32_t)nq;
        ix.plan_fb.nprobe = (uint32_t)nprobe;
        fb_out = ix.plan_fb.pairs;
    }
    const bool prune_pays = options().h16_prune == 2 || eff_pairs > (double)ix.nlist * (double)pp.T || pair_bytes >= 2.5e6;
    const bool prune2 = options().h16_prune != 0 && prune_pays && (prepared.coarse_words || prepared.probe_words || prepared.probe_dis)
        && ix.list_radius.p && k <= 128 && nprobe <= 64 && options().wave_select != 0
        && (!prepared.probe_dis || ((ix.metric == MSVS_METRIC_L2 || ix.metric == MSVS_METRIC_COSINE) && !d_alive));
    if (options().rerank_stats != 0)
    {
        pp.stat_rows = prefilter_fail_counter() + 10;
        pp.stat_first = prune2 ? 0 : 1;
    }
    pp.pairs_out = fb_out;
    launch_ivf_plan(pp, stream);
    pp.stat_rows = nullptr;
    pp.pairs_out = nullptr;
    IvfPlanParams pa = pp;
    pa.work_off = pp.work_off2;
    uint32_t * sample = scr.take<uint32_t>(nq * nprobe * H_ROWS);
    uint32_t * qstate = scr.take<uint32_t>(2 * nq);
    uint64_t * cand = scr.take<uint64_t>(nq * (size_t)pl.kc);
    uint64_t * bound = scr.take<uint64_t>(nq);
    uint32_t * failq = scr.take<uint32_t>(nq);
    if (!prep_done)
        prep(H16PrepAux{});
    // (no fill of `sample`: the sample launch writes all 32 words of every pair whose list has rows, and the cut kernels
    // take a pair whose list is empty as 32 missing rows)
    H16Params a{};
    a.H = ix.shadow.p;
    a.hoff = ix.hoff.p;
    a.nks = ix.h_nks;
    a.nch = ix.h_nch;
    a.Qh = qh;
    a.qinfo = qinfo;
    a.xnorm = ix.xnorm.p;
    a.list_off = ix.list_off.p;
    a.ids = ix.row_ids.p;
    a.alive = d_alive;
    a.nbits = (uint32_t)std::min<size_t>(nbits, 0xffffffffu);
    a.pairs = pp.pairs;
    a.pair_off = pp.pair_off;
    a.nlist = (uint32_t)ix.nlist;
    a.nprobe = (uint32_t)nprobe;
    a.xcd_order = (uint32_t)options().ivf_xcd;
    a.group_appends = (uint32_t)options().h16_group_appends;
    a.qthr = qstate;
    a.qcnt = qstate + nq;
    a.partial = partial;
    a.cand_cap = pl.h_cap;
    a.sample_out = sample;
    // persistent workgroups pulling work items from per-XCD queues: one per CU (the tile takes most of the LDS)
    const size_t lds = h16_lds_bytes(pl.h_ncb, ix.h_nch);
    const uint32_t per_cu = (uint32_t)std::min<size_t>(2, std::max<size_t>(1, (160 * 1024) / lds));
    const uint32_t grid = options().h16_grid >= 1 ? (uint32_t)options().h16_grid : device_cu_count() * per_cu;
    H16Prune pr{};
    {
        ProfileScope prof("ivf_sample_scan", stream);
        a.work_off = pa.work_off;
        const uint32_t sgrid = device_cu_count() * 8; // one wavefront per (list, 32-query column block), grid-stride
        if (scan_metric(m) == M_IP)
            hipLaunchKernelGGL((h16_sample_kernel<M_IP, 1>), dim3(sgrid), dim3(BLOCK), 0, stream, a);
        else
            hipLaunchKernelGGL((h16_sample_kernel<M_L2, 1>), dim3(sgrid), dim3(BLOCK), 0, stream, a);
        if (nprobe <= 64 && options().wave_select != 0)
        {
            // probe pruning (h16_scan_kernels.hpp: H16Prune): L2 indexes whose coarse pass left every centroid's approximate
            // distance behind; the surviving probes get their own plan for the main launch
            // ... when the lists are probed by more queries than one tile holds (then fewer pairs mean fewer passes over a list;
            // below that the second plan and the looser cut cost more than the dropped pairs save: sigma-0.3 blobs at nprobe 2)
            if (prune2)
            {
                set_prune_error_model(pr, ix, qrho, !prepared.coarse_words && prepared.probe_words != nullptr);
                pr.coarse_words = prepared.coarse_words;
                pr.npad = prepared.coarse_npad;
                pr.probe_words = prepared.coarse_words ? nullptr : prepared.probe_words;
                pr.probe_dis = prepared.coarse_words || prepared.probe_words ? nullptr : prepared.probe_dis;
                pr.radius = ix.list_radius.p;
                pr.cnorm = ix.cnorm.p;
                pr.ip = ix.metric == MSVS_METRIC_COSINE ? 1 : ix.metric == MSVS_METRIC_IP ? 2 : 0;
                pr.qnorm = qnorm;
                pr.xmax = ix.xnorm_max;
                pr.cmax = ix.cnorm_max;
                pr.k = k;
                pr.out_probes = scr.take<int32_t>(nq * nprobe);
                pr.upre = prepared.upre; // (null without the pre-pruning)
                pr.stat = options().rerank_stats != 0 ? prefilter_fail_counter() + 8 : nullptr;
                pr.stat_all = prepared.prepruned ? 0 : 1;
            }
            hipLaunchKernelGGL(h16_sample_thr_wave_kernel, dim3((unsigned)ceil_div(nq, (size_t)4)), dim3(BLOCK), 0, stream,
                               sample, plan_probes, ix.list_off.p, (uint32_t)nq, (uint32_t)nprobe, pl.h_mth, qstate,
                               qstate + nq, partial, pl.h_cap, options().wave_select == 3 ? 0 : 1, pr);
        }
        else
            hipLaunchKernelGGL(h16_sample_thr_kernel, dim3((unsigned)ceil_div(nq, (size_t)4)), dim3(BLOCK), 0, stream,
                               sample, plan_probes, ix.list_off.p, (uint32_t)nq, (uint32_t)nprobe, pl.h_mth, qstate,
                               qstate + nq, partial, pl.h_cap);
    }
    if (pr.on())
    {
        // the main launch's plan over the surviving pairs
        IvfPlanParams p2 = pp;
        p2.zero = nullptr;
        p2.nzero = 0;
        p2.probes = pr.out_probes;
        p2.cnt = counters + 2 * ix.nlist + 18;
        p2.fill = p2.cnt + ix.nlist;
        p2.pair_off = scr.take<uint32_t>(ix.nlist + 1);
        p2.work_off = scr.take<uint32_t>(ix.nlist + 1);
        p2.pairs = scr.take<uint32_t>(nq * nprobe);
        p2.seg_out = seg_words ? seg_words + 1 : nullptr;
        p2.work_off2 = nullptr;
        if (options().rerank_stats != 0)
        {
            p2.stat_rows = prefilter_fail_counter() + 10;
            p2.stat_first = 1;
        }
        launch_ivf_plan(p2, stream);
        pp.pairs = p2.pairs;
        pp.pair_off = p2.pair_off;
        pp.work_off = p2.work_off;
        pp.seg_out = p2.seg_out;
        a.pairs = pp.pairs;
        a.pair_off = pp.pair_off;
    }
    {
        ProfileScope prof("ivf_scan", stream);
        a.work_off = pp.work_off;
        a.seg_blocks = pp.seg_out;
        a.sched = sched + 8;
        {
            if (options().h16_stamps != 0)
            {
                const size_t words = (size_t)grid * H_STAMP_ITEMS * 4;
                if (g_h16_stamps.n < words)
                {
                    // (experiment knob h16_stamps: one process-wide buffer -- another stream's scan may still write stamps into the old
                    // one: the device is drained before it is replaced; the knob is not meant for concurrent searches)
                    static std::mutex stamps_mu;
                    std::lock_guard<std::mutex> lk(stamps_mu);
                    MSVS_HIP(hipDeviceSynchronize());
                    if (g_h16_stamps.n < words)
                        g_h16_stamps.alloc(words);
                }
                MSVS_HIP(hipMemsetAsync(g_h16_stamps.p, 0, words * 8, stream));
                a.stamps = g_h16_stamps.p;
                g_h16_stamp_grid = grid;
            }
            if (scan_metric(m) == M_IP)
                h16_dispatch<M_IP>(pl.h_ncb, grid, lds, a, stream);
            else
                h16_dispatch<M_L2>(pl.h_ncb, grid, lds, a, stream);
        }
    }
    MSVS_HIP(hipGetLastError());
    g_h16_last = H16Last{partial, qstate + nq, pl.h_cap, nq, stream};
    launch_cand_select(partial, qstate + nq, qstate, pl.h_cap, (uint32_t)nq, pl.kc, cand, bound, stream);
    RerankParams rp{};
    rp.Y = reinterpret_cast<const float4 *>(ix.vecs.p);
    rp.ids = ix.row_ids.p;
    rp.Q = reinterpret_cast<const float4 *>(dq);
    rp.qnorm = qnorm;
    rp.cand = cand;
    rp.bound = bound;
    rp.kc = pl.kc;
    rp.k = k;
    rp.ld4 = ld / 4;
    rp.out_ids = d_ids;
    rp.out_dis = d_dis;
    rp.cosine = ix.metric == MSVS_METRIC_COSINE;
    set_error_model_h16(rp, ix.dim, ix.h_rho, qrho);
    rp.xmax = ix.xnorm_max;
    rp.failq = failq;
    rp.nfail = nfail;
    rp.early_exit = options().rerank_early != 0 ? 1 : 0;
    const bool second = options().rerank_second != 0 && k <= RA_KMAX;
    uint64_t * ek_hint = second && options().rerank_hint != 0 ? scr.take<uint64_t>(nq) : nullptr; // written for failing queries only
    rp.ek_out = ek_hint;
    rp.stat_fail = second ? nullptr : prefilter_fail_counter(); // the statistic counts queries that reach the canonical scan
    rp.stat_skip = options().rerank_stats != 0 ? prefilter_fail_counter() + 2 : nullptr;
    // the second chance inside the re-rank launch (256-thread blocks: every shape but the rerank_groups = 32 experiment)
    const bool fused = second && options().rerank_fused != 0 && (pl.kc > 64 || options().rerank_groups != 32);
    uint32_t * failq2 = failq;
    uint32_t * nfail_final = nfail;
    RerankAllParams ra{};
    if (second)
    {
        failq2 = scr.take<uint32_t>(nq);
        nfail_final = nfail2;
        ra.partial = partial;
        ra.qcnt = qstate + nq;
        ra.qthr = qstate;
        ra.cap = pl.h_cap;
        ra.failq_in = failq;
        ra.nfail_in = nfail;
        ra.failq_out = failq2;
        ra.nfail_out = nfail2;
        ra.stat_fail = prefilter_fail_counter();
        ra.ek_in = ek_hint;
        if (fused)
        {
            rp.fuse_second = 1;
            rp.ra = ra;
        }
    }
    launch_ivf_rerank(scan_metric(m), rp, (uint32_t)nq, stream);
    rp.fuse_second = 0;
    g_prefilter_queries.fetch_add(nq, std::memory_order_relaxed);
    // a small batch through msvs_index_search (search_entry.hip: the pinned form): the host learns how many queries are without a
    // certificate and enqueues the second chance and the fallback rounds -- three launches that normally find nothing to do, ~14 of
    // the batch's ~138 us -- only when somebody is; the results are in pinned memory at its return either way
    HostSignal & hs = host_signal();
    const bool signalled = hs.armed;
    if (signalled && host_signal_round(hs, fused ? nfail2 : nfail, stream))
        return;
    // second chance of the queries without a certificate: every row of their candidate buffers, certified against the cut
    if (second && !fused)
        launch_ivf_rerank_all(scan_metric(m), rp, ra, (uint32_t)nq, stream);
    // queries without a certificate: canonical scan, one query per block (normally zero of them)
    const size_t fb_cap = fallback_cap(nq, nprobe, pl.seg_max1, k);
    uint64_t * partial1 = scr.take<uint64_t>(fb_cap * nprobe * (size_t)pl.seg_max1 * k);
    ScanParams c{};
    c.Y = rp.Y;
    c.ids = ix.row_ids.p;
    c.alive = d_alive;
    c.nbits = a.nbits;
    c.Q = rp.Q;
    c.ld4 = ld / 4;
    c.nq = (uint32_t)nq;
    c.probes = plan_probes; // (the lists the pre-pruning dropped provably hold none of the k nearest rows: not for the fallback either)
    c.list_off = ix.list_off.p;
    c.nprobe = (uint32_t)nprobe;
    c.nlist = (uint32_t)ix.nlist;
    c.k = k;
    c.partial = partial1;
    c.rows_per_block = pl.rpb1;
    c.seg_max = pl.seg_max1;
    c.qmap = failq2;
    c.qcount = nfail_final;
    IvfMergeParams fm{};
    fm.partial = partial1;
    fm.probes = plan_probes;
    fm.list_off = ix.list_off.p;
    fm.nprobe = (uint32_t)nprobe;
    fm.seg_max = pl.seg_max1;
    fm.rows_per_block = pl.rpb1;
    fm.k = k;
    fm.out_ids = d_ids;
    fm.out_dis = d_dis;
    fm.cosine = ix.metric == MSVS_METRIC_COSINE;
    fm.qmap = failq2;
    fm.qcount = nfail_final;
    run_fallback_rounds(scan_metric(m), c, fm, nq, fb_cap, pl.fb_slots, stream);
    if (signalled)
        MSVS_HIP(hipStreamSynchronize(stream));
}

/// given_probes (nullable): [nq][nprobe] list ids computed elsewhere (another rank's share of the coarse quantiser):
/// step 1 is skipped.  probes_only (nullable): run ONLY step 1 and leave the probe lists there.
static void index_search_device_one(const msvs_index & ix, const float * d_queries, size_t nq, uint32_t k, size_t nprobe,
                                    const uint64_t * d_alive, size_t nbits, int64_t * d_ids, float * d_dis,
                                    hipStream_t stream, const int32_t * given_probes = nullptr,
                                    int32_t * probes_only = nullptr, const SearchView * view = nullptr, ProbeWords words = ProbeWords{});

/// The search proper: all pointers on the device, everything enqueued on `stream`.  Very large batches are cut into
/// sub-batches of at most 2^21 (query, probe) pairs, stream-ordered one after the other: every scratch buffer of a search
/// is proportional to the pairs of ONE sub-batch, so the per-(thread, stream) arena stays bounded (~0.5 GB) whatever nq.
void index_search_device(const msvs_index & ix, const float * d_queries /* nq x dim, dense */, size_t nq, uint32_t k, size_t nprobe,
                         const uint64_t * d_alive, size_t nbits, int64_t * d_ids, float * d_dis, hipStream_t stream,
                         const int32_t * given_probes, int32_t * probes_only, const SearchView * view, ProbeWords words)
{
    const size_t np_eff = ix.type == MSVS_INDEX_IVFFLAT ? std::max<size_t>(1, std::min(nprobe, std::max<size_t>(ix.nlist, 1))) : 1;
    const size_t sub = std::max<size_t>(256, ((size_t)1 << 21) / np_eff);
    if (nq <= sub)
        return index_search_device_one(ix, d_queries, nq, k, nprobe, d_alive, nbits, d_ids, d_dis, stream, given_probes,
                                       probes_only, view, words);
    for (size_t q0 = 0; q0 < nq; q0 += sub)
    {
        const size_t m = std::min(sub, nq - q0);
        ProbeWords w = words;  // (the index-wide fields -- g_radius, g_list_off, g_xmax, g_xmin -- stay; the per-pair ones move with q0)
        w.given = words.given ? words.given + q0 * np_eff : nullptr;
        w.out = words.out ? words.out + q0 * np_eff : nullptr;
        w.pruned_out = words.pruned_out ? words.pruned_out + q0 * np_eff : nullptr;
        index_search_device_one(ix, d_queries + q0 * ix.dim, m, k, nprobe, d_alive, nbits, d_ids ? d_ids + q0 * k : nullptr,
                                d_dis ? d_dis + q0 * k : nullptr, stream, given_probes ? given_probes + q0 * np_eff : nullptr,
                                probes_only ? probes_only + q0 * np_eff : nullptr, view, w);
    }
}

static void index_search_device_one(const msvs_index & ix, const float * d_queries /* nq x dim, dense */, size_t nq,
                                    uint32_t k, size_t nprobe, const uint64_t * d_alive, size_t nbits, int64_t * d_ids,
                                    float * d_dis, hipStream_t stream, const int32_t * given_probes, int32_t * probes_only,
                                    const SearchView * view, ProbeWords words)
{
    if (!ix.ready)
        fail(MSVS_ERR_NOT_READY, "index is not ready");
    if (nq == 0 || k == 0)
        return;
    check_k(k);
    if (ix.type == MSVS_INDEX_IVFFLAT)
    {
        if (nprobe == 0)
            nprobe = 1;
        nprobe = std::min(nprobe, ix.nlist);
        check_k(nprobe);
    }
    const uint32_t d = (uint32_t)ix.dim, ld = ix.ld;
    Scratch & scr = scratch_for(stream);
    scr.reserve(index_search_scratch(ix, nq, k, nprobe, view == nullptr), stream);
    // queries: pad and/or normalise into scratch when needed
    const float * dq = d_queries;
    if (ld != d || ix.metric == MSVS_METRIC_COSINE)
    {
        float * q2 = scr.take<float>(nq * ld);
        upload_rows(q2, d_queries, nq, d, ld, MSVS_MEM_DEVICE, stream);
        if (ix.metric == MSVS_METRIC_COSINE)
            normalize_device_rows(q2, nq, d, ld, stream);
        dq = q2;
    }
    MergeParams out{};
    out.out_ids = d_ids;
    out.out_dis = d_dis;
    out.cosine = ix.metric == MSVS_METRIC_COSINE;
    const int m = ix.metric == MSVS_METRIC_L2 ? MSVS_METRIC_L2 : MSVS_METRIC_IP;
    if (ix.type == MSVS_INDEX_FLAT)
    {
        // round 5: a FEW queries over a large table take the fp16 shadow too -- the canonical scan moves 4 B per element, the shadow
        // pass 2 B plus its ~8 small launches (~80 us): it pays from ~128 M elements on for 1-4 queries, earlier when the canonical
        // scan needs several query tiles (VIWithDataPart.cpp:922-926 with IndexType::FLAT: one query per call)
        const bool shadow_few = !view && nq < 16 && options().flat_few != 0 && options().flat_mfma != 0 && options().flat_h16 != 0
            && ix.shadow_ready && ix.xnorm.p && ix.xnorm_max < 1e30f && k <= 40 && ix.n >= 256 && ix.n <= 0xfffffff0ull
            && (double)ix.n * (double)ix.dim >= (options().flat_few >= 2 ? 0.0 : nq <= 4 ? 128e6 : 32e6)
            && h16_lds_bytes(1, ix.h_nch) <= 160 * 1024;
        if (!view && (table_pass_eligible(ix.n, ix.xnorm.p, ix.xnorm_max, nq, k, options().flat_mfma) || shadow_few))
        {
            // a batch against the whole table: matrix-core candidate pass + canonical re-rank (exact, certified)
            TablePass t{};
            t.rows = ix.vecs.p;
            t.ids = ix.row_ids.p;
            t.norms = ix.xnorm.p;
            t.norm_max = ix.xnorm_max;
            t.n = ix.n;
            t.alive = d_alive;
            t.nbits = nbits;
            t.k = k;
            t.out_ids = d_ids;
            t.out_dis = d_dis;
            t.cosine = ix.metric == MSVS_METRIC_COSINE;
            t.flat_h16 = ix.shadow_ready && options().flat_h16 != 0 && h16_lds_bytes(1, ix.h_nch) <= 160 * 1024;
            t.h16_rho = ix.h_rho;
            t.prof_name = "flat_pass";
            table_candidate_pass(ix, scr, m, dq, nq, t, stream);
            g_prefilter_queries.fetch_add(nq, std::memory_order_relaxed);
            return;
        }
        flat_search_device(scr, m, ix.vecs.p, ix.row_ids.p, ix.n, ld, dq, nq, k, d_alive, nbits, out, stream, view);
        return;
    }
    // 1. coarse quantiser: exact top-nprobe of the centroids (canonical arithmetic, so probes match the oracle)
    H16Queries prepared{};
    prepared.n_counters = (uint32_t)(4 * ix.nlist + 2 + 16); // the shadow list scan's counters (h16_list_scan)
    prepared.counters = scr.take<uint32_t>(prepared.n_counters);
    int32_t * d_probes = probes_only ? probes_only : scr.take<int32_t>(nq * nprobe);
    if (given_probes)
        MSVS_HIP(hipMemcpyAsync(d_probes, given_probes, nq * nprobe * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
    else if (table_pass_eligible(ix.nlist, ix.cnorm.p, ix.cnorm_max, nq, (uint32_t)nprobe, options().coarse_mfma)
             || coarse_shadow_eligible(ix, nq, nprobe))
    {
        TablePass t{};
        // the centroid shadow: every approximate distance of the batch in one LDS-free MFMA launch (h16_scan_kernels.hpp)
        t.h16 = ix.c_shadow_ready && options().coarse_h16 != 0 && nq * round_up(ix.nlist, (size_t)H_ROWS) * 4 <= ((size_t)128 << 20);
        t.h16_rho = ix.c_rho;
        t.rows = ix.centroids.p;
        t.norms = ix.cnorm.p;
        t.norm_max = ix.cnorm_max;
        t.n = ix.nlist;
        t.k = (uint32_t)nprobe;
        t.out_probes = d_probes;
        t.h16_out = &prepared;
        t.prof_name = "coarse_pass";
        table_candidate_pass(ix, scr, m, dq, nq, t, stream);
    }
    else
    {
        MergeParams co{};
        co.mode = 1;
        co.out_probes = d_probes;
        // (a routed sharded search's front phase -- probes_only + pruned_out -- pre-prunes by these distances too)
        if ((!probes_only || words.pruned_out) && (ix.metric == MSVS_METRIC_L2 || ix.metric == MSVS_METRIC_COSINE) && options().h16_preprune != 0)
        {
            float * pd = scr.take<float>(nq * nprobe);
            co.out_probe_dis = pd;
            prepared.probe_dis = pd;
        }
        // a small batch: one self-merging launch (search_entry.hip: coarse_few_launch); a sharded search's probe lists keep the
        // sorted form the scan + merge launches leave
        if (probes_only || !coarse_few_launch(ix, dq, nq, nprobe, d_probes, co.out_probe_dis, stream))
            flat_search_device(scr, m, ix.centroids.p, nullptr, ix.nlist, ld, dq, nq, (uint32_t)nprobe, nullptr, 0, co, stream);
    }
    // 1b. pre-pruning by the list radius alone (h16_preprune_kernel): L2 and cosine indexes, unfiltered searches over the stored lists,
    // whenever the coarse stage left distances behind -- approximate words (centroid shadow; a sharded search's probe words) or the
    // canonical values of a small batch.  What it drops is gone for every path below, the canonical ones and the fallbacks included.
    // radius / list_off: the index's own lists, or (the routed sharded search) those of the whole index.
    auto preprune = [&](const int32_t * in_probes, const float * radius, const int64_t * list_off, int32_t * out_probes, bool keep_upre, float xmax,
                        float xmin) -> bool {
        if (!(options().h16_prune != 0 && options().h16_preprune != 0 && (ix.metric == MSVS_METRIC_L2 || ix.metric == MSVS_METRIC_COSINE) && !d_alive
              && !view && radius && nprobe <= 64 && nprobe >= 2 && ix.cnorm_max < 1e30f && xmax < 1e30f
              && (prepared.probe_dis || ((prepared.coarse_words || prepared.probe_words) && prepared.qnorm))))
            return false;
        H16Prune pr0{};
        set_prune_error_model(pr0, ix, prepared.qh ? prepared.qrho : nullptr, !prepared.coarse_words && prepared.probe_words != nullptr,
                              radius != ix.list_radius.p);
        if (prepared.coarse_words || prepared.probe_words)
        {
            pr0.coarse_words = prepared.coarse_words;
            pr0.npad = prepared.coarse_npad;
            pr0.probe_words = prepared.coarse_words ? nullptr : prepared.probe_words;
            pr0.qnorm = prepared.qnorm;
        }
        else
            pr0.probe_dis = prepared.probe_dis;
        pr0.radius = radius;
        pr0.xmax = xmax;
        pr0.cmax = ix.cnorm_max;
        const bool cosine_form = ix.metric == MSVS_METRIC_COSINE;
        if (cosine_form)
        {
            pr0.ip = 1;
            pr0.cnorm = ix.cnorm.p;
            pr0.xmin = xmin;
            pr0.Q = dq; // (normalised above)
            pr0.ldq = ld;
        }
        pr0.k = k;
        pr0.stat = options().rerank_stats != 0 ? prefilter_fail_counter() + 8 : nullptr;
        pr0.upre = cosine_form || !keep_upre ? nullptr : scr.take<float>(nq); // (an L2 bound: the cosine scan's second stage compares inner products)
        ProfileScope prof("ivf_plan", stream);
        hipLaunchKernelGGL(h16_preprune_kernel, dim3((unsigned)ceil_div(nq, (size_t)4)), dim3(BLOCK), 0, stream, in_probes, pr0, list_off,
                           (uint32_t)nq, (uint32_t)nprobe, out_probes);
        MSVS_HIP(hipGetLastError());
        if (keep_upre)
        {
            prepared.upre = pr0.upre;
            prepared.prepruned = true;
        }
        return true;
    };
    if (probes_only)
    {
        if (words.out)
        {
            // the words of the probes, for the rank that scans (ProbeWords); without a shadow coarse pass there are none
            if (prepared.coarse_words)
                hipLaunchKernelGGL(gather_probe_words_kernel, dim3((unsigned)ceil_div(nq * nprobe, (size_t)256)), dim3(256), 0, stream,
                                   probes_only, prepared.coarse_words, prepared.coarse_npad, (uint32_t)nprobe, nq * nprobe, words.out);
            else
                MSVS_HIP(hipMemsetAsync(words.out, 0xFF, nq * nprobe * 4, stream));
        }
        if (words.pruned_out) // the routed sharded search: which probes (of ANY rank's lists) are worth a visit
        {
            // (xnorm_max / xnorm_min of a shard are those of its own rows: the routed entry refuses indexes whose bounds were not made
            // global -- see shard.hip)
            if (!(words.g_radius && words.g_list_off && preprune(probes_only, words.g_radius, words.g_list_off, words.pruned_out, false, words.g_xmax, words.g_xmin)))
                MSVS_HIP(hipMemcpyAsync(words.pruned_out, probes_only, nq * nprobe * 4, hipMemcpyDeviceToDevice, stream));
        }
        return;
    }
    if (given_probes)
    {
        prepared.probe_words = words.given;
        prepared.prepruned = words.given_pruned; // (a routed search: the sending rank dropped what it could; the plan's own count tells the next search)
    }
    const int32_t * all_probes = d_probes;
    (void)all_probes;
    {
        int32_t * probes1 = scr.take<int32_t>(nq * nprobe);
        if (preprune(d_probes, ix.list_radius.p, ix.list_off.p, probes1, true, ix.xnorm_max, ix.xnorm_min))
            d_probes = probes1;
    }
    // 2. scan the probed lists
    if (nq * nprobe > 0x7fffffffull)
        fail(MSVS_ERR_INVALID_ARGUMENT, "nq * nprobe too large for one call");
    // a compacted view is scanned canonically: the shadow / split-bf16 passes walk the stored 32-row blocks
    const IvfSearchPlan pl = plan_ivf(ix, nq, nprobe, k, view == nullptr);
    const int64_t * list_off = view ? view->list_off : ix.list_off.p;
    uint64_t * partial = scr.take<uint64_t>(pl.mfma() ? nq * (pl.h16 ? (size_t)pl.h_cap : big_cand_cap(nprobe, pl.seg_max))
                                                      : nq * nprobe * (size_t)pl.seg_max * k);
    ScanParams a{};
    a.Y = reinterpret_cast<const float4 *>(ix.vecs.p);
    a.ids = ix.row_ids.p;
    a.alive = d_alive;
    a.nbits = (uint32_t)std::min<size_t>(nbits, 0xffffffffu);
    a.Q = reinterpret_cast<const float4 *>(dq);
    a.partial = partial;
    a.ld4 = ld / 4;
    a.k = k;
    a.nq = (uint32_t)nq;
    a.rows_per_block = pl.rpb;
    a.probes = d_probes;
    a.list_off = list_off;
    a.rowmap = view ? view->rowmap : nullptr;
    a.nprobe = (uint32_t)nprobe;
    a.seg_max = pl.seg_max;
    a.nlist = (uint32_t)ix.nlist;
    if (pl.mfma() && pl.h16)
    {
        h16_list_scan(ix, scr, m, dq, nq, k, nprobe, pl, d_probes, d_alive, nbits, partial, d_ids, d_dis, stream, prepared);
        return;
    }
    if (pl.mfma())
    {
        // many queries per list: matrix-core candidate pass, canonical re-rank, certified (mfma_scan_kernels.hpp).
        // Two phases: the first 128 rows of every probed list are a SAMPLE of the rows the query will see; its m-th
        // best candidate becomes the query's cut for the rest (about m / sample-fraction rows of everything lie below
        // it), which keeps the candidate buffers and the selection work small whatever the data looks like.
        IvfPlanParams pp{};
        pp.probes = d_probes;
        pp.list_off = ix.list_mid.p;       // phase B: rows [mid, end) of every list
        pp.list_end = ix.list_off.p + 1;
        pp.whole_off = ix.list_off.p;
        pp.n_pairs = (uint32_t)(nq * nprobe);
        pp.nlist = (uint32_t)ix.nlist;
        pp.rows_per_block = pl.rpb;
        pp.T = pl.T;
        uint32_t * counters = scr.take<uint32_t>(2 * ix.nlist + 1);
        pp.cnt = counters;
        pp.fill = counters + ix.nlist;
        uint32_t * nfail = counters + 2 * ix.nlist;
        pp.pair_off = scr.take<uint32_t>(ix.nlist + 1);
        pp.work_off = scr.take<uint32_t>(ix.nlist + 1);
        pp.pairs = scr.take<uint32_t>(nq * nprobe);
        MSVS_HIP(hipMemsetAsync(counters, 0, (2 * ix.nlist + 1) * sizeof(uint32_t), stream));
        launch_ivf_plan(pp, stream);
        IvfPlanParams pa = pp;             // phase A: rows [begin, mid), one 128-row work item per (list, query tile)
        pa.list_off = ix.list_off.p;
        pa.list_end = ix.list_mid.p;
        pa.rows_per_block = BG_ROWS;
        pa.work_off = scr.take<uint32_t>(ix.nlist + 1);
        launch_ivf_plan_rescan(pa, stream);
        float * qnorm = scr.take<float>(nq);
        launch_row_sqnorm(dq, qnorm, nq, ld / 4, nullptr, stream);
        float4 * qsplit = scr.take<float4>(nq * (size_t)ceil_div((size_t)ld / 4, (size_t)8) * 8);
        launch_split_queries(dq, (uint32_t)nq, ld / 4, qsplit, stream);
        a.Qsplit = qsplit;
        a.pairs = pp.pairs;
        a.pair_off = pp.pair_off;
        a.xcd_order = (uint32_t)options().ivf_xcd;
        a.k = pl.kc;
        a.qnorm = qnorm;
        a.xnorm = ix.xnorm.p;
        // capacity of a query's candidate buffer (= `partial`): the worst case (every slice full) up to MSVS_CAND_CAP
        // keys; beyond that the buffer may overflow, which only costs the query its certificate
        const size_t cand_cap = big_cand_cap(nprobe, pl.seg_max);
        // per query: running cut (0xFFFFFFFF = none yet) and append cursor of its candidate buffer
        uint32_t * qstate = scr.take<uint32_t>(2 * nq);
        a.qthr = qstate;
        a.qcnt = qstate + nq;
        a.cand_cap = (uint32_t)cand_cap;
        MSVS_HIP(hipMemsetAsync(a.qthr, 0xFF, nq * sizeof(uint32_t), stream));
        MSVS_HIP(hipMemsetAsync(a.qcnt, 0, nq * sizeof(uint32_t), stream));
        uint64_t * cand = scr.take<uint64_t>(nq * (size_t)pl.kc);
        uint64_t * bound = scr.take<uint64_t>(nq);
        {
            ScanParams sa = a;
            sa.list_off = pa.list_off;
            sa.list_end = pa.list_end;
            sa.work_off = pa.work_off;
            sa.rows_per_block = BG_ROWS;
            launch_ivf_mfma_scan(scan_metric(m), pl.nqg, pl.grid, sa, stream, "ivf_sample_scan", false);
            launch_cand_select(partial, a.qcnt, a.qthr, a.cand_cap, (uint32_t)nq, pl.kc, cand, bound, stream);
            // m-th sample candidate: ~8k rows of everything below the cut (see sample_cut_kernel, table pass)
            const size_t avg_len = std::max<size_t>(1, ix.n / std::max<size_t>(ix.nlist, 1));
            const size_t mth = 2 + ceil_div((size_t)8 * k * std::min<size_t>(avg_len, BG_ROWS), avg_len);
            launch_sample_cut(cand, pl.kc, (uint32_t)std::min<size_t>(pl.kc, std::max<size_t>(6, mth)), (uint32_t)nq,
                              a.qthr, stream);
        }
        a.list_off = pp.list_off;
        a.list_end = pp.list_end;
        a.work_off = pp.work_off;
        launch_ivf_mfma_scan(scan_metric(m), pl.nqg, pl.grid, a, stream);
        launch_cand_select(partial, a.qcnt, a.qthr, a.cand_cap, (uint32_t)nq, pl.kc, cand, bound, stream);
        uint32_t * failq = scr.take<uint32_t>(nq);
        RerankParams rp{};
        rp.Y = a.Y;
        rp.ids = ix.row_ids.p;
        rp.Q = a.Q;
        rp.qnorm = qnorm;
        rp.cand = cand;
        rp.bound = bound;
        rp.kc = pl.kc;
        rp.k = k;
        rp.ld4 = ld / 4;
        rp.out_ids = d_ids;
        rp.out_dis = d_dis;
        rp.cosine = ix.metric == MSVS_METRIC_COSINE;
        set_error_model(rp, ix.dim);
        rp.xmax = ix.xnorm_max;
        rp.failq = failq;
        rp.nfail = nfail;
        rp.early_exit = options().rerank_early != 0 ? 1 : 0;
        rp.stat_fail = prefilter_fail_counter();
        launch_ivf_rerank(scan_metric(m), rp, (uint32_t)nq, stream);
        g_prefilter_queries.fetch_add(nq, std::memory_order_relaxed);
        // queries without a certificate: canonical scan, one query per block (normally zero of them)
        const size_t fb_cap = fallback_cap(nq, nprobe, pl.seg_max1, k);
        uint64_t * partial1 = scr.take<uint64_t>(fb_cap * nprobe * (size_t)pl.seg_max1 * k);
        ScanParams c = a;
        c.list_off = ix.list_off.p; // the fallback scans whole lists
        c.list_end = nullptr;
        c.k = k;
        c.partial = partial1;
        c.rows_per_block = pl.rpb1;
        c.seg_max = pl.seg_max1;
        c.qmap = failq;
        c.qcount = nfail;
        IvfMergeParams fm{};
        fm.partial = partial1;
        fm.probes = d_probes;
        fm.list_off = ix.list_off.p;
        fm.nprobe = (uint32_t)nprobe;
        fm.seg_max = pl.seg_max1;
        fm.rows_per_block = pl.rpb1;
        fm.k = k;
        fm.out_ids = d_ids;
        fm.out_dis = d_dis;
        fm.cosine = ix.metric == MSVS_METRIC_COSINE;
        fm.qmap = failq;
        fm.qcount = nfail;
        run_fallback_rounds(scan_metric(m), c, fm, nq, fb_cap, pl.fb_slots, stream);
        return;
    }
    if (pl.T == 1)
    {
        // few queries: one (query, list, segment) per block, no grouping pass
        for (size_t q0 = 0; q0 < nq; q0 += 32768)
        {
            ScanParams c = a;
            c.nq = (uint32_t)std::min<size_t>(32768, nq - q0);
            c.Q = reinterpret_cast<const float4 *>(dq + q0 * ld);
            c.partial = partial + q0 * nprobe * (size_t)pl.seg_max * k;
            c.probes = d_probes + q0 * nprobe;
            launch_ivf_scan(scan_metric(m), c, stream);
        }
    }
    else
    {
        // group the (query, list) pairs by list, then one pass over each list segment per tile of T queries
        IvfPlanParams pp{};
        pp.probes = d_probes;
        pp.list_off = list_off;
        pp.whole_off = list_off;
        pp.n_pairs = (uint32_t)(nq * nprobe);
        pp.nlist = (uint32_t)ix.nlist;
        pp.rows_per_block = pl.rpb;
        pp.T = pl.T;
        uint32_t * counters = scr.take<uint32_t>(2 * ix.nlist);
        pp.cnt = counters;
        pp.fill = counters + ix.nlist;
        pp.pair_off = scr.take<uint32_t>(ix.nlist + 1);
        pp.work_off = scr.take<uint32_t>(ix.nlist + 1);
        pp.pairs = scr.take<uint32_t>(nq * nprobe);
        MSVS_HIP(hipMemsetAsync(counters, 0, 2 * ix.nlist * sizeof(uint32_t), stream));
        launch_ivf_plan(pp, stream);
        a.pairs = pp.pairs;
        a.pair_off = pp.pair_off;
        a.work_off = pp.work_off;
        a.xcd_order = (uint32_t)options().ivf_xcd; // experiment knob; default on
        launch_ivf_batched_scan(scan_metric(m), pl.T, pl.grid, a, stream);
    }
    // 3. per-query top-k over the valid segments of its probed lists
    IvfMergeParams im{};
    im.partial = partial;
    im.probes = d_probes;
    im.list_off = list_off;
    im.nprobe = (uint32_t)nprobe;
    im.seg_max = pl.seg_max;
    im.rows_per_block = pl.rpb;
    im.k = k;
    im.out_ids = d_ids;
    im.out_dis = d_dis;
    im.cosine = ix.metric == MSVS_METRIC_COSINE;
    launch_ivf_merge(scan_metric(m), im, (uint32_t)nq, stream);
}

}

/// The row-id maps of a decoupled part (SegmentId::getMergedMaps, VIWithDataPart.cpp:722): once set, a search takes its
/// filter in the MERGED part's row space (getRealBitmap) and reports the MERGED part's rows (transferToNewRowIds).
extern "C" int msvs_index_set_merged_maps(msvs_index_t * ix, const uint64_t * row_ids_map, size_t n_old,
                                          const uint64_t * inverted_row_ids_map, const uint8_t * inverted_row_sources_map,
                                          size_t n_new, uint32_t own_id)
{
    return guarded([&] {
        DeviceGuard on_device(ix ? ix->device : -1);
        if (!ix || (n_old && !row_ids_map) || (n_new && (!inverted_row_ids_map || !inverted_row_sources_map)))
            fail(MSVS_ERR_INVALID_ARGUMENT, "null index / map");
        auto cur = ix->get_meta();
        auto next = std::make_shared<msvs_index::Meta>();
        if (cur && cur->delete_alive.n)
        {
            next->delete_alive.alloc(cur->delete_alive.n);
            MSVS_HIP(hipMemcpy(next->delete_alive.p, cur->delete_alive.p, cur->delete_alive.bytes(), hipMemcpyDeviceToDevice));
            next->delete_nbits = cur->delete_nbits;
        }
        if (n_old)
        {
            next->row_ids_map.alloc(n_old);
            MSVS_HIP(hipMemcpy(next->row_ids_map.p, row_ids_map, n_old * 8, hipMemcpyHostToDevice));
            next->row_ids_n = n_old;
        }
        if (n_new)
        {
            next->inv_row_ids.alloc(n_new);
            next->inv_sources.alloc(n_new);
            MSVS_HIP(hipMemcpy(next->inv_row_ids.p, inverted_row_ids_map, n_new * 8, hipMemcpyHostToDevice));
            MSVS_HIP(hipMemcpy(next->inv_sources.p, inverted_row_sources_map, n_new, hipMemcpyHostToDevice));
            next->inv_n = n_new;
        }
        next->own_id = own_id;
        std::lock_guard<std::mutex> lk(ix->meta_mu);
        ix->meta = next;
    });
}

extern "C" int msvs_index_scanned_rows(const msvs_index_t * ix, const float * queries, size_t nq, int nprobe,
                                       uint64_t * rows, uint64_t * rows_streamed, uint64_t * rows_unique)
{
    return guarded([&] {
        DeviceGuard on_device(ix ? ix->device : -1);
        if (!ix || !rows || (nq && !queries))
            fail(MSVS_ERR_INVALID_ARGUMENT, "null argument");
        if (!ix->ready)
            fail(MSVS_ERR_NOT_READY, "index is not ready");
        *rows = 0;
        if (rows_streamed)
            *rows_streamed = 0;
        if (rows_unique)
            *rows_unique = 0;
        if (nq == 0)
            return;
        if (ix->type == MSVS_INDEX_FLAT)
        {
            *rows = (uint64_t)nq * ix->n;
            if (rows_streamed)
                *rows_streamed = (uint64_t)plan_flat(ix->n, nq, ix->ld / 4, 10).n_qtiles * ix->n;
            if (rows_unique)
                *rows_unique = ix->n;
            return;
        }
        size_t np = std::min<size_t>((size_t)std::max(nprobe, 1), ix->nlist);
        check_k(np);
        hipStream_t stream = nullptr;
        const uint32_t d = (uint32_t)ix->dim, ld = ix->ld;
        DevBuf<float> dq(nq * ld);
        DevBuf<int32_t> d_probes(nq * np);
        upload_rows(dq.p, queries, nq, d, ld, MSVS_MEM_HOST, stream);
        if (ix->metric == MSVS_METRIC_COSINE)
            normalize_device_rows(dq.p, nq, d, ld, stream);
        Scratch & scr = scratch_for(stream);
        scr.reserve(flat_scratch_bytes(ix->nlist, nq, (uint32_t)np, ix->ld) + 4096, stream);
        MergeParams co{};
        co.mode = 1;
        co.out_probes = d_probes.p;
        const int m = ix->metric == MSVS_METRIC_L2 ? MSVS_METRIC_L2 : MSVS_METRIC_IP;
        flat_search_device(scr, m, ix->centroids.p, nullptr, ix->nlist, ld, dq.p, nq, (uint32_t)np, nullptr, 0, co,
                           stream);
        std::vector<int32_t> h(nq * np);
        MSVS_HIP(hipMemcpyAsync(h.data(), d_probes.p, h.size() * 4, hipMemcpyDeviceToHost, stream));
        MSVS_HIP(hipStreamSynchronize(stream));
        uint64_t total = 0;
        std::vector<uint32_t> cnt(ix->nlist, 0);
        for (int32_t l : h)
            if (l >= 0)
            {
                total += (uint64_t)(ix->h_list_off[l + 1] - ix->h_list_off[l]);
                cnt[l]++;
            }
        *rows = total;
        if (rows_streamed)
        {
            const uint32_t T = plan_ivf(*ix, nq, np, 10).T;
            uint64_t st = 0;
            for (size_t l = 0; l < ix->nlist; l++)
                st += (uint64_t)ceil_div(cnt[l], T) * (uint64_t)(ix->h_list_off[l + 1] - ix->h_list_off[l]);
            *rows_streamed = st;
        }
        if (rows_unique)
        {
            uint64_t u = 0;
            for (size_t l = 0; l < ix->nlist; l++)
                if (cnt[l])
                    u += (uint64_t)(ix->h_list_off[l + 1] - ix->h_list_off[l]);
            *rows_unique = u;
        }
    });
}

extern "C" int msvs_profile_enable(int on)
{
    return guarded([&] { profile_enable(on != 0); });
}

extern "C" int msvs_profile_get(const char * name, uint64_t * calls, double * total_ms)
{
    return guarded([&] {
        if (!name)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null name");
        profile_get(name, calls, total_ms);
    });
}

extern "C" int msvs_prefilter_stats(uint64_t * queries, uint64_t * fallbacks)
{
    return guarded([&] {
        unsigned long long f = 0;
        unsigned long long * p = prefilter_fail_counter();
        MSVS_HIP(hipDeviceSynchronize());
        MSVS_HIP(hipMemcpy(&f, p, 8, hipMemcpyDeviceToHost));
        if (queries)
            *queries = g_prefilter_queries.load();
        if (fallbacks)
            *fallbacks = f;
    });
}

/// Experiments (option rerank_stats = 1): out[0..3] = candidates an early exit could have skipped / candidates re-ranked, for the
/// result passes and for the coarse quantiser's passes; out[4], out[5] = rows really skipped (result / coarse passes).
extern "C" __attribute__((visibility("default"))) int msvs_debug_rerank_stats(uint64_t * out)
{
    return guarded([&] {
        unsigned long long v[6];
        MSVS_HIP(hipDeviceSynchronize());
        MSVS_HIP(hipMemcpy(v, prefilter_fail_counter() + 2, 48, hipMemcpyDeviceToHost));
        for (int i = 0; i < 6; i++)
            out[i] = v[i];
    });
}

/// Measurement (not in msvs.h; needs rerank_stats = 1): rows of the lists the shadow list scan's launches read, cumulative --
/// out2[0] = main launch (rows beyond block 0 of the lists with surviving pairs), out2[1] = sample launch (block 0 of the lists
/// probed after the pre-pruning).  bench.py prices these: what the launches move, not what the reference's scan would visit.
extern "C" __attribute__((visibility("default"))) int msvs_debug_scan_rows(uint64_t * out2)
{
    return guarded([&] {
        unsigned long long v[2];
        MSVS_HIP(hipDeviceSynchronize());
        MSVS_HIP(hipMemcpy(v, prefilter_fail_counter() + 10, 16, hipMemcpyDeviceToHost));
        out2[0] = v[0];
        out2[1] = v[1];
    });
}

/// Experiments / tests (not in msvs.h; needs rerank_stats = 1): (query, list) pairs the probe pruning dropped, pairs it looked at.
extern "C" __attribute__((visibility("default"))) int msvs_debug_prune_stats(uint64_t * out2)
{
    return guarded([&] {
        unsigned long long v[2];
        MSVS_HIP(hipDeviceSynchronize());
        MSVS_HIP(hipMemcpy(v, prefilter_fail_counter() + 8, 16, hipMemcpyDeviceToHost));
        out2[0] = v[0];
        out2[1] = v[1];
    });
}

/// The same counters for the coarse quantiser's candidate passes (batches of >= 512 queries on nlist >= 256).
extern "C" int msvs_coarse_stats(uint64_t * queries, uint64_t * fallbacks)
{
    return guarded([&] {
        unsigned long long f = 0;
        unsigned long long * p = prefilter_fail_counter();
        MSVS_HIP(hipDeviceSynchronize());
        MSVS_HIP(hipMemcpy(&f, p + 1, 8, hipMemcpyDeviceToHost));
        if (queries)
            *queries = g_coarse_queries.load();
        if (fallbacks)
            *fallbacks = f;
    });
}

/// Give the calling thread's scratch arenas back (they are grow-only per (thread, stream) otherwise, with a lazy shrink in
/// Scratch::reserve): for hosts that park client threads.
extern "C" int msvs_release_scratch(size_t * freed_bytes)
{
    return guarded([&] {
        const size_t f = release_thread_arenas();
        if (freed_bytes)
            *freed_bytes = f;
    });
}

extern "C" int msvs_set_option(const char * name, const char * value)
{
    return guarded([&] {
        if (!set_option(name, value))
            fail(MSVS_ERR_INVALID_ARGUMENT, "unknown option `%s`", name ? name : "(null)");
    });
}

extern "C" int msvs_profile_reset(void)
{
    return guarded([&] { profile_reset(); });
}



/// Tests only (not in msvs.h): the candidate keys of this thread's last shadow pass -- (ordered approximate distance << 32 |
/// stored row position) -- up to cap_out per query, and how many each query has.
extern "C" __attribute__((visibility("default"))) int msvs_debug_h16_keys(uint64_t * keys_out, size_t cap_out, uint32_t * counts_out,
                                                                         size_t nq)
{
    return guarded([&] {
        const H16Last & h = g_h16_last;
        if (!h.partial || nq != h.nq)
            fail(MSVS_ERR_INVALID_ARGUMENT, "no shadow pass of %zu queries on record", nq);
        MSVS_HIP(hipStreamSynchronize(h.stream));
        MSVS_HIP(hipMemcpy(counts_out, h.qcnt, nq * 4, hipMemcpyDeviceToHost));
        const size_t take = std::min<size_t>(cap_out, h.cap);
        MSVS_HIP(hipMemcpy2D(keys_out, cap_out * 8, h.partial, (size_t)h.cap * 8, take * 8, nq, hipMemcpyDeviceToHost));
        g_h16_last = H16Last{}; // one fetch per pass: a later search that takes another path must not be mistaken for it
    });
}

/// Experiments only (not in msvs.h; option h16_stamps): out[grid][64][4] of the last shadow main launch, *grid its workgroups.
extern "C" __attribute__((visibility("default"))) int msvs_debug_h16_stamps(uint64_t * out, size_t cap_words, uint32_t * grid)
{
    return guarded([&] {
        MSVS_HIP(hipDeviceSynchronize());
        const size_t words = (size_t)g_h16_stamp_grid * H_STAMP_ITEMS * 4;
        *grid = g_h16_stamp_grid;
        if (words && cap_words >= words)
            MSVS_HIP(hipMemcpy(out, g_h16_stamps.p, words * 8, hipMemcpyDeviceToHost));
    });
}

/// Tests only (not in msvs.h): the approximate distance words of this thread's last centroid-shadow coarse pass, [nq][npad]
/// (ordered words: f2ord(distance) for L2, ~f2ord(inner product) otherwise; 0xFFFFFFFF past the last centroid).
extern "C" __attribute__((visibility("default"))) int msvs_debug_coarse_words(uint32_t * out, size_t nq, size_t npad)
{
    return guarded([&] {
        const CoarseLast & c = g_coarse_last;
        if (!c.words || c.nq != nq || c.npad != npad)
            fail(MSVS_ERR_INVALID_ARGUMENT, "no coarse pass of %zu queries x %zu words on record (last: %u x %u)", nq, npad, c.nq, c.npad);
        MSVS_HIP(hipStreamSynchronize(c.stream));
        MSVS_HIP(hipMemcpy(out, c.words, nq * npad * 4, hipMemcpyDeviceToHost));
        g_coarse_last = CoarseLast{};
    });
}

/// Tests only: the constants of the certificate's error model for the shadow pass at this dimension.
extern "C" __attribute__((visibility("default"))) void msvs_debug_error_model_h16(size_t dim, double * c_dot, double * c_norm,
                                                                                 double * c_canon)
{
    RerankParams rp{};
    set_error_model_h16(rp, dim);
    *c_dot = rp.c_dot;
    *c_norm = rp.c_norm;
    *c_canon = rp.c_canon;
}

/// Tests only: the measured form of the same model for an index's rows (centroids = 0) or its centroid shadow (1): the table's rho,
/// the coefficient of |x||q| without the query's share, and the factor of the query image's own rho (set_error_model_h16).
extern "C" __attribute__((visibility("default"))) void msvs_debug_error_model_h16_measured(const msvs_index_t * ix, int centroids,
                                                                                          double * rho_table, double * c_dot_table,
                                                                                          double * qrho_scale)
{
    static const float one_query_rho = 0.f; // (any non-null pointer selects the per-query form)
    RerankParams rp{};
    const float rho = centroids ? ix->c_rho : ix->h_rho;
    set_error_model_h16(rp, ix->dim, rho, &one_query_rho);
    *rho_table = (double)rho;
    *c_dot_table = rp.c_dot;
    *qrho_scale = rp.qrho ? rp.qrho_scale : 0.0;
}
