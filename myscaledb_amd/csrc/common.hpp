// common.hpp -- host-side plumbing shared by the libmsvs translation units:
// error reporting (thread-local message + status code), RAII device buffers, small param parser.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/msvs.h"

namespace msvs
{

struct Error
{
    int code;
    std::string msg;
};

void set_last_error(const std::string & m);

[[noreturn]] inline void fail(int code, const char * fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw Error{code, buf};
}

#define MSVS_HIP(expr)                                                                                              \
    do                                                                                                              \
    {                                                                                                               \
        hipError_t e_ = (expr);                                                                                     \
        if (e_ != hipSuccess)                                                                                       \
            ::msvs::fail(e_ == hipErrorOutOfMemory ? MSVS_ERR_OUT_OF_MEMORY : MSVS_ERR_DEVICE, "%s failed: %s (%s:%d)", \
                         #expr, hipGetErrorString(e_), __FILE__, __LINE__);                                         \
    } while (0)

/// The host-pointer entry points take an index from ANY host thread (the reference searches from a pool of scan threads:
/// MergeTreeVSManager.cpp:973), and a fresh thread's current device is 0: on a multi-GPU host the call runs on the device the index
/// lives on, and the caller's device is what it was when the call returns.  (Device-pointer entries take the caller's stream: the
/// caller's device is the right one by construction.)
struct DeviceGuard
{
    int prev = -1;
    explicit DeviceGuard(int dev)
    {
        int cur = -1;
        if (dev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != dev)
        {
            MSVS_HIP(hipSetDevice(dev));
            prev = cur;
        }
    }
    ~DeviceGuard()
    {
        if (prev >= 0)
            (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard & operator=(const DeviceGuard &) = delete;
};

/// Runs a C-ABI body, translating exceptions into status codes + msvs_last_error().
template <typename F>
inline int guarded(F && f)
{
    try
    {
        f();
        return MSVS_OK;
    }
    catch (const Error & e)
    {
        set_last_error(e.msg);
        return e.code;
    }
    catch (const std::bad_alloc &)
    {
        set_last_error("host allocation failed");
        return MSVS_ERR_OUT_OF_MEMORY;
    }
    catch (const std::exception & e)
    {
        set_last_error(e.what());
        return MSVS_ERR_DEVICE;
    }
}

/// Non-owning device pointer with DevBuf's `.p` (arena slices used where a DevBuf used to be).
template <typename T>
struct DevView
{
    T * p;
};

/// Owning device allocation.
template <typename T>
struct DevBuf
{
    T * p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf &) = delete;
    DevBuf & operator=(const DevBuf &) = delete;
    DevBuf(DevBuf && o) noexcept : p(o.p), n(o.n)
    {
        o.p = nullptr;
        o.n = 0;
    }
    DevBuf & operator=(DevBuf && o) noexcept
    {
        if (this != &o)
        {
            release();
            p = o.p;
            n = o.n;
            o.p = nullptr;
            o.n = 0;
        }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t count)
    {
        release();
        n = count;
        if (count)
            MSVS_HIP(hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T)));
    }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    size_t bytes() const { return n * sizeof(T); }
};

inline size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }
inline size_t ceil_div(size_t x, size_t m) { return (x + m - 1) / m; }

/// "k=v,k=v" or a flat JSON object {"k":"v","k2":3}: enough for the reference's Search::Parameters
/// (string->string map, VICommon.h:127,186-212).
std::map<std::string, std::string> parse_params(const char * s);
long param_int(const std::map<std::string, std::string> & m, const char * key, long dflt);

/// Experiment / test knobs (never needed in production).  Read ONCE per process from the MSVS_<NAME> environment
/// variables and changed afterwards only through msvs_set_option(): a search never calls getenv.
struct Options
{
    double ivf_pass = 1;      // candidate pass of the list scan: 0 never, 1 from ~2 queries per list on, 2 whenever eligible
    double ivf_h16 = 1;       // 1: over the fp16 shadow (h16_scan_kernels.hpp); 0: split-bf16 over the f32 rows
    double coarse_h16 = 1;    // coarse quantiser of batches through the centroid shadow (0: the split-bf16 table pass)
    double wave_select = 1;   // candidate selection by the bitwise wave search (0: WaveTopK insertion kernels)
    double plan_lds = 1;      // plan histogram / scatter aggregated in LDS per 2048 pairs (0: one global atomic per pair)
    double bm25_fine_sample = 1; // BM25 wave scorer: the sample pass walks items of spi / 8 sub-ranges (0: the emit pass's items)
    double lat_select = 1;    // few-query path: probe list by register selection in the last block of stage 1 (0: list merge)
    double rerank_early = 1;  // re-rank of result passes: sorted candidates, skip those beyond the exact k-th of the first rounds +- eps
    double rerank_groups = 16; // candidate rows in flight per re-rank block (32: 512-thread blocks, measured slower: 70 vs 47 us)
    double rerank_stats = 0;  // experiments: count the candidates an early exit of the re-rank could skip (msvs_debug_rerank_stats)
    double combine = 8;       // msvs_index_search: single-query callers beyond this many in flight are batched by the next finisher (0: off)
    double combine_spin = 0;  // the worker thread that runs the combined batches polls this many microseconds for the next one before it sleeps
                              // (under load it collects the next batch itself; polling measured no gain: 166 k QPS either way at 64 callers)
    double combine_batches = 1; // ... and at most this many combined batches in flight (2: measured slower -- a batch of any size up to 64 costs the device the same ~0.2 ms and two of them do not overlap: the larger the batches the better)
    double h16_k128 = 1;      // shadow pass for 40 < k <= 128 with 256 candidates (0: the canonical scan as before)
    double coarse_h16_min_q = 192; // coarse quantiser through the centroid shadow from this many queries on (0: only with 128-query
                                   // tiles, ~1000 queries); measured on nlist 1024: 512 queries -6 %, 256 -2.6 %, 64 +10 % (ten launches)
    double h16_target = 0;    // rows of a query's probed lists the sample cut aims to keep (0: 15 k up to nprobe 64, 25 k beyond; 10 k beyond k = 40)
    double h16_kc = 0;        // candidates re-ranked per query after the fp16-shadow list scan (0: 32 for k <= 12, else 64)
    double coarse_kc = 0;     // ... after the centroid-shadow pass
    double fb_segs = 0;       // segments per list of the canonical fallback scan (0: automatic 4 / 16)
    double coarse_mfma = 1;   // same switch for the coarse quantiser ...
    double flat_mfma = 1;     // ... and for FLAT batches
    double ivf_nqg = 1;       // split-bf16 pass: 128- (1) or 256-query (2) tiles
    double ivf_rpb = 0;       // rows per work item (0 = planned)
    double ivf_grid = 0;      // grid size (0 = planned)
    double ivf_t = 0;         // canonical batched scan: query tile (0 = planned)
    double ivf_xcd = 1;       // XCD-contiguous work ranges
    double cand_cap = 0;      // candidate-buffer capacity per query (0 = planned; small values force overflow)
    double ivf_eps_scale = 1; // multiplies the certificate's error bound (1e12: every query takes the fallback)
    double h16_grid = 0;      // shadow pass: grid size (0 = planned)
    double h16_min_pairs = 0.25; // shadow pass from this many (query, list) pairs per list on
    double fb_cap = 0;        // queries per round of the canonical fallback (0 = by memory; small values: many rounds)
    double h16_nocut = 0;     // shadow pass: no sample cut, every probed row becomes a candidate (tests)
    double h16_ncb = 0;       // shadow pass: column blocks (32 queries each) per tile, 0 = planned
    double rerank_fused = 1;  // shadow list scan: the second chance runs inside the re-rank launch, by the block of the query that failed (0: a launch of its own)
    double rerank_second = 1; // queries whose first certificate fails get their whole candidate buffer re-ranked before the canonical scan
    double h16_segs = 1;      // shadow list scan: lists cut into row segments on the device so that a launch has ~4 items per workgroup; 0: one item per (list, tile)
    double h16_rho = 1;       // fp16 shadow passes: error bound from the MEASURED rounding error of the stored rows and of each query image; 0: the worst case per element
    double flat_h16 = 1;      // FLAT batches through the index's fp16 shadow (h16_flat_kernel); 0: the split-bf16 pass over the f32 rows; 3: one row block per wavefront
    double flat_ncb = 0;      // FLAT shadow pass: column blocks per tile (0: by batch size)
    double flat_lazy_flush = 1; // FLAT shadow pass, several tiles: a wavefront's survivors leave its LDS stage when it is full or the item ends (0: after every block)
    double flat_rot = 0;      // FLAT shadow pass, several tiles: tile t starts flat_rot * t rounds into its segment and wraps around (0: all tiles in step)
    double flat_segb = 0;     // FLAT shadow pass: blocks per segment (0: 64 / 128 / 256 for one / several / many tiles)
    double h16_stamps = 0;    // experiments: the main launch records per-item wall-clock stamps (msvs_debug_h16_stamps)
    double lat_path = 1;      // few-query IVFFLAT searches in two self-merging launches (latency_kernels.hpp): 0 off,
                              // 1 for 1-2 queries per call, 2 up to 4
    double filter_compact_below = -1;  // filtered searches run over a compacted view when less than this fraction of the
                                       // rows passes the filter (1: always, 0: never, < 0: by batch size,
                                       // profiles/r02_filter.txt)
    double lat_hint = 1;      // few-query path: stage 2's grid sized by the rows the last call left after the radius pruning (0: always two blocks per CU)
    double lat_items = 0;     // few-query path: work items of stage 2 per query (0 = planned: two blocks per CU over the call)
    double lat_prune = 1;     // few-query path (L2, no filter): probed lists the list radius rules out get no work items (0: off)
    double h16_preprune = 1;  // shadow list scan (L2, no filter): pairs the list radius alone rules out leave before the sample launch (0: off)
    double h16_group_appends = 32; // shadow passes: tiles of at most this many queries append their survivors with one atomic per (wavefront, query) (0: one per record)
    double h16_feedback = 1;  // shadow list scan: the second pruning stage is skipped while the last search of the index (same batch shape) came out of the
                              // pre-pruning with too few pairs for it to pay (0: decided from nq * nprobe alone, as before round 6)
    double h16_prune = 1;     // shadow list scan (L2): drop (query, list) pairs that provably cannot hold one of the query's k nearest rows when the lists are probed by more than a tile of queries (0: off, 2: always)
    double coarse_gemm_tq = 64; // coarse_gemm_kernel: queries per workgroup tile (64 or 128)
    double coarse_gemm_tc = 128; // ... and centroids (64 or 128)
    double coarse_slow_inline = 1; // coarse_tail_kernel: a query without a band computed exactly by its own wavefront while such queries are rare (0: always the queue + its three launches)
    double coarse_slow_window = 64; // ... rare = none in this many batched searches of the index
    double route_streams = 1; // msvs_shard_search_routed_device_async: 1 = the exchanges in the compute stream's order; 2 = on their own stream (four event hand-overs
                              // per step, and -- FRONT(i) being enqueued ahead of BACK(i - 1) -- nothing they could overlap with: 0.70 against 0.63 ms on one rank)
    double merge_small = 1;   // msvs_merge_topk_device: nparts * k <= 256 keys per query merged by one wavefront (0: pack + block merge)
    double coarse_tail = 1;   // coarse quantiser of batches: selection + band re-rank in one launch, a wavefront per query (0: coarse_select_kernel + ivf_rerank_kernel)
    double coarse_band = 1;   // coarse quantiser of batches: only the candidates near the top-nprobe boundary are evaluated canonically (0: all 64)
    double rerank_hint = 1;   // second-chance re-rank: skip rows that cannot beat the first stage's k-th exact distance (0: evaluate the whole buffer)
    double bm25_wave = 1;     // BM25: wave-private streaming scorer (1) or the barrier-synchronised block scorer (0)
    double bm25_posting = 1;  // BM25: posting-as-unit scorer (bm25p_kernel) for batches of sparse terms; 0: always the dense accumulator, 2: always the posting scorer
    double bm25_dbg = 0;      // BM25 posting scorer: experiment masks (1 no owner search, 2 no fieldnorm gather, 4 no output, 8 windows only)
    double bm25_sub_docs = 0; // BM25 posting scorer: documents per sub-range (0: from the batch's posting density)
    double bm25_emit = 1;     // BM25 over long corpora: sample / cut / emit (1) or per-block top-k lists only (0)
    double bm25_cand_cap = 0; // BM25 candidate slots per query (0 = 2048; small values force the fallback)
    double flat_few = 1;      // FLAT index, fewer than 16 queries over a large table: through the fp16 shadow pass (0: the canonical f32 scan; 2: whatever the table size)
    double bm25_rec = 1;      // BM25 posting scorer over score-ready records (bm25r_kernel; 0: bm25p_kernel over doc ids / tfs / gathered fieldnorms)
    double bm25_slots = 0;    // bm25r_kernel: hash slots of the shared-document filter (0 = 8192, 4 workgroups per CU; 16384: 3 per CU, fewer false alarms)
    double bm25_cutk = 1;     // BM25 cut by a register radix select per query (0: the list merge kernel)
    double bm25_lean = 1;     // BM25 record scorer: the four-term form (bm25l_kernel) when every query of the chunk has <= 4 terms (0: bm25r_kernel)
    double coarse_few = 128;     // the canonical coarse quantiser of a batch of at most this many queries in one self-merging launch (0: scan + merge launches)
    double host_pinned = 256;    // msvs_index_search over an IVFFLAT index, unfiltered batches of at most this many queries: queries and results through pinned memory (0: staged copies)
    double host_signal_batch = 1; // ... and their shadow list scan tells the host how many queries lost their certificate instead of launching the (normally empty) second chance and fallback rounds (0: always launched, stream synchronisation)
    double coarse_dense = 1;     // ... over a table of <= 2048 centroids: every key written, no top-k in the scanning blocks (0: nprobe keys per block)
    double plan_fused = 1;       // grouping the (query, list) pairs of a small batch (<= 8192 pairs) in one launch (0: memset + three launches)
    double flat_sample_few = 1;  // FLAT shadow pass, <= 32 queries: sample + cut in one launch (flat_sample_few_kernel; 0: coarse_h16_kernel + flat_cut_kernel)
    double flat_host_signal = 1; // FLAT, a few queries, host pointers: pinned in / out + a completion word (0: copies + stream synchronisation)
    double bm25_items_per_wave = 0; // BM25 emit pass: equal-postings items per resident wavefront (0 = 2)
    double bm25_select2 = 1;  // BM25 top-k of the candidates by selection (bm25_select2_kernel; 0: rank every candidate against every other)
    double bm25_skip = 1;     // BM25 posting sets carry a skip table of their frequent terms (read at msvs_postings_create)
    double bm25_tables_ride = 0; // BM25: the batch's tables are copied to the device by the bounds launch instead of a copy kernel of their own in front
                                 // of it (measured: 1.97 -> 1.96-1.99 us/query at 64, 0.66 -> 0.68 at 1024 -- the bounds launch reads its terms over the link: off)
    double bm25_bounds8 = 1;  // BM25 sub-range bounds by an 8-ary search (0: binary)
    double pinned_fetch = 1;     // small host -> device hand-overs out of pinned memory by a copy kernel instead of hipMemcpyAsync (device_ops.hpp: fetch_from_pinned)
    double pinned_fetch_max = 1048576; // ... up to this many bytes
    double route_self_rccl = 0; // routed sharded search over an RCCL communicator: a rank's OWN piece also travels through ncclSend / ncclRecv (to itself, grouped)
                                // instead of a device copy -- lets one rank on a 1-GPU box execute the point-to-point group path (tests)
};
const Options & options();
/// name without the MSVS_ prefix, any case; value nullptr / "" restores the default.  false = unknown name.
bool set_option(const char * name, const char * value);

}
