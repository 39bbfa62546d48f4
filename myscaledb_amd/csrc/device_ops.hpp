// device_ops.hpp -- host-side launchers of the scan / merge kernels (all device pointers, stream-ordered).
#pragma once

#include "common.hpp"
#include "scan_kernels.hpp"
#include "mfma_scan_kernels.hpp"

namespace msvs
{

/// Grow-only scratch arena, one per (host thread, stream): search calls are stream-ordered, so consecutive
/// calls on one stream may reuse it without synchronisation; different host threads never share one.
struct Scratch
{
    DevBuf<unsigned char> buf;
    size_t used = 0;
    size_t recent_max = 0; // largest reservation of the last `ops` operations (shrink policy, see reserve)
    uint32_t ops = 0;
    void reset() { used = 0; }
    /// Reserve must be called once per operation with the total need BEFORE any take() (it may reallocate).
    void reserve(size_t bytes, hipStream_t stream);
    template <typename T>
    T * take(size_t count)
    {
        size_t off = round_up(used, 256);
        size_t need = off + count * sizeof(T);
        if (need > buf.n)
            fail(MSVS_ERR_DEVICE, "internal: scratch overflow (%zu > %zu)", need, buf.n);
        used = need;
        return reinterpret_cast<T *>(buf.p + off);
    }
};

/// The stream of the host-pointer entry points: one per (host thread, device), non-blocking, so that concurrent client
/// threads (the reference runs searches on up to 2 x cores of them, ScanThreadLimiter.h) overlap on the GPU instead of
/// queueing on the null stream.  Never destroyed: thread exit may come after the runtime is gone.
hipStream_t thread_stream();

Scratch & scratch_for(hipStream_t stream);
/// A second arena of the same kind for the host-pointer entry points' staging copies (queries in, results out): the
/// device-level search underneath owns scratch_for(); a hipMalloc / hipFree pair per call costs ~0.1 ms.
Scratch & staging_for(hipStream_t stream);
/// A third one for the effective filter of a search (per-search filter AND resident delete bitmap, id-space conversion).
Scratch & aux_for(hipStream_t stream);
/// ... and one for the exchange buffers of a sharded search (probe lists, packed partial top-k of every rank).
Scratch & shard_for(hipStream_t stream);
Scratch & route_for(hipStream_t stream); // the routed sharded search's exchange buffers (shard.hip)
/// ... and one for the compacted view of a filtered search and small per-filter counters.
Scratch & view_for(hipStream_t stream);
/// Host -> device hand-over of a SMALL block that already sits in pinned host memory (query batches of the host-pointer calls, a BM25
/// batch's tables): a copy KERNEL that streams it over the link (nontemporal 16-byte loads) instead of hipMemcpyAsync.  The runtime's
/// copy is a blit surrounded by barrier packets: ~10 us of idle device before and after it in the kernel trace
/// (profiles/r05_bm25_64_kernel_trace_timeline.txt: 12.2 + 10.4 us around a 3 us copy); a kernel follows the previous kernel with no gap.
/// bytes is rounded up to 16: both buffers must be 16-byte aligned and hold the rounded size.  Larger blocks (option pinned_fetch_max,
/// default 1 MB) and option pinned_fetch = 0 take hipMemcpyAsync.
void fetch_from_pinned(void * d_dst, const void * h_pinned_src, size_t bytes, hipStream_t stream);

/// Free every arena of the calling host thread (after a device synchronisation); returns the bytes released.
size_t release_thread_arenas();

/// Optional HIP-event timing of kernel launches (msvs_profile_* in the C-ABI); a no-op unless enabled.
struct ProfileScope
{
    ProfileScope(const char * name, hipStream_t stream);
    ~ProfileScope();
    const char * name;
    hipStream_t stream;
    hipEvent_t start = nullptr;
};
void profile_enable(bool on);
void profile_get(const char * name, uint64_t * calls, double * total_ms);
void profile_reset();

inline int r_for_k(uint32_t k) { return k <= 64 ? 1 : (k <= 128 ? 2 : 4); }

struct FlatPlan
{
    uint32_t T, n_qtiles, rows_per_block, n_blocks;
};

/// ld4 / k: the tile shrinks until its LDS image (scan_lds_bytes) fits SCAN_LDS_BUDGET.
FlatPlan plan_flat(size_t n_rows, size_t nq, uint32_t ld4, uint32_t k);

/// bytes of partial keys a flat scan writes
inline size_t flat_partial_keys(const FlatPlan & p, size_t nq, uint32_t k) { return nq * (size_t)p.n_blocks * k; }

/// Exhaustive scan of `n_rows` rows for nq queries -> partial[nq][n_blocks][k] keys.
void launch_flat_scan(int metric, const FlatPlan & plan, ScanParams a, hipStream_t stream);

/// partial[nq][n_lists][k] -> final results.
void launch_merge(int metric, MergeParams a, uint32_t nq, hipStream_t stream);

/// IVF list scan, one query per block, grid (seg_max, nprobe, nq).
void launch_ivf_scan(int metric, ScanParams a, hipStream_t stream);

/// Group the (query, probed list) pairs by list: histogram, scans, scatter (p.cnt / p.fill must be zeroed).
void launch_ivf_plan(const IvfPlanParams & p, hipStream_t stream);
/// ... in ONE launch (small batches: nothing to zero beforehand, p.zero cleared on the way)?
bool ivf_plan_fused(const IvfPlanParams & p);
/// Only the two exclusive scans again (pair_off, work_off) for another row range / work-item size of the same pairs.
void launch_ivf_plan_rescan(const IvfPlanParams & p, hipStream_t stream);

/// List-batched IVF scan over the plan's work items; T in {2, 4, 8}; fixed grid of `grid` blocks.
void launch_ivf_batched_scan(int metric, uint32_t T, uint32_t grid, ScanParams a, hipStream_t stream);

/// Per-query top-k over the valid segments of its probed lists.
void launch_ivf_merge(int metric, IvfMergeParams a, uint32_t nq, hipStream_t stream);

// ---- matrix-core candidate pass (mfma_scan_kernels.hpp)

/// out[r] = |X[r]|^2 for n rows of ld4 float4; max_bits: nullable running maximum of the float bit patterns.
void launch_row_sqnorm(const float * X, float * out, size_t n, uint32_t ld4, uint32_t * max_bits, hipStream_t stream);

/// Queries (nq rows of ld4 float4) -> split-bf16 step layout, nq * ceil(ld4 / 8) * 128 bytes (split_queries_kernel).
void launch_split_queries(const float * Q, uint32_t nq, uint32_t ld4, void * out, hipStream_t stream);

/// Approximate (split-bf16 MFMA) scan over the plan's work items; nqg in {1, 2}: 128- or 256-query tiles (plan built
/// with T = BG_TQ * nqg); appends candidates to a.partial through a.qcnt / a.qthr (see mfma_scan_kernels.hpp).
void launch_ivf_mfma_scan(int metric, uint32_t nqg, uint32_t grid, ScanParams a, hipStream_t stream,
                          const char * profile_name = "ivf_scan", bool main_phase = true);

/// One-list plan over rows [row_begin, row_end) of a plain row table (see single_list_plan_kernel).
void launch_single_list_plan(uint32_t nq, uint32_t row_begin, uint32_t row_end, uint32_t rows_per_block, uint32_t tq,
                             uint32_t * pairs, int32_t * probes0, int64_t * list_off, uint32_t * pair_off,
                             uint32_t * work_off, hipStream_t stream);

/// qthr[q] = min(qthr[q], distance word of cand[q][m-1]) (see sample_cut_kernel).
void launch_sample_cut(const uint64_t * cand, uint32_t kc, uint32_t m, uint32_t nq, uint32_t * qthr, hipStream_t stream);

/// Big-tile pass only: the kc best of every query's appended candidates + the bound of what the slices cut.
void launch_cand_select(const uint64_t * buf, const uint32_t * qcnt, const uint32_t * qthr, uint32_t cap, uint32_t nq,
                        uint32_t kc, uint64_t * out, uint64_t * bound, hipStream_t stream);

/// Canonical re-rank of the candidates + certificate; failing queries are appended to a.failq.
void launch_ivf_rerank(int metric, RerankParams a, uint32_t nq, hipStream_t stream);
/// Second chance of the queries on a.failq-style list b.failq_in: re-rank of their whole candidate buffers (mfma_scan_kernels.hpp).
void launch_ivf_rerank_all(int metric, const RerankParams & a, const RerankAllParams & b, uint32_t nq, hipStream_t stream);

/// Canonical scan / merge of the queries listed in (a.qmap, *a.qcount) with `z` block slots per (segment, probe).
void launch_ivf_scan_subset(int metric, ScanParams a, uint32_t z, hipStream_t stream);
void launch_ivf_merge_subset(int metric, IvfMergeParams a, uint32_t blocks, hipStream_t stream);

}
