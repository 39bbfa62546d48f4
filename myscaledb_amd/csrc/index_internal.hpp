// index_internal.hpp -- what the translation units of libmsvs.so share about an index object: the msvs_index state itself and
// the stream-ordered search / filter / merge entry points behind the C-ABI (msvs_capi.hip defines them; shard.hip -- the
// multi-GPU search and the top-k merge -- calls them).
#pragma once
#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <vector>

#include "device_ops.hpp"

using msvs::DevBuf;

struct msvs_index
{
    int type = MSVS_INDEX_FLAT;
    int metric = MSVS_METRIC_L2;
    size_t dim = 0;
    uint32_t ld = 0;
    int device = 0;
    // build parameters
    size_t ncentroids = 1024;
    int kmeans_iters = 10;
    double split_big = 1.7, split_small = 0.75; // trainer: clusters above / below these multiples of the average size trade a centroid
    size_t train_empty_last = 0; // empty clusters the last k-means iteration re-seeded (diagnostics)
    size_t train_sample = 0;
    uint64_t seed = 1234;
    int shard_rank = 0, shard_world = 1;
    int (*is_cancelled)(void *) = nullptr; // msvs_index_set_cancel: the host's check_cancelled callback, polled by train / add / build
    void * cancel_ctx = nullptr;
    // coarse quantiser (IVFFLAT)
    size_t nlist = 0;
    DevBuf<float> centroids; // nlist x ld
    // staging (between add and build)
    struct Chunk
    {
        DevBuf<float> x; // n x ld (normalised for cosine)
        std::vector<int64_t> ids;
        std::vector<int32_t> assign;
        size_t n = 0;
    };
    std::vector<Chunk> chunks;
    size_t staged = 0;
    // final storage
    DevBuf<float> vecs;       // n x ld, list-major (IVF) / id order (FLAT)
    DevBuf<uint32_t> row_ids; // n
    int64_t last_id = -1;        // largest label so far while the labels arrive strictly ascending
    bool ids_may_repeat = false; // labels were given by the caller and are not known to be distinct (add does not forbid duplicates):
                                 // a filter's population count then does not bound the rows that pass it (build_view)
    DevBuf<int64_t> list_off; // nlist + 1
    std::vector<int64_t> h_list_off;
    size_t n = 0;
    size_t max_list_len = 0;
    uint64_t max_id = 0; // largest stored row id (size of the id space the filter bitmaps range over)
    // matrix-core candidate pass (mfma_scan_kernels.hpp): |x|^2 of every stored row and their maximum
    DevBuf<float> xnorm;
    float xnorm_max = 0.f;
    float xnorm_min = 0.f; // smallest |x|^2 over the rows (IVFFLAT: the cosine form of the pre-pruning needs the spread of the row norms)
    DevBuf<float> cnorm; // same for the centroids (the coarse quantiser goes through the same pass)
    DevBuf<float> list_radius; // nlist: an upper bound of max ||x - c_l|| over the rows of list l (probe pruning)
    float cnorm_max = 0.f;
    DevBuf<int64_t> list_mid; // nlist: end of the SAMPLE slice of list l = min(list_off[l] + 128, list_off[l+1])
    // fp16 shadow of the lists (h16_scan_kernels.hpp): the list scan of batched searches reads this instead of vecs
    int want_shadow = 1;        // build parameter `shadow=0|1`
    DevBuf<uint4> shadow;       // blocks of 32 rows in MFMA operand order
    DevBuf<uint32_t> hoff;      // nlist + 1: first block of list l
    DevBuf<int64_t> list_mid32; // nlist: end of block 0 of list l = min(list_off[l] + 32, list_off[l+1])
    uint32_t h_nks = 0, h_nch = 0;
    float h_scale = 0.f, h_inv_scale = 0.f; // stored value = fp16(x * h_scale)
    float h_rho = -1.f, c_rho = -1.f; // measured max |x' - x| / |x| over the stored rows / the centroid shadow (h16_rho_kernel; < 0: unknown)
    bool shadow_ready = false;
    // fp16 shadow of the CENTROID table (same scale, same block layout: ceil(nlist / 32) blocks) for the coarse quantiser of
    // batches, and the G-lists-of-one-block view the sample kernel walks it through
    DevBuf<uint4> c_shadow;
    DevBuf<uint32_t> c_hoff;     // [G + 1]: block g
    DevBuf<int64_t> c_list_off;  // [G + 1]: centroid 32 g (last: nlist)
    bool c_shadow_ready = false;
    bool ready = false;
    // VIWithMeta (src/VectorIndex/Cache/VICacheObject.h:40-117): state that rides on a cached index.  Swapped under `meta_mu`
    // (setDeleteBitmap is an atomic_store in the reference); a search keeps its own shared_ptr while it runs.
    struct Meta
    {
        DevBuf<uint64_t> delete_alive; // 1 = not deleted, over the index labels; empty = nothing deleted
        size_t delete_nbits = 0;
        DevBuf<uint64_t> row_ids_map;  // decoupled part: label -> row of the merged part (transferToNewRowIds)
        size_t row_ids_n = 0;
        DevBuf<uint64_t> inv_row_ids;  // merged-part row -> label of its source part ...
        DevBuf<uint8_t> inv_sources;   // ... and which source part (getRealBitmap keeps those of own_id)
        size_t inv_n = 0;
        uint32_t own_id = 0;
    };
    // the routed sharded search (shard.hip): radius and length of every list of the WHOLE index (all ranks' shards) and the
    // extremes of the row norms, gathered at the first routed search on a communicator
    struct Global
    {
        DevBuf<float> radius;      // [nlist]
        DevBuf<int64_t> list_off;  // [nlist + 1] prefix of the global list lengths
        float xmax = 0.f, xmin = 0.f;
        std::vector<uint64_t> instances; // [W]: the shard objects (msvs_index::instance of every rank) this was gathered from
    };
    mutable std::shared_ptr<Global> global;
    /// What the plan of the LAST shadow list scan of this index held after the pre-pruning (a pinned word the plan kernel writes, read
    /// by the next search without synchronisation) and the batch shape it belongs to: a performance hint only -- whether the second
    /// pruning stage can still pay (h16_list_scan) -- never a correctness input; racing searches may see each other's value.
    struct PlanFeedback
    {
        uint32_t * pairs = nullptr; // pinned host memory, 0xFFFFFFFF: nothing yet
        mutable uint32_t nq = 0, nprobe = 0;
        // pairs[1], pairs[2]: the sequence number (`seq`) of the last batched search whose coarse stage met a query without a band, and
        // whether there has been one at all (coarse_tail_kernel writes them; table_candidate_pass picks the form of the queue by them)
        mutable std::atomic<uint32_t> seq{0};
        // the inline form is retried after `window` searches without such a query; a retry that meets one again quadruples the window
        mutable std::atomic<uint32_t> window{0}, last_inline{0}, bumped{0};
        ~PlanFeedback()
        {
            if (pairs)
                (void)hipHostFree(pairs);
        }
    };
    PlanFeedback plan_fb;
    /// Identity of this object among the shard objects a rank has held (an index is immutable once built; a reloaded shard is a new
    /// object): the routed search compares it with the instance its Global was gathered from.
    const uint64_t instance = next_instance();
    static uint64_t next_instance();
    mutable std::mutex meta_mu;
    std::shared_ptr<Meta> meta;
    std::shared_ptr<Meta> get_meta() const
    {
        std::lock_guard<std::mutex> lk(meta_mu);
        return meta;
    }
};

inline uint64_t msvs_index::next_instance()
{
    static std::atomic<uint64_t> counter{0};
    static const uint64_t salt = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() * 0x9E3779B97F4A7C15ull;
    return (salt & ~0xFFFFFull) + counter.fetch_add(1) + 1;
}

namespace msvs
{
/// A compacted view of an index for one search (filter_kernels.hpp): the rows a selective filter lets through.
struct SearchView
{
    const int64_t * list_off; // [nlist + 1] view offsets of the lists (device)
    const uint32_t * rowmap;  // view row -> stored row
    const uint32_t * n_rows;  // device: rows in the view
    size_t n_upper;           // host: an upper bound of it
};


inline hipStream_t as_stream(void * s) { return reinterpret_cast<hipStream_t>(s); }
inline uint32_t padded_dim(size_t d) { return (uint32_t)round_up(d, 4); }
inline int scan_metric(int metric) { return metric == MSVS_METRIC_L2 ? M_L2 : M_IP; }
inline void check_k(size_t k)
{
    if (k > MSVS_MAX_K)
        fail(MSVS_ERR_UNSUPPORTED_K, "k = %zu exceeds the device top-k limit %d", k, MSVS_MAX_K);
}

/// Exhaustive top-k of device queries (nq x ld) against device rows (n x ld) -> device ids / distances (msvs_capi.hip).
/// `scr` must have been reserved by the caller for flat_scratch_bytes().
size_t flat_scratch_bytes(size_t n, size_t nq, uint32_t k, uint32_t ld);
void flat_search_device(Scratch & scr, int metric, const float * d_rows, const uint32_t * d_row_ids, size_t n, uint32_t ld,
                        const float * d_q, size_t nq, uint32_t k, const uint64_t * d_alive, size_t nbits, MergeParams out,
                        hipStream_t stream, const SearchView * view = nullptr);
/// VectorDataset::normalize on device rows (brute_force.hip): the reference's sequential sum, a wavefront per row.
void normalize_device_rows(float * d_x, size_t n, uint32_t d, uint32_t ld, hipStream_t stream);
/// Copy n rows of d floats (host or device) into a device buffer with row stride ld (zero padded).
void upload_rows(float * dst, const float * src, size_t n, uint32_t d, uint32_t ld, int mem, hipStream_t stream);
/// Row norms for the approximate pass and its error bound, then the fp16 shadows; called once the final storage is in place.
void index_finalize_norms(msvs_index & ix, hipStream_t stream);
uint32_t device_cu_count();
/// The per-index queue of concurrent host callers forgets an index that is being freed (msvs_capi.hip).
void combiner_forget(const msvs_index * ix);
/// The search proper: all pointers on the device, everything enqueued on `stream` (msvs_capi.hip).  given_probes (nullable):
/// [nq][nprobe] list ids computed elsewhere (another rank's share of the coarse quantiser): step 1 is skipped.  probes_only
/// (nullable): run ONLY step 1 and leave the probe lists there.
/// ProbeWords: the probe lists of a sharded search travel with the coarse pass's distance word of every probe (the probe pruning of the list
/// scan needs it and the rank that scans did not run the coarse pass of that query): [nq][nprobe] each, nullable.
struct ProbeWords
{
    const uint32_t * given = nullptr; // with given_probes
    uint32_t * out = nullptr;         // with probes_only (0xFFFFFFFF where the coarse pass left no word)
    // with probes_only, the routed sharded search (shard.hip): the pre-pruning by the list radius over the lists of the WHOLE index
    // (radius and list lengths of every rank's lists, gathered once) -- pruned_out [nq][nprobe] gets the probes that survive (-1:
    // dropped), all of them when the pre-pruning cannot run for this index / batch
    const float * g_radius = nullptr;
    const int64_t * g_list_off = nullptr;
    float g_xmax = 0.f, g_xmin = 0.f; // max / min |x|^2 over the rows of every rank (the bounds' error terms)
    int32_t * pruned_out = nullptr;
    bool given_pruned = false; // with given_probes: they are what a pre-pruning on the rank they came from left (the routed search's back phase)
};
/// search_entry.hip: the canonical coarse quantiser of a small batch in one launch (-> false: not for this shape).
bool coarse_few_launch(const msvs_index & ix, const float * dq, size_t nq, size_t nprobe, int32_t * d_probes, float * d_probe_dis,
                       hipStream_t stream);

/// A host-pointer search of a few queries over a FLAT shadow (search_entry.hip: flat_few_search_host) arms this before it calls
/// index_search_device: the table pass then ends with a one-thread launch that copies the certificate-failure count to `nfail` and
/// sets `flag` to `seq` (both in pinned memory), waits for the word itself, and runs the canonical fallback only when somebody
/// failed -- the two (normally empty) fallback launches, both result copies and the stream synchronisation leave the call.
struct HostSignal
{
    uint32_t * flag = nullptr;
    uint32_t * nfail = nullptr;
    uint32_t seq = 0;
    bool armed = false, used = false;
    /// Before a call arms the signal: the completion word reads "not yet" (seq is never 0) and the failure count "unknown" -- the slot
    /// lives inside a reused pinned block at an offset that depends on (nq, k, dim), so it may hold ids of an earlier call.  No
    /// kernel of an earlier call can still write there: host-pointer calls return after their stream has run dry.
    void reset() const
    {
        __atomic_store_n(flag, 0u, __ATOMIC_RELAXED);
        __atomic_store_n(nfail, 0xFFFFFFFFu, __ATOMIC_RELEASE);
    }
};
HostSignal & host_signal();

void index_search_device(const msvs_index & ix, const float * d_queries /* nq x dim, dense */, size_t nq, uint32_t k, size_t nprobe,
                         const uint64_t * d_alive, size_t nbits, int64_t * d_ids, float * d_dis, hipStream_t stream,
                         const int32_t * given_probes = nullptr, int32_t * probes_only = nullptr, const SearchView * view = nullptr,
                         ProbeWords words = ProbeWords{});
/// The filter a search really runs with: (per-search filter, converted to label space for a decoupled part) AND the resident
/// delete bitmap.  Returns the device pointer (nullptr = no filter) and its valid bits; scratch from aux_for(stream).
const uint64_t * effective_filter(const msvs_index & ix, const msvs_index::Meta * meta, const uint64_t * d_alive, size_t nbits,
                                  size_t * eff_nbits, hipStream_t stream);
void apply_row_ids_map(const msvs_index::Meta * meta, int64_t * d_ids, size_t n, hipStream_t stream);
/// A search under a filter with a known population count (filters.hip): bit test inside the scan, or a compacted view.
void index_search_filtered(const msvs_index & ix, const float * d_queries, size_t nq, uint32_t k, size_t nprobe, const uint64_t * eff,
                           size_t eff_bits, uint64_t alive_count, int64_t * d_ids, float * d_dis, hipStream_t stream);
/// The host-pointer search behind msvs_index_search / msvs_index_search_filter (msvs_capi.hip): staging, parameter string,
/// exact rounds for large k.
int index_search_host_call(const msvs_index_t * ix, const float * queries, size_t nq, int k, const char * params,
                           const uint64_t * alive_bits, size_t nbits, int64_t * ids, float * dis);
/// Canonical merge of nparts partial top-k lists per query (shard.hip); strides in elements between the parts' [nq][k] arrays.
void merge_topk_device(const int64_t * d_ids, size_t ids_stride, const float * d_dis, size_t dis_stride, size_t nparts, size_t nq,
                       size_t k, int metric, int64_t * d_out_ids, float * d_out_dis, hipStream_t stream);
}

/// Clear the filter bits of the (non-negative) ids just returned, so the next round of a large-k search skips them.
static __global__ void clear_bits_kernel(uint64_t * bits, const int64_t * ids, uint32_t n)
{
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
    {
        const int64_t id = ids[i];
        if (id >= 0)
            atomicAnd(reinterpret_cast<unsigned long long *>(bits + (id >> 6)), ~(1ull << (id & 63)));
    }
}
