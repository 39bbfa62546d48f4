// shard.hip -- the top-k merge of partial lists and the multi-GPU search of libmsvs.so (include/msvs.h: msvs_merge_topk*,
// msvs_comm_*, msvs_shard_search_device).
#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>

#include "index_internal.hpp"

using namespace msvs;

// =========================================================================================== merge

namespace msvs
{
void merge_topk_device(const int64_t * d_ids, size_t ids_stride, const float * d_dis, size_t dis_stride,
                              size_t nparts, size_t nq, size_t k, int metric, int64_t * d_out_ids, float * d_out_dis,
                              hipStream_t stream)
{
    if (metric != MSVS_METRIC_L2 && metric != MSVS_METRIC_IP)
        fail(MSVS_ERR_NOT_IMPLEMENTED, "merge supports L2 / IP ordering (cosine distances are ascending: use L2)");
    if (nq == 0 || k == 0)
        return;
    check_k(k);
    Scratch & scr = scratch_for(stream);
    const size_t total = nparts * nq * k;
    scr.reserve(total * 8 + 8192, stream);
    uint64_t * keys_q = scr.take<uint64_t>(total); // [nq][nparts][k]
    if (metric == MSVS_METRIC_IP)
        hipLaunchKernelGGL((pack_keys_kernel<M_IP>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, stream, d_ids,
                           ids_stride, d_dis, dis_stride, keys_q, (uint32_t)nparts, (uint32_t)nq, (uint32_t)k);
    else
        hipLaunchKernelGGL((pack_keys_kernel<M_L2>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, stream, d_ids,
                           ids_stride, d_dis, dis_stride, keys_q, (uint32_t)nparts, (uint32_t)nq, (uint32_t)k);
    MSVS_HIP(hipGetLastError());
    MergeParams m{};
    m.partial = keys_q;
    m.n_lists = (uint32_t)nparts;
    m.k = (uint32_t)k;
    m.out_ids = d_out_ids;
    m.out_dis = d_out_dis;
    launch_merge(scan_metric(metric), m, (uint32_t)nq, stream);
}
}

extern "C" int msvs_merge_topk_device(const int64_t * d_ids, const float * d_dis, size_t nparts, size_t nq, size_t k,
                                      int metric, int64_t * d_out_ids, float * d_out_dis, void * hip_stream)
{
    return guarded([&] {
        if (nparts && nq && k && (!d_ids || !d_dis || !d_out_ids || !d_out_dis))
            fail(MSVS_ERR_INVALID_ARGUMENT, "null buffer");
        merge_topk_device(d_ids, nq * k, d_dis, nq * k, nparts, nq, k, metric, d_out_ids, d_out_dis,
                          as_stream(hip_stream));
    });
}

extern "C" int msvs_merge_topk_device_strided(const int64_t * d_ids, size_t ids_part_stride, const float * d_dis,
                                              size_t dis_part_stride, size_t nparts, size_t nq, size_t k, int metric,
                                              int64_t * d_out_ids, float * d_out_dis, void * hip_stream)
{
    return guarded([&] {
        if (nparts && nq && k && (!d_ids || !d_dis || !d_out_ids || !d_out_dis))
            fail(MSVS_ERR_INVALID_ARGUMENT, "null buffer");
        if (ids_part_stride < nq * k || dis_part_stride < nq * k)
            fail(MSVS_ERR_INVALID_ARGUMENT, "part stride smaller than nq * k");
        merge_topk_device(d_ids, ids_part_stride, d_dis, dis_part_stride, nparts, nq, k, metric, d_out_ids, d_out_dis,
                          as_stream(hip_stream));
    });
}

extern "C" int msvs_merge_topk(const int64_t * ids, const float * dis, size_t nparts, size_t nq, size_t k, int metric,
                               int64_t * out_ids, float * out_dis)
{
    return guarded([&] {
        if (nparts == 0 || nq == 0 || k == 0)
            return;
        if (!ids || !dis || !out_ids || !out_dis)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null buffer");
        const size_t total = nparts * nq * k;
        DevBuf<int64_t> d_ids(total), d_oi(nq * k);
        DevBuf<float> d_dis(total), d_od(nq * k);
        MSVS_HIP(hipMemcpy(d_ids.p, ids, total * 8, hipMemcpyHostToDevice));
        MSVS_HIP(hipMemcpy(d_dis.p, dis, total * 4, hipMemcpyHostToDevice));
        merge_topk_device(d_ids.p, nq * k, d_dis.p, nq * k, nparts, nq, k, metric, d_oi.p, d_od.p, nullptr);
        MSVS_HIP(hipMemcpy(out_ids, d_oi.p, nq * k * 8, hipMemcpyDeviceToHost));
        MSVS_HIP(hipMemcpy(out_dis, d_od.p, nq * k * 4, hipMemcpyDeviceToHost));
    });
}

// =========================================================================================== multi-GPU
//
// One process per GPU, lists sharded list_id % world (index params shard_rank / shard_world; FLAT: row ranges), centroids
// replicated.  A sharded search is ONE call on ONE stream that owns its communicator -- RCCL over xGMI, straight from
// rccl.h, no Python in the data path:
//   1. the coarse quantiser is sharded BY QUERY: rank r ranks the centroids for queries [r c, (r + 1) c), c = ceil(nq / W),
//      and ONE all-gather of c * nprobe int32 per rank gives every rank the probe lists of the whole batch -- nothing of
//      the per-step work is replicated (round 1 ran coarse quantiser, plan and selection on every rank: Amdahl ~3x at 8);
//   2. every rank scans its LOCAL probed lists for the whole batch (pairs that point at lists it does not own are dropped
//      by the plan), exact local top-k straight into its slot of the packed exchange buffer {ids i64 | dis f32}[nq][k];
//   3. ONE all-gather of nq * k * 12 B per rank, then the canonical W-way merge in place (identical on every rank) --
//      the device-side getTotalTopSearchResultImpl (MergeTreeBaseSearchManager.cpp:207-299).
// RCCL is resolved with dlopen at first use (the host process -- ClickHouse, or PyTorch in the bench -- may already carry
// its own librccl: the loader then hands back that one instead of a second copy).
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace
{
struct RcclApi
{
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char * (*GetErrorString)(ncclResult_t) = nullptr;
};

const RcclApi & rccl()
{
    static RcclApi api;
    static std::once_flag once;
    static std::string err;
    std::call_once(once, [] {
        void * h = nullptr;
        for (const char * name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)))
                break;
        if (!h)
        {
            err = std::string("cannot load librccl: ") + dlerror();
            return;
        }
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(h, "ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather)
            err = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
    });
    if (!err.empty())
        msvs::fail(MSVS_ERR_DEVICE, "%s", err.c_str());
    return api;
}

void nccl_check(ncclResult_t r, const char * what)
{
    if (r != ncclSuccess)
        msvs::fail(MSVS_ERR_DEVICE, "%s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(r) : "rccl error");
}
}

struct msvs_comm
{
    int nranks = 1, rank = 0;
    ncclComm_t nccl = nullptr;
    msvs_allgather_fn custom = nullptr;
    void * ctx = nullptr;
    /// every rank contributes `bytes` at d_buf + rank * bytes; afterwards d_buf holds all nranks slots (in place)
    void all_gather(unsigned char * d_buf, size_t bytes, hipStream_t stream) const
    {
        if ((nranks == 1 && !nccl) || bytes == 0)
            return;
        if (custom)
        {
            if (custom(ctx, d_buf + (size_t)rank * bytes, d_buf, bytes, stream) != 0)
                msvs::fail(MSVS_ERR_DEVICE, "the caller-supplied all-gather failed");
            return;
        }
        nccl_check(rccl().AllGather(d_buf + (size_t)rank * bytes, d_buf, bytes, ncclInt8, nccl, stream), "ncclAllGather");
    }
};

extern "C" int msvs_comm_unique_id(void * id_out)
{
    return guarded([&] {
        static_assert(MSVS_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
        if (!id_out)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null id buffer");
        ncclUniqueId id;
        nccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
        memcpy(id_out, &id, sizeof(id));
    });
}

extern "C" int msvs_comm_init(const void * id, int nranks, int rank, msvs_comm_t ** out)
{
    return guarded([&] {
        if (!out || nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !id))
            fail(MSVS_ERR_INVALID_ARGUMENT, "bad communicator arguments");
        std::unique_ptr<msvs_comm> c(new msvs_comm);
        c->nranks = nranks;
        c->rank = rank;
        if (nranks > 1 || id) // a single rank WITH an id still gets a real RCCL communicator (self-test of the transport)
        {
            ncclUniqueId uid;
            memcpy(&uid, id, sizeof(uid));
            nccl_check(rccl().CommInitRank(&c->nccl, nranks, uid, rank), "ncclCommInitRank"); // on the current device
        }
        *out = c.release();
    });
}

extern "C" int msvs_comm_init_custom(int nranks, int rank, msvs_allgather_fn all_gather, void * ctx, msvs_comm_t ** out)
{
    return guarded([&] {
        if (!out || nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !all_gather))
            fail(MSVS_ERR_INVALID_ARGUMENT, "bad communicator arguments");
        std::unique_ptr<msvs_comm> c(new msvs_comm);
        c->nranks = nranks;
        c->rank = rank;
        c->custom = all_gather;
        c->ctx = ctx;
        *out = c.release();
    });
}

extern "C" void msvs_comm_free(msvs_comm_t * c)
{
    if (!c)
        return;
    if (c->nccl)
        (void)rccl().CommDestroy(c->nccl);
    delete c;
}

extern "C" int msvs_comm_all_reduce_u64(const msvs_comm_t * comm, uint64_t * values, size_t n, void * hip_stream)
{
    return guarded([&] {
        if (!comm || (n && !values))
            fail(MSVS_ERR_INVALID_ARGUMENT, "null communicator / values");
        if (n == 0 || (comm->nranks == 1 && !comm->nccl))
            return;
        // a few dozen counters (BM25: documents, tokens per column, document frequency per query term): one all-gather of
        // every rank's vector on the communicator the searches use, summed locally -- the same transport whatever it is
        // (RCCL or the caller's), no second collective type to support
        hipStream_t stream = as_stream(hip_stream);
        const size_t W = (size_t)comm->nranks, bytes = n * 8;
        DevBuf<unsigned char> buf(W * bytes);
        MSVS_HIP(hipMemcpyAsync(buf.p + (size_t)comm->rank * bytes, values, bytes, hipMemcpyHostToDevice, stream));
        comm->all_gather(buf.p, bytes, stream);
        std::vector<uint64_t> all(W * n);
        MSVS_HIP(hipMemcpyAsync(all.data(), buf.p, W * bytes, hipMemcpyDeviceToHost, stream));
        MSVS_HIP(hipStreamSynchronize(stream));
        for (size_t i = 0; i < n; i++)
        {
            uint64_t sum = 0;
            for (size_t r = 0; r < W; r++)
                sum += all[r * n + i];
            values[i] = sum;
        }
    });
}

extern "C" int msvs_comm_rank(const msvs_comm_t * c) { return c ? c->rank : -1; }
extern "C" int msvs_comm_size(const msvs_comm_t * c) { return c ? c->nranks : 0; }

/// The pipelined form's state, owned by the communicator: a compute stream, an exchange stream, and per parity of the call
/// count the events that order the stages and the arena that holds the exchange buffers.
struct ShardPipe
{
    hipStream_t compute = nullptr, xchg = nullptr;
    hipEvent_t in_ev[2] = {nullptr, nullptr}, ab_ev[2] = {nullptr, nullptr}, x1_ev[2] = {nullptr, nullptr}, b_ev[2] = {nullptr, nullptr},
               done_ev[2] = {nullptr, nullptr};
    bool used[2] = {false, false};
    uint64_t calls = 0;
    std::mutex mu;
};

static ShardPipe & pipe_of(const msvs_comm_t * comm)
{
    static std::mutex mu;
    static std::map<const msvs_comm_t *, std::unique_ptr<ShardPipe>> pipes; // (never torn down: a handful per process)
    std::lock_guard<std::mutex> lk(mu);
    auto & p = pipes[comm];
    if (!p)
    {
        p.reset(new ShardPipe);
        MSVS_HIP(hipStreamCreateWithFlags(&p->compute, hipStreamNonBlocking));
        MSVS_HIP(hipStreamCreateWithFlags(&p->xchg, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++)
            for (hipEvent_t * e : {&p->in_ev[i], &p->ab_ev[i], &p->x1_ev[i], &p->b_ev[i], &p->done_ev[i]})
                MSVS_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    return *p;
}

/// cs: the stream of the coarse pass, the list scan and the small copies; xs: the stream of the two all-gathers and the merge
/// (== cs for the one-stream form); ev_a / ev_x1 / ev_b order the stages across the two when they differ; sh: where the exchange
/// buffers live (one arena per call in flight).
static void shard_search_stages(const msvs_index_t * ix, const msvs_comm_t * comm, const float * d_queries, size_t nq, int k, int nprobe,
                                const uint64_t * d_alive_bits, size_t nbits, int64_t * d_ids, float * d_dis, hipStream_t cs,
                                hipStream_t xs, hipEvent_t ev_a, hipEvent_t ev_x1, hipEvent_t ev_b, Scratch & sh)
{
    auto hand_over = [&](hipStream_t from, hipStream_t to, hipEvent_t ev) {
        if (from == to)
            return;
        MSVS_HIP(hipEventRecord(ev, from));
        MSVS_HIP(hipStreamWaitEvent(to, ev, 0));
    };
    const auto meta = ix->get_meta();
    size_t eff_bits = nbits;
    const uint64_t * eff = effective_filter(*ix, meta.get(), d_alive_bits, nbits, &eff_bits, cs);
    const size_t W = (size_t)comm->nranks, r = (size_t)comm->rank;
    const size_t part = round_up(nq * (size_t)k * 12, 16); // one rank's {ids | dis}
    const bool ivf = ix->type == MSVS_INDEX_IVFFLAT;
    const size_t np = ivf ? std::min<size_t>(std::max(nprobe, 1), ix->nlist) : 0;
    const size_t chunk = ceil_div(nq, W);
    sh.reserve(W * part + 4 * W * chunk * np * 4 + 8192, cs);
    unsigned char * packed = sh.take<unsigned char>(W * part);
    int64_t * my_ids = reinterpret_cast<int64_t *>(packed + r * part);
    float * my_dis = reinterpret_cast<float *>(packed + r * part + nq * (size_t)k * 8);
    if (ivf)
    {
        // one record per rank: its chunk's probe lists [chunk][np] followed by the coarse pass's distance word of every probe
        // [chunk][np] -- the probe pruning of the list scan runs on every rank, for every query, although the rank ran the
        // coarse pass of a W-th of them only (nq np 4 B more in the same all-gather)
        const size_t rec = 2 * chunk * np; // 32-bit words per rank
        int32_t * exch = sh.take<int32_t>(W * rec);
        int32_t * mine = exch + r * rec;
        uint32_t * mine_words = reinterpret_cast<uint32_t *>(mine + chunk * np);
        const size_t q0 = std::min(nq, r * chunk), m = std::min(chunk, nq - q0);
        MSVS_HIP(hipMemsetAsync(mine, 0xFF, rec * 4, cs)); // queries past nq: no probes, no words
        if (m)
            index_search_device(*ix, d_queries + q0 * ix->dim, m, 1, np, nullptr, 0, nullptr, nullptr, cs, nullptr, mine, nullptr,
                                ProbeWords{nullptr, mine_words});
        hand_over(cs, xs, ev_a);
        {
            ProfileScope prof("shard_exchange", xs);
            comm->all_gather(reinterpret_cast<unsigned char *>(exch), rec * 4, xs);
        }
        hand_over(xs, cs, ev_x1);
        // [rank][probes | words] -> probes [nq][np], words [nq][np]
        int32_t * probes = sh.take<int32_t>(W * chunk * np);
        uint32_t * pwords = sh.take<uint32_t>(W * chunk * np);
        MSVS_HIP(hipMemcpy2DAsync(probes, chunk * np * 4, exch, rec * 4, chunk * np * 4, W, hipMemcpyDeviceToDevice, cs));
        MSVS_HIP(hipMemcpy2DAsync(pwords, chunk * np * 4, exch + chunk * np, rec * 4, chunk * np * 4, W, hipMemcpyDeviceToDevice, cs));
        index_search_device(*ix, d_queries, nq, (uint32_t)k, np, eff, eff_bits, my_ids, my_dis, cs, probes, nullptr, nullptr,
                            ProbeWords{pwords, nullptr});
    }
    else
        index_search_device(*ix, d_queries, nq, (uint32_t)k, 0, eff, eff_bits, my_ids, my_dis, cs);
    hand_over(cs, xs, ev_b);
    {
        ProfileScope prof("shard_exchange", xs);
        comm->all_gather(packed, part, xs);
    }
    // cosine distances leave the search as 1 - ip: ascending like L2
    const int order = ix->metric == MSVS_METRIC_IP ? MSVS_METRIC_IP : MSVS_METRIC_L2;
    merge_topk_device(reinterpret_cast<const int64_t *>(packed), part / 8, reinterpret_cast<const float *>(packed + nq * (size_t)k * 8),
                      part / 4, W, nq, (size_t)k, order, d_ids, d_dis, xs);
    apply_row_ids_map(meta.get(), d_ids, nq * (size_t)k, xs);
}

static void shard_check(const msvs_index_t * ix, const msvs_comm_t * comm, const float * d_queries, size_t nq, int k, const int64_t * d_ids,
                        const float * d_dis)
{
    if (!ix || !comm || (nq && (!d_queries || !d_ids || !d_dis)) || k < 0)
        fail(MSVS_ERR_INVALID_ARGUMENT, "null index / communicator / buffer or negative k");
    if (comm->nranks != ix->shard_world || comm->rank != ix->shard_rank)
        fail(MSVS_ERR_INVALID_ARGUMENT, "the index is shard %d of %d but the communicator is rank %d of %d", ix->shard_rank,
             ix->shard_world, comm->rank, comm->nranks);
}

extern "C" int msvs_shard_search_device(const msvs_index_t * ix, const msvs_comm_t * comm, const float * d_queries, size_t nq,
                                        int k, int nprobe, const uint64_t * d_alive_bits, size_t nbits, int64_t * d_ids,
                                        float * d_dis, void * hip_stream)
{
    return guarded([&] {
        shard_check(ix, comm, d_queries, nq, k, d_ids, d_dis);
        hipStream_t stream = as_stream(hip_stream);
        if (nq == 0 || k == 0)
            return;
        if (comm->nranks == 1 && !comm->nccl)
        {
            const auto meta = ix->get_meta();
            size_t eff_bits = nbits;
            const uint64_t * eff = effective_filter(*ix, meta.get(), d_alive_bits, nbits, &eff_bits, stream);
            index_search_device(*ix, d_queries, nq, (uint32_t)k, (size_t)std::max(nprobe, 0), eff, eff_bits, d_ids, d_dis, stream);
            apply_row_ids_map(meta.get(), d_ids, nq * (size_t)k, stream);
            return;
        }
        shard_search_stages(ix, comm, d_queries, nq, k, nprobe, d_alive_bits, nbits, d_ids, d_dis, stream, stream, nullptr, nullptr, nullptr,
                            shard_for(stream));
    });
}

/// Two batches in flight (msvs.h): batch i's top-k all-gather and merge run on the communicator's exchange stream while the
/// coarse pass and the list scan of batch i + 1 run on its compute stream.  Every collective of the communicator is issued on
/// the ONE exchange stream, in call order: the order RCCL needs is the program order of the calls, the same on every rank.
extern "C" int msvs_shard_search_device_async(const msvs_index_t * ix, const msvs_comm_t * comm, const float * d_queries, size_t nq,
                                              int k, int nprobe, const uint64_t * d_alive_bits, size_t nbits, int64_t * d_ids,
                                              float * d_dis, void * hip_stream, void ** done_event)
{
    return guarded([&] {
        shard_check(ix, comm, d_queries, nq, k, d_ids, d_dis);
        if (!done_event)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null done_event");
        ShardPipe & pp = pipe_of(comm);
        std::lock_guard<std::mutex> lk(pp.mu);
        const int p = (int)(pp.calls++ & 1);
        // inputs: whatever the caller's stream has enqueued so far
        MSVS_HIP(hipEventRecord(pp.in_ev[p], as_stream(hip_stream)));
        MSVS_HIP(hipStreamWaitEvent(pp.compute, pp.in_ev[p], 0));
        // this parity's exchange buffers were last read by the merge of the call before the previous one
        if (pp.used[p])
            MSVS_HIP(hipStreamWaitEvent(pp.compute, pp.done_ev[p], 0));
        pp.used[p] = true;
        if (nq && k)
            shard_search_stages(ix, comm, d_queries, nq, k, nprobe, d_alive_bits, nbits, d_ids, d_dis, pp.compute, pp.xchg, pp.ab_ev[p],
                                pp.x1_ev[p], pp.b_ev[p], shard_for(p ? pp.xchg : pp.compute));
        else
        {
            MSVS_HIP(hipEventRecord(pp.b_ev[p], pp.compute));
            MSVS_HIP(hipStreamWaitEvent(pp.xchg, pp.b_ev[p], 0));
        }
        MSVS_HIP(hipEventRecord(pp.done_ev[p], pp.xchg));
        *done_event = pp.done_ev[p];
    });
}

extern "C" int msvs_shard_search_drain(const msvs_comm_t * comm, void * hip_stream)
{
    return guarded([&] {
        if (!comm)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null communicator");
        ShardPipe & pp = pipe_of(comm);
        std::lock_guard<std::mutex> lk(pp.mu);
        for (int p = 0; p < 2; p++)
            if (pp.used[p])
                MSVS_HIP(hipStreamWaitEvent(as_stream(hip_stream), pp.done_ev[p], 0));
    });
}
