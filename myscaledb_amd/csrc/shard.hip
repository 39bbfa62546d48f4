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
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr; // (the routed search: optional)
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
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
        api.Send = reinterpret_cast<decltype(api.Send)>(dlsym(h, "ncclSend"));
        api.Recv = reinterpret_cast<decltype(api.Recv)>(dlsym(h, "ncclRecv"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(h, "ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
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
    /// The routed search's exchange: rank s sends mat[s][t] bytes to rank t (mat: the whole nranks x nranks matrix, known on every
    /// host).  d_send: this rank's pieces back to back in destination order; d_recv: the pieces it receives, back to back in source
    /// order.  RCCL: one group of point-to-point sends / receives of the exact sizes over xGMI.  A caller-supplied transport has an
    /// all-gather only: every rank's whole send buffer (padded to the largest) is gathered and the pieces are picked out of it --
    /// nranks times the traffic, the same result (tests: two ranks on one GPU over gloo).  `tmp`: nranks * max row sum bytes (custom only).
    void exchange(const unsigned char * d_send, unsigned char * d_recv, const std::vector<size_t> & mat, unsigned char * tmp, hipStream_t stream) const
    {
        const size_t W = (size_t)nranks, r = (size_t)rank;
        std::vector<size_t> soff(W + 1, 0), roff(W + 1, 0);
        for (size_t t = 0; t < W; t++)
        {
            soff[t + 1] = soff[t] + mat[r * W + t];
            roff[t + 1] = roff[t] + mat[t * W + r];
        }
        if (mat[r * W + r]) // own piece: a copy
            MSVS_HIP(hipMemcpyAsync(d_recv + roff[r], d_send + soff[r], mat[r * W + r], hipMemcpyDeviceToDevice, stream));
        if (W == 1)
            return;
        if (!custom)
        {
            const RcclApi & api = rccl();
            if (!api.Send || !api.Recv || !api.GroupStart || !api.GroupEnd)
                msvs::fail(MSVS_ERR_DEVICE, "librccl lacks ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd");
            nccl_check(api.GroupStart(), "ncclGroupStart");
            for (size_t t = 0; t < W; t++)
            {
                if (t == r)
                    continue;
                if (mat[r * W + t])
                    nccl_check(api.Send(d_send + soff[t], mat[r * W + t], ncclInt8, (int)t, nccl, stream), "ncclSend");
                if (mat[t * W + r])
                    nccl_check(api.Recv(d_recv + roff[t], mat[t * W + r], ncclInt8, (int)t, nccl, stream), "ncclRecv");
            }
            nccl_check(api.GroupEnd(), "ncclGroupEnd");
            return;
        }
        size_t slot = 0;
        std::vector<size_t> row_off(W * (W + 1), 0); // rank s's send offsets
        for (size_t s = 0; s < W; s++)
        {
            for (size_t t = 0; t < W; t++)
                row_off[s * (W + 1) + t + 1] = row_off[s * (W + 1) + t] + mat[s * W + t];
            slot = std::max(slot, row_off[s * (W + 1) + W]);
        }
        slot = msvs::round_up(std::max<size_t>(slot, 16), (size_t)16);
        if (soff[W])
            MSVS_HIP(hipMemcpyAsync(tmp + r * slot, d_send, soff[W], hipMemcpyDeviceToDevice, stream));
        all_gather(tmp, slot, stream);
        for (size_t s = 0; s < W; s++)
            if (s != r && mat[s * W + r])
                MSVS_HIP(hipMemcpyAsync(d_recv + roff[s], tmp + s * slot + row_off[s * (W + 1) + r], mat[s * W + r], hipMemcpyDeviceToDevice, stream));
    }
    size_t exchange_tmp_bytes(const std::vector<size_t> & mat) const
    {
        if (!custom || nranks == 1)
            return 0;
        const size_t W = (size_t)nranks;
        size_t slot = 0;
        for (size_t s = 0; s < W; s++)
        {
            size_t sum = 0;
            for (size_t t = 0; t < W; t++)
                sum += mat[s * W + t];
            slot = std::max(slot, sum);
        }
        return W * msvs::round_up(std::max<size_t>(slot, 16), (size_t)16);
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

// =========================================================================================== routed search
//
// msvs_shard_search_device gives every rank the SAME batch: the list scan is sharded, but the per-query stages of the step (cut,
// selection, re-ranks, plans) run for the whole batch on every rank -- ~0.23 of 0.58 ms on the bench step, Amdahl ~1.8 x at 8
// ranks.  Here every rank brings its OWN batch (the queries that arrived at its server: StorageDistributed.cpp:1213-1255 sends a
// query to one replica of every shard) and a query visits only the ranks that own lists it still needs:
//   1. home rank: coarse quantiser of its own queries, then the pre-pruning by the list radius over the lists of the WHOLE index
//      (radius / length of every rank's lists and the extremes of the row norms are gathered once per index);
//   2. one all-gather of the W x W matrix of (source, destination) query counts, read back by the host (the one synchronisation
//      of the step: the point-to-point sizes must be known on both sides);
//   3. exchange: {query index, its surviving probes among the destination's lists, their coarse words, the query vector} go to
//      the ranks that own surviving lists -- ncclSend / ncclRecv of the exact sizes, grouped (xGMI is point to point: a query that
//      needs one rank travels over one link);
//   4. every rank searches what arrived (the given-probes form of the search: plan, sample, cut, pruning second stage, scan,
//      re-rank, certificate -- over its own lists only) and returns exact local top-k lists;
//   5. exchange back; the home rank merges the <= W lists of each of its queries canonically (getTotalTopSearchResultImpl,
//      MergeTreeBaseSearchManager.cpp:207-299) -- the result of the unsharded index, bit for bit.
// Per rank and step: its own nq queries' coarse stage + ~nq (1 + spill) routed queries' list scan, whatever W is.

namespace
{
constexpr uint32_t ROUTE_MAX_RANKS = 32; // destination masks are 32-bit words

/// mask[q] = ranks that own a surviving probe of query q; cnt[t] += queries that go to rank t.
__global__ void route_mask_kernel(const int32_t * probes, uint32_t nq, uint32_t np, uint32_t W, uint32_t * mask, uint32_t * cnt)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq)
        return;
    uint32_t m = 0;
    for (uint32_t j = 0; j < np; j++)
    {
        const int32_t l = probes[(size_t)q * np + j];
        if (l >= 0)
            m |= 1u << ((uint32_t)l % W);
    }
    mask[q] = m;
    for (uint32_t t = 0; t < W; t++)
        if ((m >> t) & 1u)
            atomicAdd(&cnt[t], 1u);
}

struct RoutePack
{
    const float * Q;         // [nq][d]
    const int32_t * probes;  // [nq][np] survivors (-1: dropped)
    const uint32_t * words;  // [nq][np]
    const uint32_t * mask;
    uint32_t nq, np, d, W;
    uint32_t * cursor;       // [W] zeroed
    const uint64_t * region; // [W] byte offset of destination t's region in `send`
    const uint32_t * cnt;    // [W] entries of the region
    unsigned char * send;
    uint32_t * sent_q;       // [sum cnt]: query of (destination, slot), destinations back to back
    const uint32_t * sent_off; // [W] first entry of destination t in sent_q
};

/// One wavefront per query: an entry in the region of every destination in its mask.  Region of cnt entries:
/// qidx u32[cnt] | probes i32[cnt][np] (the destination's lists only) | words u32[cnt][np] | vectors f32[cnt][d].
__global__ __launch_bounds__(64) void route_pack_kernel(const RoutePack a)
{
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    uint32_t m = a.mask[q];
    while (m)
    {
        const uint32_t t = (uint32_t)__builtin_ctz(m);
        m &= m - 1;
        uint32_t slot = 0;
        if (lane == 0)
            slot = atomicAdd(&a.cursor[t], 1u);
        slot = (uint32_t)__shfl((int)slot, 0);
        const uint32_t cnt = a.cnt[t];
        unsigned char * const reg = a.send + a.region[t];
        uint32_t * const qidx = reinterpret_cast<uint32_t *>(reg);
        int32_t * const pr = reinterpret_cast<int32_t *>(reg + (size_t)cnt * 4) + (size_t)slot * a.np;
        uint32_t * const wd = reinterpret_cast<uint32_t *>(reg + (size_t)cnt * 4 * (1 + a.np)) + (size_t)slot * a.np;
        float * const vec = reinterpret_cast<float *>(reg + (size_t)cnt * 4 * (1 + 2 * (size_t)a.np)) + (size_t)slot * a.d;
        if (lane == 0)
        {
            qidx[slot] = q;
            a.sent_q[a.sent_off[t] + slot] = q;
        }
        for (uint32_t j = lane; j < a.np; j += 64)
        {
            const int32_t l = a.probes[(size_t)q * a.np + j];
            const bool here = l >= 0 && (uint32_t)l % a.W == t;
            pr[j] = here ? l : -1;
            wd[j] = here ? a.words[(size_t)q * a.np + j] : 0xFFFFFFFFu;
        }
        for (uint32_t c = lane; c < a.d; c += 64)
            vec[c] = a.Q[(size_t)q * a.d + c];
    }
}

struct RouteUnpack
{
    const unsigned char * recv;
    uint64_t region[ROUTE_MAX_RANKS]; // byte offset of source s's region
    uint32_t cnt[ROUTE_MAX_RANKS], first[ROUTE_MAX_RANKS]; // its entries, its first row in the dense arrays
    uint32_t W, np, d;
    int32_t * probes;  // [n_in][np]
    uint32_t * words;  // [n_in][np]
    float * Q;         // [n_in][d]
};

/// One wavefront per received entry: the regions of the W sources -> dense arrays in source order.
__global__ __launch_bounds__(64) void route_unpack_kernel(const RouteUnpack a, uint32_t n_in)
{
    const uint32_t i = blockIdx.x, lane = threadIdx.x;
    uint32_t s = 0;
    while (s + 1 < a.W && i >= a.first[s + 1])
        s++;
    const uint32_t slot = i - a.first[s], cnt = a.cnt[s];
    const unsigned char * const reg = a.recv + a.region[s];
    const int32_t * const pr = reinterpret_cast<const int32_t *>(reg + (size_t)cnt * 4) + (size_t)slot * a.np;
    const uint32_t * const wd = reinterpret_cast<const uint32_t *>(reg + (size_t)cnt * 4 * (1 + a.np)) + (size_t)slot * a.np;
    const float * const vec = reinterpret_cast<const float *>(reg + (size_t)cnt * 4 * (1 + 2 * (size_t)a.np)) + (size_t)slot * a.d;
    for (uint32_t j = lane; j < a.np; j += 64)
    {
        a.probes[(size_t)i * a.np + j] = pr[j];
        a.words[(size_t)i * a.np + j] = wd[j];
    }
    for (uint32_t c = lane; c < a.d; c += 64)
        a.Q[(size_t)i * a.d + c] = vec[c];
}

/// The results that came back, destination t's block = {ids i64[cnt_t][k] | dis f32[cnt_t][k]} at back[t]: entry (t, slot) belongs
/// to query sent_q[sent_off[t] + slot] and becomes part t of its merge input (parts the query did not visit stay "no hit").
__global__ void route_scatter_kernel(const unsigned char * back, const uint64_t * region, const uint32_t * cnt, const uint32_t * sent_q,
                                     const uint32_t * sent_off, uint32_t W, uint32_t nq, uint32_t k, int64_t * m_ids, float * m_dis)
{
    const uint32_t t = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t c = cnt[t];
    if (i >= (size_t)c * k)
        return;
    const uint32_t slot = (uint32_t)(i / k), j = (uint32_t)(i - (size_t)slot * k);
    const uint32_t q = sent_q[sent_off[t] + slot];
    const unsigned char * const reg = back + region[t];
    m_ids[((size_t)t * nq + q) * k + j] = reinterpret_cast<const int64_t *>(reg)[i];
    m_dis[((size_t)t * nq + q) * k + j] = reinterpret_cast<const float *>(reg + (size_t)c * k * 8)[i];
}

__global__ void route_global_kernel(const float * radii /* [W][nlist] */, const uint32_t * lens /* [W][nlist] */, uint32_t W, uint32_t nlist,
                                    float * radius, int64_t * len64)
{
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nlist)
        return;
    float r = 0.f;
    int64_t n = 0;
    for (uint32_t s = 0; s < W; s++)
    {
        r = fmaxf(r, radii[(size_t)s * nlist + l]);
        n += lens[(size_t)s * nlist + l];
    }
    radius[l] = r;
    len64[l] = n;
}

__global__ void route_local_lens_kernel(const int64_t * list_off, uint32_t nlist, uint32_t * lens)
{
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l < nlist)
        lens[l] = (uint32_t)(list_off[l + 1] - list_off[l]);
}

/// Radius and length of every list of the whole index + the extremes of the row norms: one all-gather at the first routed search.
std::shared_ptr<msvs_index::Global> route_global(const msvs_index_t * ix, const msvs_comm_t * comm, hipStream_t stream)
{
    {
        std::lock_guard<std::mutex> lk(ix->meta_mu);
        if (ix->global)
            return ix->global;
    }
    const size_t W = (size_t)comm->nranks, r = (size_t)comm->rank, nl = ix->nlist;
    const size_t rec = round_up((2 * nl + 2) * 4, (size_t)16); // radius f32[nl] | len u32[nl] | xmax | xmin
    DevBuf<unsigned char> buf(W * rec);
    unsigned char * mine = buf.p + r * rec;
    if (ix->list_radius.p)
        MSVS_HIP(hipMemcpyAsync(mine, ix->list_radius.p, nl * 4, hipMemcpyDeviceToDevice, stream));
    else
        MSVS_HIP(hipMemsetAsync(mine, 0x7f, nl * 4, stream)); // (no radii: huge ones -- nothing can be pruned)
    hipLaunchKernelGGL(route_local_lens_kernel, dim3((unsigned)ceil_div(nl, (size_t)256)), dim3(256), 0, stream, ix->list_off.p, (uint32_t)nl,
                       reinterpret_cast<uint32_t *>(mine + nl * 4));
    const float ext[2] = {ix->xnorm_max, ix->xnorm_min};
    MSVS_HIP(hipMemcpyAsync(mine + 2 * nl * 4, ext, 8, hipMemcpyHostToDevice, stream));
    comm->all_gather(buf.p, rec, stream);
    std::vector<float> exts(2 * W);
    for (size_t s = 0; s < W; s++)
        MSVS_HIP(hipMemcpyAsync(&exts[2 * s], buf.p + s * rec + 2 * nl * 4, 8, hipMemcpyDeviceToHost, stream));
    auto g = std::make_shared<msvs_index::Global>();
    g->radius.alloc(nl);
    g->list_off.alloc(nl + 1);
    DevBuf<float> radii(W * nl);
    DevBuf<uint32_t> lens(W * nl);
    DevBuf<int64_t> len64(nl);
    MSVS_HIP(hipMemcpy2DAsync(radii.p, nl * 4, buf.p, rec, nl * 4, W, hipMemcpyDeviceToDevice, stream));
    MSVS_HIP(hipMemcpy2DAsync(lens.p, nl * 4, buf.p + nl * 4, rec, nl * 4, W, hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(route_global_kernel, dim3((unsigned)ceil_div(nl, (size_t)256)), dim3(256), 0, stream, radii.p, lens.p, (uint32_t)W, (uint32_t)nl,
                       g->radius.p, len64.p);
    std::vector<int64_t> h_len(nl), h_off(nl + 1, 0);
    MSVS_HIP(hipMemcpyAsync(h_len.data(), len64.p, nl * 8, hipMemcpyDeviceToHost, stream));
    MSVS_HIP(hipStreamSynchronize(stream));
    for (size_t l = 0; l < nl; l++)
        h_off[l + 1] = h_off[l] + h_len[l];
    MSVS_HIP(hipMemcpy(g->list_off.p, h_off.data(), (nl + 1) * 8, hipMemcpyHostToDevice));
    g->xmax = exts[0];
    g->xmin = exts[1];
    for (size_t s = 1; s < W; s++)
    {
        g->xmax = exts[2 * s] > g->xmax || exts[2 * s] != exts[2 * s] ? exts[2 * s] : g->xmax; // (NaN: unusable bounds, like a local NaN)
        g->xmin = exts[2 * s + 1] < g->xmin ? exts[2 * s + 1] : g->xmin;
    }
    std::lock_guard<std::mutex> lk(ix->meta_mu);
    ix->global = g;
    return g;
}
}

extern "C" int msvs_shard_search_routed_device(const msvs_index_t * ix, const msvs_comm_t * comm, const float * d_queries, size_t nq, int k,
                                               int nprobe, int64_t * d_ids, float * d_dis, void * hip_stream, uint64_t * routed_pairs)
{
    return guarded([&] {
        shard_check(ix, comm, d_queries, nq, k, d_ids, d_dis);
        if (ix->type != MSVS_INDEX_IVFFLAT)
            fail(MSVS_ERR_NOT_IMPLEMENTED, "the routed search serves IVFFLAT shards (a FLAT index has no lists to route by: msvs_shard_search_device)");
        if (comm->nranks > (int)ROUTE_MAX_RANKS)
            fail(MSVS_ERR_INVALID_ARGUMENT, "the routed search serves up to %u ranks", ROUTE_MAX_RANKS);
        if (k == 0)
            return; // (nq = 0 is a valid contribution: the rank still serves the others' queries)
        check_k((size_t)k);
        hipStream_t cs = as_stream(hip_stream);
        const auto meta = ix->get_meta();
        if (meta && (meta->delete_nbits || meta->row_ids_n))
            fail(MSVS_ERR_NOT_IMPLEMENTED, "the routed search does not take delete bitmaps / row id maps yet (msvs_shard_search_device does)");
        const size_t W = (size_t)comm->nranks, r = (size_t)comm->rank, d = ix->dim;
        const size_t np = std::min<size_t>(std::max(nprobe, 1), ix->nlist);
        const auto g = route_global(ix, comm, cs);
        Scratch & sh = shard_for(cs);
        // ---- 1. own queries: probes, words, survivors; destination masks and counts
        sh.reserve(nq * np * 12 + nq * 4 + W * W * 4 + 4096 + 64 * W, cs);
        int32_t * probes = sh.take<int32_t>(std::max<size_t>(nq * np, 1));
        uint32_t * words = sh.take<uint32_t>(std::max<size_t>(nq * np, 1));
        int32_t * alive_probes = sh.take<int32_t>(std::max<size_t>(nq * np, 1));
        uint32_t * mask = sh.take<uint32_t>(std::max<size_t>(nq, 1));
        uint32_t * cmat = sh.take<uint32_t>(W * W);
        uint32_t * cursor = sh.take<uint32_t>(W);
        MSVS_HIP(hipMemsetAsync(cmat + r * W, 0, W * 4, cs));
        MSVS_HIP(hipMemsetAsync(cursor, 0, W * 4, cs));
        if (nq)
        {
            ProbeWords pw{};
            pw.out = words;
            pw.g_radius = g->radius.p;
            pw.g_list_off = g->list_off.p;
            pw.g_xmax = g->xmax;
            pw.g_xmin = g->xmin;
            pw.pruned_out = alive_probes;
            index_search_device(*ix, d_queries, nq, (uint32_t)k, np, nullptr, 0, nullptr, nullptr, cs, nullptr, probes, nullptr, pw);
            hipLaunchKernelGGL(route_mask_kernel, dim3((unsigned)ceil_div(nq, (size_t)256)), dim3(256), 0, cs, alive_probes, (uint32_t)nq, (uint32_t)np,
                               (uint32_t)W, mask, cmat + r * W);
            MSVS_HIP(hipGetLastError());
        }
        // ---- 2. the count matrix on every host
        comm->all_gather(reinterpret_cast<unsigned char *>(cmat), W * 4, cs);
        std::vector<uint32_t> h_c(W * W);
        MSVS_HIP(hipMemcpyAsync(h_c.data(), cmat, W * W * 4, hipMemcpyDeviceToHost, cs));
        MSVS_HIP(hipStreamSynchronize(cs));
        const size_t ent = 4 * (1 + 2 * np + d); // bytes of one routed entry
        std::vector<size_t> mat1(W * W), mat2(W * W);
        size_t n_out = 0, n_in = 0;
        for (size_t s = 0; s < W; s++)
            for (size_t t = 0; t < W; t++)
            {
                mat1[s * W + t] = (size_t)h_c[s * W + t] * ent;
                mat2[t * W + s] = (size_t)h_c[s * W + t] * (size_t)k * 12; // the results travel the other way
            }
        std::vector<uint64_t> h_sreg(W), h_rreg(W), h_breg(W);
        std::vector<uint32_t> h_soff(W), h_scnt(W);
        size_t sbytes = 0, rbytes = 0, bbytes = 0;
        RouteUnpack un{};
        for (size_t t = 0; t < W; t++)
        {
            h_sreg[t] = sbytes;
            h_breg[t] = bbytes;
            h_soff[t] = (uint32_t)n_out;
            h_scnt[t] = h_c[r * W + t];
            sbytes += mat1[r * W + t];
            bbytes += (size_t)h_c[r * W + t] * (size_t)k * 12;
            n_out += h_c[r * W + t];
            un.region[t] = rbytes;
            un.cnt[t] = h_c[t * W + r];
            un.first[t] = (uint32_t)n_in;
            rbytes += mat1[t * W + r];
            n_in += h_c[t * W + r];
        }
        if (routed_pairs)
            *routed_pairs = n_in;
        // ---- 3. pack and exchange
        Scratch & sx = route_for(cs);
        const size_t res_bytes = n_in * (size_t)k * 12;
        sx.reserve(sbytes + rbytes + 2 * comm->exchange_tmp_bytes(mat1) + n_out * 4 + 5 * W * 8 + n_in * (np * 8 + d * 4) + 2 * res_bytes + bbytes
                       + comm->exchange_tmp_bytes(mat2) + W * nq * (size_t)k * 12 + 65536,
                   cs);
        unsigned char * send = sx.take<unsigned char>(std::max<size_t>(sbytes, 16));
        unsigned char * recv = sx.take<unsigned char>(std::max<size_t>(rbytes, 16));
        unsigned char * tmp1 = sx.take<unsigned char>(std::max<size_t>(comm->exchange_tmp_bytes(mat1), 16));
        uint32_t * sent_q = sx.take<uint32_t>(std::max<size_t>(n_out, 1));
        uint64_t * d_sreg = sx.take<uint64_t>(W);
        uint64_t * d_breg = sx.take<uint64_t>(W);
        uint32_t * d_soff = sx.take<uint32_t>(W);
        uint32_t * d_scnt = sx.take<uint32_t>(W);
        MSVS_HIP(hipMemcpyAsync(d_sreg, h_sreg.data(), W * 8, hipMemcpyHostToDevice, cs));
        MSVS_HIP(hipMemcpyAsync(d_breg, h_breg.data(), W * 8, hipMemcpyHostToDevice, cs));
        MSVS_HIP(hipMemcpyAsync(d_soff, h_soff.data(), W * 4, hipMemcpyHostToDevice, cs));
        MSVS_HIP(hipMemcpyAsync(d_scnt, h_scnt.data(), W * 4, hipMemcpyHostToDevice, cs));
        if (nq)
        {
            RoutePack pk{};
            pk.Q = d_queries;
            pk.probes = alive_probes;
            pk.words = words;
            pk.mask = mask;
            pk.nq = (uint32_t)nq;
            pk.np = (uint32_t)np;
            pk.d = (uint32_t)d;
            pk.W = (uint32_t)W;
            pk.cursor = cursor;
            pk.region = d_sreg;
            pk.cnt = d_scnt;
            pk.send = send;
            pk.sent_q = sent_q;
            pk.sent_off = d_soff;
            hipLaunchKernelGGL(route_pack_kernel, dim3((unsigned)nq), dim3(64), 0, cs, pk);
            MSVS_HIP(hipGetLastError());
        }
        {
            ProfileScope prof("shard_exchange", cs);
            comm->exchange(send, recv, mat1, tmp1, cs);
        }
        // ---- 4. what arrived, searched over this rank's lists
        unsigned char * res = sx.take<unsigned char>(std::max<size_t>(res_bytes, 16)); // source s's block: ids[cnt][k] | dis[cnt][k]
        if (n_in)
        {
            int32_t * rp = sx.take<int32_t>(n_in * np);
            uint32_t * rw = sx.take<uint32_t>(n_in * np);
            float * rq = sx.take<float>(n_in * d);
            int64_t * r_ids = sx.take<int64_t>(n_in * (size_t)k);
            float * r_dis = sx.take<float>(n_in * (size_t)k);
            un.recv = recv;
            un.W = (uint32_t)W;
            un.np = (uint32_t)np;
            un.d = (uint32_t)d;
            un.probes = rp;
            un.words = rw;
            un.Q = rq;
            hipLaunchKernelGGL(route_unpack_kernel, dim3((unsigned)n_in), dim3(64), 0, cs, un, (uint32_t)n_in);
            MSVS_HIP(hipGetLastError());
            ProbeWords gw{};
            gw.given = rw;
            index_search_device(*ix, rq, n_in, (uint32_t)k, np, nullptr, 0, r_ids, r_dis, cs, rp, nullptr, nullptr, gw);
            // per source: {ids | dis} blocks, the layout of the way back
            size_t off = 0;
            for (size_t s = 0; s < W; s++)
            {
                const size_t c = un.cnt[s];
                if (!c)
                    continue;
                MSVS_HIP(hipMemcpyAsync(res + off, r_ids + (size_t)un.first[s] * k, c * (size_t)k * 8, hipMemcpyDeviceToDevice, cs));
                MSVS_HIP(hipMemcpyAsync(res + off + c * (size_t)k * 8, r_dis + (size_t)un.first[s] * k, c * (size_t)k * 4, hipMemcpyDeviceToDevice, cs));
                off += c * (size_t)k * 12;
            }
        }
        // ---- 5. results back to their home ranks, merged there
        unsigned char * back = sx.take<unsigned char>(std::max<size_t>(bbytes, 16));
        unsigned char * tmp2 = sx.take<unsigned char>(std::max<size_t>(comm->exchange_tmp_bytes(mat2), 16));
        {
            ProfileScope prof("shard_exchange", cs);
            comm->exchange(res, back, mat2, tmp2, cs);
        }
        if (!nq)
            return;
        int64_t * m_ids = sx.take<int64_t>(W * nq * (size_t)k);
        float * m_dis = sx.take<float>(W * nq * (size_t)k);
        MSVS_HIP(hipMemsetAsync(m_ids, 0xFF, W * nq * (size_t)k * 8, cs)); // -1: no hit
        MSVS_HIP(hipMemsetAsync(m_dis, 0, W * nq * (size_t)k * 4, cs));
        size_t cmax = 0;
        for (size_t t = 0; t < W; t++)
            cmax = std::max<size_t>(cmax, h_scnt[t]);
        if (cmax)
        {
            hipLaunchKernelGGL(route_scatter_kernel, dim3((unsigned)ceil_div(cmax * (size_t)k, (size_t)256), (unsigned)W), dim3(256), 0, cs, back, d_breg,
                               d_scnt, sent_q, d_soff, (uint32_t)W, (uint32_t)nq, (uint32_t)k, m_ids, m_dis);
            MSVS_HIP(hipGetLastError());
        }
        const int order = ix->metric == MSVS_METRIC_IP ? MSVS_METRIC_IP : MSVS_METRIC_L2; // (cosine distances leave the search as 1 - ip)
        merge_topk_device(m_ids, nq * (size_t)k, m_dis, nq * (size_t)k, W, nq, (size_t)k, order, d_ids, d_dis, cs);
    });
}
