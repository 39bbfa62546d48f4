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
    if (nparts * k <= 256 && nq <= 0xfffffff0ull && options().merge_small != 0)
    {
        // a few short lists per query: one wavefront ranks them (merge_small_kernel)
        const dim3 grid((unsigned)ceil_div(nq, (size_t)(BLOCK / WAVE)));
        if (metric == MSVS_METRIC_IP)
            hipLaunchKernelGGL((merge_small_kernel<M_IP>), grid, dim3(BLOCK), 0, stream, d_ids, ids_stride, d_dis, dis_stride, (uint32_t)nparts,
                               (uint32_t)nq, (uint32_t)k, d_out_ids, d_out_dis);
        else
            hipLaunchKernelGGL((merge_small_kernel<M_L2>), grid, dim3(BLOCK), 0, stream, d_ids, ids_stride, d_dis, dis_stride, (uint32_t)nparts,
                               (uint32_t)nq, (uint32_t)k, d_out_ids, d_out_dis);
        MSVS_HIP(hipGetLastError());
        return;
    }
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
        const bool self_rccl = nccl && !custom && msvs::options().route_self_rccl != 0; // (tests: the own piece through the group path too)
        if (mat[r * W + r] && !self_rccl) // own piece: a copy
            MSVS_HIP(hipMemcpyAsync(d_recv + roff[r], d_send + soff[r], mat[r * W + r], hipMemcpyDeviceToDevice, stream));
        if (W == 1 && !self_rccl)
            return;
        if (!custom)
        {
            const RcclApi & api = rccl();
            if (!api.Send || !api.Recv || !api.GroupStart || !api.GroupEnd)
                msvs::fail(MSVS_ERR_DEVICE, "librccl lacks ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd");
            nccl_check(api.GroupStart(), "ncclGroupStart");
            for (size_t t = 0; t < W; t++)
            {
                if (t == r && !self_rccl)
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

static void forget_pipes(const msvs_comm_t * c);

extern "C" void msvs_comm_free(msvs_comm_t * c)
{
    if (!c)
        return;
    forget_pipes(c); // (a later communicator may live at the same address with another size)
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

static std::mutex g_pipes_mu;
static std::map<const msvs_comm_t *, std::unique_ptr<ShardPipe>> g_pipes; // (entries dropped by msvs_comm_free)

static ShardPipe * pipe_of(const msvs_comm_t * comm, bool create)
{
    std::lock_guard<std::mutex> lk(g_pipes_mu);
    if (!create)
    {
        auto it = g_pipes.find(comm);
        return it == g_pipes.end() ? nullptr : it->second.get();
    }
    auto & p = g_pipes[comm];
    if (!p)
    {
        p.reset(new ShardPipe);
        MSVS_HIP(hipStreamCreateWithFlags(&p->compute, hipStreamNonBlocking));
        MSVS_HIP(hipStreamCreateWithFlags(&p->xchg, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++)
            for (hipEvent_t * e : {&p->in_ev[i], &p->ab_ev[i], &p->x1_ev[i], &p->b_ev[i], &p->done_ev[i]})
                MSVS_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    return p.get();
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
        ShardPipe & pp = *pipe_of(comm, true);
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

// =========================================================================================== routed search (rounds 5-6)
//
// msvs_shard_search_device gives every rank the SAME batch: the list scan is sharded, but the per-query stages of the step (cut,
// selection, re-ranks, plans) run for the whole batch on every rank -- ~0.23 of 0.58 ms on the bench step, Amdahl ~1.8 x at 8
// ranks.  Here every rank brings its OWN batch (the queries that arrived at its server: StorageDistributed.cpp:1213-1255 sends a
// query to one replica of every shard) and a query visits only the ranks that own lists it still needs.  A step has two phases:
//   FRONT 1. home rank: coarse quantiser of its own queries, then the pre-pruning by the list radius over the lists of the WHOLE
//            index (radius / length of every rank's lists and the extremes of the row norms: msvs_index::Global, gathered when any
//            rank asks for it);
//         2. one all-gather of the W x (W + 6) matrix: (source, destination) query counts + a header per rank -- status of its
//            front phase, "I have no Global", "I search under a filter / delete bitmap", "I pre-pruned", the instance of its shard
//            object -- copied to pinned host memory;
//   BACK  (the host has read the matrix: every rank takes the SAME decision from it -- fail together when any rank failed, gather
//         the Global again and redo the front phase when a rank asks or a shard object changed, redo it without the pre-pruning
//         when some rank pre-pruned while another filters: the radius bound counts rows a filter may hide)
//         3. exchange: {query index, its surviving probes among the destination's lists, their coarse words, the query vector} go
//            to the ranks that own surviving lists -- ncclSend / ncclRecv of the exact sizes, grouped (xGMI is point to point: a
//            query that needs one rank travels over one link);
//         4. every rank searches what arrived over its own lists under ITS filter (per-search bitmap AND resident delete bitmap,
//            VIWithDataPart.cpp:903-908) and returns exact local top-k lists;
//         5. exchange back; the home rank merges the <= W lists of each of its queries canonically (getTotalTopSearchResultImpl,
//            MergeTreeBaseSearchManager.cpp:207-299) and applies its row-id map -- the result of the unsharded index, bit for bit.
// msvs_shard_search_routed_device runs both phases back to back on the caller's stream (one host synchronisation in between).
// msvs_shard_search_routed_device_async keeps TWO STEPS IN FLIGHT: call i enqueues FRONT(i), then BACK(i - 1) -- whose matrix was
// gathered one call ago, so the host does not wait for it while the GPU still has the list scan of step i - 1 and the coarse stage
// of step i to run.  Every collective of a communicator is issued on its ONE exchange stream in call order, the same on every rank.

namespace
{
constexpr uint32_t ROUTE_MAX_RANKS = 32; // destination masks are 32-bit words
constexpr uint32_t ROUTE_HDR = 8;        // words after a rank's W counts: status | need | filtered | pruned | instance lo | hi | k | nprobe
enum { RH_STATUS = 0, RH_NEED, RH_FILTERED, RH_PRUNED, RH_INST_LO, RH_INST_HI, RH_K, RH_NP };
struct RouteHdr
{
    uint32_t w[ROUTE_HDR];
};

/// Per-rank tables a step's kernels need, passed by value (<= 32 ranks: 512 B of kernel arguments, no H2D copies in the step).
struct RouteTab
{
    uint64_t region[ROUTE_MAX_RANKS]; // byte offset of rank t's region in the buffer the kernel walks
    uint32_t cnt[ROUTE_MAX_RANKS];    // its entries
    uint32_t first[ROUTE_MAX_RANKS];  // its first entry in the dense (rank-major) arrays
};

/// The count-matrix row of this rank before the front phase fills it: counts and pack cursors zeroed, the header in place.
__global__ void route_init_kernel(uint32_t * row, uint32_t W, uint32_t * cursor, const RouteHdr h)
{
    const uint32_t t = threadIdx.x;
    if (t < W)
    {
        row[t] = 0;
        cursor[t] = 0;
    }
    if (t < ROUTE_HDR)
        row[W + t] = h.w[t];
}

/// The gathered count matrix to the host: a kernel that writes pinned memory and then a completion word (system-scope release), instead
/// of hipMemcpyAsync + an event -- two barrier packets, ~10 us of idle device each (what the BM25 batch's hand-over measured); the host
/// spins on the word.  n <= 32 * 40 words: one block.
__global__ void route_matrix_out_kernel(const uint32_t * cmat, uint32_t * h_c, uint32_t n, uint32_t * h_flag, uint32_t seq)
{
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
        h_c[i] = cmat[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(h_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void route_fill_kernel(int64_t * ids, float * dis, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    {
        ids[i] = -1;
        dis[i] = 0.f;
    }
}

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
    unsigned char * send;
    uint32_t * sent_q;       // [sum cnt]: query of (destination, slot), destinations back to back
    RouteTab to;             // region / cnt / first of every destination
};

/// An entry in the region of every destination in a query's mask.  Region of cnt entries:
/// qidx u32[cnt] | probes i32[cnt][np] (the destination's lists only) | words u32[cnt][np] | vectors f32[cnt][d].
/// A workgroup takes 16 queries: their slots in the regions are reserved with ONE atomic per destination and workgroup (a returning
/// atomic per query and destination on the same word serialises in L2: 54 us for 4096 queries going to one rank, 25 to two), then each
/// of the four wavefronts copies four queries.
constexpr uint32_t ROUTE_PACK_Q = 16;
__global__ __launch_bounds__(256) void route_pack_kernel(const RoutePack a)
{
    __shared__ uint32_t s_slot[ROUTE_PACK_Q][ROUTE_MAX_RANKS];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q0 = blockIdx.x * ROUTE_PACK_Q;
    if (wave == 0)
    {
        const uint32_t mq = lane < ROUTE_PACK_Q && q0 + lane < a.nq ? a.mask[q0 + lane] : 0u;
        for (uint32_t t = 0; t < a.W; t++)
        {
            const bool has = (mq >> t) & 1u;
            const uint64_t b = __ballot(has);
            if (!b)
                continue;
            uint32_t base = 0;
            if (lane == 0)
                base = atomicAdd(&a.cursor[t], (uint32_t)__popcll(b));
            base = (uint32_t)__shfl((int)base, 0);
            if (has)
                s_slot[lane][t] = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
        }
    }
    __syncthreads();
    for (uint32_t i = 0; i < ROUTE_PACK_Q / 4; i++)
    {
        const uint32_t qi = wave * (ROUTE_PACK_Q / 4) + i, q = q0 + qi;
        if (q >= a.nq)
            break;
        uint32_t m = a.mask[q];
        while (m)
        {
            const uint32_t t = (uint32_t)__builtin_ctz(m);
            m &= m - 1;
            const uint32_t slot = s_slot[qi][t];
            const uint32_t cnt = a.to.cnt[t];
            unsigned char * const reg = a.send + a.to.region[t];
            uint32_t * const qidx = reinterpret_cast<uint32_t *>(reg);
            int32_t * const pr = reinterpret_cast<int32_t *>(reg + (size_t)cnt * 4) + (size_t)slot * a.np;
            uint32_t * const wd = reinterpret_cast<uint32_t *>(reg + (size_t)cnt * 4 * (1 + a.np)) + (size_t)slot * a.np;
            float * const vec = reinterpret_cast<float *>(reg + (size_t)cnt * 4 * (1 + 2 * (size_t)a.np)) + (size_t)slot * a.d;
            if (lane == 0)
            {
                qidx[slot] = q;
                a.sent_q[a.to.first[t] + slot] = q;
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
}

struct RouteUnpack
{
    const unsigned char * recv;
    RouteTab from; // region / cnt / first of every source
    uint32_t W, np, d;
    int32_t * probes;  // [n_in][np]
    uint32_t * words;  // [n_in][np]
    float * Q;         // [n_in][d]
};

__device__ inline uint32_t route_owner(const RouteTab & t, uint32_t W, uint32_t i)
{
    uint32_t s = 0;
    while (s + 1 < W && i >= t.first[s + 1])
        s++;
    return s;
}

/// One wavefront per received entry: the regions of the W sources -> dense arrays in source order.
__global__ __launch_bounds__(64) void route_unpack_kernel(const RouteUnpack a, uint32_t n_in)
{
    const uint32_t i = blockIdx.x, lane = threadIdx.x;
    const uint32_t s = route_owner(a.from, a.W, i);
    const uint32_t slot = i - a.from.first[s], cnt = a.from.cnt[s];
    const unsigned char * const reg = a.recv + a.from.region[s];
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

/// The way back: the dense results of the entries this rank served -> one block per source, {ids i64[cnt][k] | dis f32[cnt][k]}
/// at a 16-byte aligned offset (back.region).
__global__ void route_result_kernel(const int64_t * r_ids, const float * r_dis, uint32_t n_in, uint32_t k, uint32_t W, const RouteTab back,
                                    unsigned char * res)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_in * k)
        return;
    const uint32_t e = (uint32_t)(i / k), j = (uint32_t)(i - (size_t)e * k);
    const uint32_t s = route_owner(back, W, e);
    const uint32_t slot = e - back.first[s], cnt = back.cnt[s];
    unsigned char * const reg = res + back.region[s];
    reinterpret_cast<int64_t *>(reg)[(size_t)slot * k + j] = r_ids[i];
    reinterpret_cast<float *>(reg + (size_t)cnt * k * 8)[(size_t)slot * k + j] = r_dis[i];
}

/// The results that came back, destination t's block at back + home.region[t]: entry (t, slot) belongs to query
/// sent_q[home.first[t] + slot] and becomes part t of its merge input (parts the query did not visit stay "no hit").
__global__ void route_scatter_kernel(const unsigned char * back, const RouteTab home, const uint32_t * sent_q, uint32_t nq, uint32_t k,
                                     int64_t * m_ids, float * m_dis)
{
    const uint32_t t = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t c = home.cnt[t];
    if (i >= (size_t)c * k)
        return;
    const uint32_t slot = (uint32_t)(i / k), j = (uint32_t)(i - (size_t)slot * k);
    const uint32_t q = sent_q[home.first[t] + slot];
    const unsigned char * const reg = back + home.region[t];
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

/// Radius and length of every list of the whole index + the extremes of the row norms, and the shard instances they were gathered
/// from.  COLLECTIVE (one all-gather): the back phase calls it on every rank when the count matrix says so; synchronises `stream`.
std::shared_ptr<msvs_index::Global> route_global(const msvs_index_t * ix, const msvs_comm_t * comm, const std::vector<uint64_t> & instances,
                                                 hipStream_t stream)
{
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
    g->instances = instances;
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

/// One routed step between its two phases.
struct RoutedStep
{
    bool pending = false;
    // the call
    const msvs_index_t * ix = nullptr;
    const float * d_queries = nullptr;
    size_t nq = 0, np = 0, nbits = 0;
    int k = 0, nprobe = 0;
    const uint64_t * d_alive = nullptr;
    int64_t * d_ids = nullptr;
    float * d_dis = nullptr;
    uint64_t * routed_pairs = nullptr;
    // what the front phase left (arena `front`)
    int32_t * probes = nullptr;
    uint32_t * words = nullptr;
    int32_t * alive_probes = nullptr;
    uint32_t * mask = nullptr;
    uint32_t * cursor = nullptr;
    bool pruned = false;
    int status = MSVS_OK;
    std::string error;
    // owned
    Scratch front;
    uint32_t * cmat = nullptr; // device [W][W + ROUTE_HDR]
    uint32_t * h_c = nullptr;  // pinned, the same
    uint32_t a_seq = 0;            // the completion word's value for the front phase in flight (h_c[W * RW])
    hipStream_t a_stream = nullptr; // ... and the stream it was enqueued on
    hipEvent_t in_ev = nullptr, a_ev = nullptr, done_ev = nullptr;
};

/// The routed search's state on a communicator: the two streams and hand-over events of the pipelined form, two steps' slots
/// (the stream-at-a-time form uses slot 0 and the caller's stream), the exchange arena, and what the last count matrix said about
/// filters on the ranks (a hint for the next front phase; the matrix of the step itself decides).
struct RoutePipe
{
    hipStream_t compute = nullptr, xchg = nullptr;
    hipEvent_t hand[2] = {nullptr, nullptr};
    RoutedStep step[2];
    Scratch back;
    uint64_t calls = 0;
    bool peers_filtered = false;
    // route_streams = 1: the pipelined steps run on the CALLER's stream (no internal stream, no events while the caller stays on one
    // stream); `last` = the stream the step in flight was enqueued on
    hipStream_t last = nullptr;
    bool has_last = false;
    std::mutex mu;
};

std::mutex g_route_pipes_mu;
std::map<const msvs_comm_t *, std::unique_ptr<RoutePipe>> g_route_pipes; // (entries dropped by msvs_comm_free)

RoutePipe * route_pipe_of(const msvs_comm_t * comm, bool create)
{
    std::lock_guard<std::mutex> lk(g_route_pipes_mu);
    if (!create)
    {
        auto it = g_route_pipes.find(comm);
        return it == g_route_pipes.end() ? nullptr : it->second.get();
    }
    auto & p = g_route_pipes[comm];
    if (!p)
    {
        p.reset(new RoutePipe);
        const size_t W = (size_t)comm->nranks, bytes = W * (W + ROUTE_HDR) * 4;
        MSVS_HIP(hipStreamCreateWithFlags(&p->compute, hipStreamNonBlocking));
        MSVS_HIP(hipStreamCreateWithFlags(&p->xchg, hipStreamNonBlocking));
        for (hipEvent_t * e : {&p->hand[0], &p->hand[1]})
            MSVS_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        for (RoutedStep & st : p->step)
        {
            MSVS_HIP(hipMalloc(reinterpret_cast<void **>(&st.cmat), bytes));
            MSVS_HIP(hipHostMalloc(reinterpret_cast<void **>(&st.h_c), bytes + 64, hipHostMallocCoherent)); // (+ the completion word)
            memset(st.h_c, 0, bytes + 64);
            for (hipEvent_t * e : {&st.in_ev, &st.a_ev, &st.done_ev})
                MSVS_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        }
    }
    return p.get();
}

void hand_over(RoutePipe & pp, int which, hipStream_t from, hipStream_t to)
{
    if (from == to)
        return;
    MSVS_HIP(hipEventRecord(pp.hand[which], from));
    MSVS_HIP(hipStreamWaitEvent(to, pp.hand[which], 0));
}

/// A reservation in an arena that work on BOTH streams may still read.
void reserve_two(Scratch & sc, size_t bytes, hipStream_t cs, hipStream_t xs)
{
    if (xs != cs && (bytes > sc.buf.n || sc.ops >= 63))
        MSVS_HIP(hipStreamSynchronize(xs));
    sc.reserve(bytes, cs);
}

/// FRONT: coarse stage + pre-pruning of the rank's own queries on `cs`, this rank's row of the count matrix, the all-gather and its
/// copy to pinned memory on `xs`.  A failure of the rank's own part is PUBLISHED in the row instead of thrown: the collective is
/// issued whatever happened, the back phase fails on every rank together.
void routed_front(RoutePipe & pp, RoutedStep & st, const msvs_comm_t * comm, bool allow_prune, hipStream_t cs, hipStream_t xs)
{
    const size_t W = (size_t)comm->nranks, r = (size_t)comm->rank, RW = W + ROUTE_HDR;
    uint32_t * const row = st.cmat + r * RW;
    std::shared_ptr<msvs_index::Global> g;
    st.status = MSVS_OK;
    st.error.clear();
    st.pruned = false;
    bool filtered = false;
    uint64_t inst = 0;
    bool row_ready = false;
    try
    {
        const msvs_index_t * ix = st.ix;
        shard_check(ix, comm, st.d_queries, st.nq, st.k, st.d_ids, st.d_dis);
        if (ix->type != MSVS_INDEX_IVFFLAT)
            fail(MSVS_ERR_NOT_IMPLEMENTED, "the routed search serves IVFFLAT shards (a FLAT index has no lists to route by: msvs_shard_search_device)");
        if (!ix->ready)
            fail(MSVS_ERR_NOT_READY, "index is not ready");
        if (st.k)
            check_k((size_t)st.k);
        inst = ix->instance;
        st.np = std::min<size_t>(std::max(st.nprobe, 1), ix->nlist);
        const auto meta = ix->get_meta();
        filtered = st.d_alive != nullptr || (meta && meta->delete_nbits);
        {
            std::lock_guard<std::mutex> lk(ix->meta_mu);
            g = ix->global;
        }
        const size_t nq = st.k ? st.nq : 0, np = st.np;
        reserve_two(st.front, nq * np * 12 + nq * 4 + 4096 + 64 * W, cs, xs);
        st.probes = st.front.take<int32_t>(std::max<size_t>(nq * np, 1));
        st.words = st.front.take<uint32_t>(std::max<size_t>(nq * np, 1));
        st.alive_probes = st.front.take<int32_t>(std::max<size_t>(nq * np, 1));
        st.mask = st.front.take<uint32_t>(std::max<size_t>(nq, 1));
        st.cursor = st.front.take<uint32_t>(W);
        // the pre-pruning bounds the k-th distance by rows it has not seen: every row of a list counts, so nobody may filter
        const bool prune = allow_prune && g && !filtered && !pp.peers_filtered && nq;
        hipLaunchKernelGGL(route_init_kernel, dim3(1), dim3(64), 0, cs, row, (uint32_t)W, st.cursor,
                           RouteHdr{{(uint32_t)MSVS_OK, g ? 0u : 1u, filtered ? 1u : 0u, prune ? 1u : 0u, (uint32_t)inst, (uint32_t)(inst >> 32),
                                     (uint32_t)st.k, (uint32_t)np}});
        MSVS_HIP(hipGetLastError());
        row_ready = true;
        if (nq)
        {
            ProbeWords pw{};
            pw.out = st.words;
            if (prune)
            {
                pw.g_radius = g->radius.p;
                pw.g_list_off = g->list_off.p;
                pw.g_xmax = g->xmax;
                pw.g_xmin = g->xmin;
            }
            pw.pruned_out = st.alive_probes; // (no Global / no pruning: every probe survives)
            index_search_device(*ix, st.d_queries, nq, (uint32_t)st.k, np, nullptr, 0, nullptr, nullptr, cs, nullptr, st.probes, nullptr, pw);
            hipLaunchKernelGGL(route_mask_kernel, dim3((unsigned)ceil_div(nq, (size_t)256)), dim3(256), 0, cs, st.alive_probes, (uint32_t)nq,
                               (uint32_t)np, (uint32_t)W, st.mask, row);
            MSVS_HIP(hipGetLastError());
        }
        st.pruned = prune;
    }
    catch (const Error & e)
    {
        st.status = e.code ? e.code : MSVS_ERR_DEVICE;
        st.error = e.msg;
    }
    catch (const std::bad_alloc &)
    {
        st.status = MSVS_ERR_OUT_OF_MEMORY;
        st.error = "out of host memory";
    }
    if (st.status != MSVS_OK || !row_ready)
    {
        // nothing of this rank's queries travels; the others learn why
        if (!st.cursor)
            st.cursor = row; // (any W words: the row itself, rewritten below)
        hipLaunchKernelGGL(route_init_kernel, dim3(1), dim3(64), 0, cs, row, (uint32_t)W, row,
                           RouteHdr{{(uint32_t)(st.status ? st.status : MSVS_ERR_DEVICE), 0u, 0u, 0u, (uint32_t)inst, (uint32_t)(inst >> 32), 0u, 0u}});
        MSVS_HIP(hipGetLastError());
    }
    hand_over(pp, 0, cs, xs);
    comm->all_gather(reinterpret_cast<unsigned char *>(st.cmat), RW * 4, xs);
    st.a_seq = st.a_seq + 1 ? st.a_seq + 1 : 1; // never 0
    hipLaunchKernelGGL(route_matrix_out_kernel, dim3(1), dim3(256), 0, xs, st.cmat, st.h_c, (uint32_t)(W * RW), st.h_c + W * RW, st.a_seq);
    MSVS_HIP(hipGetLastError());
    st.a_stream = xs;
}

/// BACK: see the banner.  Returns after everything is enqueued; the results are complete when `cs` has run dry (st.done_ev).
void routed_back(RoutePipe & pp, RoutedStep & st, const msvs_comm_t * comm, hipStream_t cs, hipStream_t xs)
{
    const msvs_index_t * ix = st.ix;
    const size_t W = (size_t)comm->nranks, r = (size_t)comm->rank, RW = W + ROUTE_HDR;
    std::vector<uint32_t> h_c(W * W);
    bool senders_pruned = false; // (some rank's front phase pre-pruned what it sends)
    for (int attempt = 0;; attempt++)
    {
        // the matrix of this step (front phase): wait for its completion word
        for (uint64_t spins = 1; __atomic_load_n(st.h_c + W * RW, __ATOMIC_ACQUIRE) != st.a_seq; spins++)
        {
            __builtin_ia32_pause();
            if ((spins & 0xfff) == 0)
            {
                const hipError_t e = hipStreamQuery(st.a_stream);
                if (e == hipSuccess)
                    break;
                if (e != hipErrorNotReady)
                    fail(MSVS_ERR_DEVICE, "routed search: %s", hipGetErrorString(e));
            }
        }
        if (__atomic_load_n(st.h_c + W * RW, __ATOMIC_ACQUIRE) != st.a_seq)
        {
            MSVS_HIP(hipStreamSynchronize(st.a_stream));
            if (__atomic_load_n(st.h_c + W * RW, __ATOMIC_ACQUIRE) != st.a_seq)
                fail(MSVS_ERR_DEVICE, "routed search: the count matrix never arrived (launch lost)");
        }
        // ---- the decision every rank takes alike
        int peer_status = MSVS_OK;
        size_t peer = 0;
        bool need = false, any_filtered = false, any_pruned = false;
        std::vector<uint64_t> inst(W);
        for (size_t s = 0; s < W; s++)
        {
            const uint32_t * h = st.h_c + s * RW + W;
            if (h[RH_STATUS] != MSVS_OK && peer_status == MSVS_OK)
            {
                peer_status = (int)h[RH_STATUS];
                peer = s;
            }
            need |= h[RH_NEED] != 0;
            any_filtered |= h[RH_FILTERED] != 0;
            any_pruned |= h[RH_PRUNED] != 0;
            inst[s] = (uint64_t)h[RH_INST_LO] | ((uint64_t)h[RH_INST_HI] << 32);
        }
        if (st.status != MSVS_OK)
            throw Error{st.status, st.error};
        if (peer_status != MSVS_OK)
            fail(peer_status, "routed search: rank %zu failed in its front phase (status %d); no rank searched", peer, peer_status);
        for (size_t s = 0; s < W; s++) // the result blocks and the entries are sized by k and nprobe: the same on every rank, or nobody searches
            if (st.h_c[s * RW + W + RH_K] != (uint32_t)st.k || st.h_c[s * RW + W + RH_NP] != (uint32_t)st.np)
                fail(MSVS_ERR_INVALID_ARGUMENT, "routed search: rank %zu was called with k = %u, nprobe = %u, this rank with %d, %zu", s,
                     st.h_c[s * RW + W + RH_K], st.h_c[s * RW + W + RH_NP], st.k, st.np);
        pp.peers_filtered = any_filtered;
        std::shared_ptr<msvs_index::Global> g;
        {
            std::lock_guard<std::mutex> lk(ix->meta_mu);
            g = ix->global;
        }
        const bool stale = need || !g || g->instances != inst; // (`need` and the instances are the same words on every rank; a rank whose
                                                               // Global is missing or older said so itself through `need` / its instance)
        const bool unsafe = any_filtered && any_pruned;
        if (!stale && !unsafe)
        {
            senders_pruned = any_pruned;
            for (size_t s = 0; s < W; s++)
                for (size_t t = 0; t < W; t++)
                    h_c[s * W + t] = st.h_c[s * RW + t];
            break;
        }
        if (attempt >= 2)
            fail(MSVS_ERR_DEVICE, "routed search: the ranks did not agree on the index state after two refreshes");
        if (stale)
            route_global(ix, comm, inst, xs);
        routed_front(pp, st, comm, !unsafe, cs, xs);
    }
    const size_t nq = st.k ? st.nq : 0, np = st.np, d = ix->dim, k = (size_t)st.k;
    if (!k)
        return;
    // ---- the tables of the step
    const size_t ent = 4 * (1 + 2 * np + d);          // bytes of one routed entry
    auto blk = [&](size_t c) { return round_up(c * k * 12, (size_t)16); }; // a result block {ids | dis} of c entries
    std::vector<size_t> mat1(W * W), mat2(W * W);
    for (size_t s = 0; s < W; s++)
        for (size_t t = 0; t < W; t++)
        {
            mat1[s * W + t] = (size_t)h_c[s * W + t] * ent;
            mat2[t * W + s] = blk(h_c[s * W + t]); // the results travel the other way
        }
    RouteTab to{}, from{}, res_of{}, home{};
    size_t sbytes = 0, rbytes = 0, bbytes = 0, res_bytes = 0, n_out = 0, n_in = 0, cmax = 0;
    for (size_t t = 0; t < W; t++)
    {
        to.region[t] = sbytes;
        to.cnt[t] = h_c[r * W + t];
        to.first[t] = (uint32_t)n_out;
        home.region[t] = bbytes;
        home.cnt[t] = h_c[r * W + t];
        home.first[t] = (uint32_t)n_out;
        sbytes += mat1[r * W + t];
        bbytes += blk(h_c[r * W + t]);
        n_out += h_c[r * W + t];
        cmax = std::max<size_t>(cmax, h_c[r * W + t]);
        from.region[t] = rbytes;
        from.cnt[t] = h_c[t * W + r];
        from.first[t] = (uint32_t)n_in;
        res_of.region[t] = res_bytes;
        res_of.cnt[t] = h_c[t * W + r];
        res_of.first[t] = (uint32_t)n_in;
        rbytes += mat1[t * W + r];
        res_bytes += blk(h_c[t * W + r]);
        n_in += h_c[t * W + r];
    }
    if (st.routed_pairs)
        *st.routed_pairs = n_in;
    // ---- 3. pack and exchange
    Scratch & sx = pp.back;
    reserve_two(sx, sbytes + rbytes + comm->exchange_tmp_bytes(mat1) + n_out * 4 + n_in * (np * 8 + d * 4 + k * 12) + res_bytes + bbytes
                        + comm->exchange_tmp_bytes(mat2) + W * nq * k * 12 + 65536,
                cs, xs);
    unsigned char * send = sx.take<unsigned char>(std::max<size_t>(sbytes, 16));
    unsigned char * recv = sx.take<unsigned char>(std::max<size_t>(rbytes, 16));
    unsigned char * tmp1 = sx.take<unsigned char>(std::max<size_t>(comm->exchange_tmp_bytes(mat1), 16));
    uint32_t * sent_q = sx.take<uint32_t>(std::max<size_t>(n_out, 1));
    if (nq)
    {
        RoutePack pk{};
        pk.Q = st.d_queries;
        pk.probes = st.alive_probes;
        pk.words = st.words;
        pk.mask = st.mask;
        pk.nq = (uint32_t)nq;
        pk.np = (uint32_t)np;
        pk.d = (uint32_t)d;
        pk.W = (uint32_t)W;
        pk.cursor = st.cursor;
        pk.send = send;
        pk.sent_q = sent_q;
        pk.to = to;
        hipLaunchKernelGGL(route_pack_kernel, dim3((unsigned)ceil_div(nq, (size_t)ROUTE_PACK_Q)), dim3(256), 0, cs, pk);
        MSVS_HIP(hipGetLastError());
    }
    hand_over(pp, 0, cs, xs);
    {
        ProfileScope prof("shard_exchange", xs);
        comm->exchange(send, recv, mat1, tmp1, xs);
    }
    hand_over(pp, 1, xs, cs);
    // ---- 4. what arrived, searched over this rank's lists under this rank's filter
    unsigned char * res = sx.take<unsigned char>(std::max<size_t>(res_bytes, 16));
    if (n_in)
    {
        int32_t * rp = sx.take<int32_t>(n_in * np);
        uint32_t * rw = sx.take<uint32_t>(n_in * np);
        float * rq = sx.take<float>(n_in * d);
        int64_t * r_ids = sx.take<int64_t>(n_in * k);
        float * r_dis = sx.take<float>(n_in * k);
        RouteUnpack un{};
        un.recv = recv;
        un.from = from;
        un.W = (uint32_t)W;
        un.np = (uint32_t)np;
        un.d = (uint32_t)d;
        un.probes = rp;
        un.words = rw;
        un.Q = rq;
        hipLaunchKernelGGL(route_unpack_kernel, dim3((unsigned)n_in), dim3(64), 0, cs, un, (uint32_t)n_in);
        MSVS_HIP(hipGetLastError());
        const auto meta = ix->get_meta();
        size_t eff_bits = st.nbits;
        const uint64_t * eff = effective_filter(*ix, meta.get(), st.d_alive, st.nbits, &eff_bits, cs);
        ProbeWords gw{};
        gw.given = rw;
        gw.given_pruned = senders_pruned;
        index_search_device(*ix, rq, n_in, (uint32_t)k, np, eff, eff_bits, r_ids, r_dis, cs, rp, nullptr, nullptr, gw);
        hipLaunchKernelGGL(route_result_kernel, dim3((unsigned)ceil_div(n_in * k, (size_t)256)), dim3(256), 0, cs, r_ids, r_dis, (uint32_t)n_in,
                           (uint32_t)k, (uint32_t)W, res_of, res);
        MSVS_HIP(hipGetLastError());
    }
    // ---- 5. results back to their home ranks, merged there
    unsigned char * back = sx.take<unsigned char>(std::max<size_t>(bbytes, 16));
    unsigned char * tmp2 = sx.take<unsigned char>(std::max<size_t>(comm->exchange_tmp_bytes(mat2), 16));
    hand_over(pp, 0, cs, xs);
    {
        ProfileScope prof("shard_exchange", xs);
        comm->exchange(res, back, mat2, tmp2, xs);
    }
    hand_over(pp, 1, xs, cs);
    if (!nq)
        return;
    int64_t * m_ids = sx.take<int64_t>(W * nq * k);
    float * m_dis = sx.take<float>(W * nq * k);
    // -1 / 0: "no hit" in the parts a query did not visit -- one fill kernel (two hipMemsetAsync cost ~7.6 us each between kernels,
    // tools/micro/copy_gap.hip)
    hipLaunchKernelGGL(route_fill_kernel, dim3((unsigned)std::min<size_t>(256, ceil_div(W * nq * k, (size_t)256))), dim3(256), 0, cs, m_ids, m_dis,
                       W * nq * k);
    MSVS_HIP(hipGetLastError());
    if (cmax)
    {
        hipLaunchKernelGGL(route_scatter_kernel, dim3((unsigned)ceil_div(cmax * k, (size_t)256), (unsigned)W), dim3(256), 0, cs, back, home, sent_q,
                           (uint32_t)nq, (uint32_t)k, m_ids, m_dis);
        MSVS_HIP(hipGetLastError());
    }
    const int order = ix->metric == MSVS_METRIC_IP ? MSVS_METRIC_IP : MSVS_METRIC_L2; // (cosine distances leave the search as 1 - ip)
    merge_topk_device(m_ids, nq * k, m_dis, nq * k, W, nq, k, order, st.d_ids, st.d_dis, cs);
    apply_row_ids_map(ix->get_meta().get(), st.d_ids, nq * k, cs);
}

void routed_fill(RoutedStep & st, const msvs_index_t * ix, const float * d_queries, size_t nq, int k, int nprobe, const uint64_t * d_alive,
                 size_t nbits, int64_t * d_ids, float * d_dis, uint64_t * routed_pairs)
{
    st.ix = ix;
    st.d_queries = d_queries;
    st.nq = nq;
    st.k = k;
    st.nprobe = nprobe;
    st.d_alive = d_alive;
    st.nbits = nbits;
    st.d_ids = d_ids;
    st.d_dis = d_dis;
    st.routed_pairs = routed_pairs;
    st.cursor = nullptr;
}

void routed_args_check(const msvs_index_t * ix, const msvs_comm_t * comm)
{
    // (what cannot be published: without a communicator there is nobody to tell)
    if (!ix || !comm)
        fail(MSVS_ERR_INVALID_ARGUMENT, "null index / communicator");
    if (comm->nranks > (int)ROUTE_MAX_RANKS)
        fail(MSVS_ERR_INVALID_ARGUMENT, "the routed search serves up to %u ranks", ROUTE_MAX_RANKS);
}
}

extern "C" int msvs_shard_search_routed_filtered_device(const msvs_index_t * ix, const msvs_comm_t * comm, const float * d_queries, size_t nq,
                                                        int k, int nprobe, const uint64_t * d_alive_bits, size_t nbits, int64_t * d_ids,
                                                        float * d_dis, void * hip_stream, uint64_t * routed_pairs)
{
    return guarded([&] {
        routed_args_check(ix, comm);
        RoutePipe & pp = *route_pipe_of(comm, true);
        std::lock_guard<std::mutex> lk(pp.mu);
        if (pp.step[0].pending || pp.step[1].pending)
            fail(MSVS_ERR_INVALID_ARGUMENT, "a pipelined routed step is still in flight on this communicator: msvs_shard_search_drain first");
        hipStream_t cs = as_stream(hip_stream);
        RoutedStep & st = pp.step[0];
        routed_fill(st, ix, d_queries, nq, k, nprobe, d_alive_bits, nbits, d_ids, d_dis, routed_pairs);
        routed_front(pp, st, comm, true, cs, cs);
        routed_back(pp, st, comm, cs, cs);
    });
}

extern "C" int msvs_shard_search_routed_device(const msvs_index_t * ix, const msvs_comm_t * comm, const float * d_queries, size_t nq, int k,
                                               int nprobe, int64_t * d_ids, float * d_dis, void * hip_stream, uint64_t * routed_pairs)
{
    return msvs_shard_search_routed_filtered_device(ix, comm, d_queries, nq, k, nprobe, nullptr, 0, d_ids, d_dis, hip_stream, routed_pairs);
}

extern "C" int msvs_shard_search_routed_device_async(const msvs_index_t * ix, const msvs_comm_t * comm, const float * d_queries, size_t nq, int k,
                                                     int nprobe, const uint64_t * d_alive_bits, size_t nbits, int64_t * d_ids, float * d_dis,
                                                     void * hip_stream, uint64_t * routed_pairs, void ** prev_done_event)
{
    return guarded([&] {
        routed_args_check(ix, comm);
        RoutePipe & pp = *route_pipe_of(comm, true);
        std::lock_guard<std::mutex> lk(pp.mu);
        const int p = (int)(pp.calls++ & 1);
        RoutedStep & st = pp.step[p];
        RoutedStep & prev = pp.step[1 - p];
        if (st.pending)
            fail(MSVS_ERR_DEVICE, "internal: routed slot still pending");
        if (prev_done_event)
            *prev_done_event = nullptr;
        hipStream_t const caller = as_stream(hip_stream);
        // route_streams = 1 (the default): everything in the caller's stream's order -- FRONT(i), then BACK(i - 1): no internal stream, no
        // event while the caller stays on one stream (an event record between two kernels is a barrier packet: ~8 us of idle device; the
        // two-stream form had five hand-overs per step and -- FRONT(i) being enqueued ahead of BACK(i - 1) -- nothing to overlap the
        // exchanges with).  What the pipelined form buys is that the host never waits for the count matrix.
        const bool own = options().route_streams == 1;
        hipStream_t const cs = own ? caller : pp.compute, xs = own ? caller : pp.xchg;
        if (own)
        {
            if (pp.has_last && pp.last != caller) // the step in flight was enqueued elsewhere: this stream continues behind it
            {
                MSVS_HIP(hipEventRecord(pp.hand[1], pp.last));
                MSVS_HIP(hipStreamWaitEvent(caller, pp.hand[1], 0));
            }
            pp.last = caller;
            pp.has_last = true;
        }
        else
        {
            if (pp.has_last) // (the option changed while a step was in flight on a caller's stream: the internal stream continues behind it)
            {
                MSVS_HIP(hipEventRecord(pp.hand[1], pp.last));
                MSVS_HIP(hipStreamWaitEvent(pp.compute, pp.hand[1], 0));
                pp.has_last = false;
            }
            // inputs: whatever the caller's stream has enqueued so far
            MSVS_HIP(hipEventRecord(st.in_ev, caller));
            MSVS_HIP(hipStreamWaitEvent(pp.compute, st.in_ev, 0));
        }
        routed_fill(st, ix, d_queries, nq, k, nprobe, d_alive_bits, nbits, d_ids, d_dis, routed_pairs);
        routed_front(pp, st, comm, true, cs, xs);
        st.pending = true;
        if (prev.pending)
        {
            prev.pending = false; // (whatever happens below, the step is over)
            routed_back(pp, prev, comm, cs, xs);
            if (!own || prev_done_event)
            {
                MSVS_HIP(hipEventRecord(prev.done_ev, cs));
                if (prev_done_event)
                    *prev_done_event = prev.done_ev;
            }
        }
    });
}

/// `hip_stream` waits for every batch still in flight on the communicator.  With a pipelined routed step pending this is COLLECTIVE:
/// its back phase (two exchanges) runs here.
extern "C" int msvs_shard_search_drain(const msvs_comm_t * comm, void * hip_stream)
{
    return guarded([&] {
        if (!comm)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null communicator");
        if (ShardPipe * pp = pipe_of(comm, false))
        {
            std::lock_guard<std::mutex> lk(pp->mu);
            for (int p = 0; p < 2; p++)
                if (pp->used[p])
                    MSVS_HIP(hipStreamWaitEvent(as_stream(hip_stream), pp->done_ev[p], 0));
        }
        RoutePipe * rpp = route_pipe_of(comm, false);
        if (!rpp)
            return;
        RoutePipe & rp = *rpp;
        std::lock_guard<std::mutex> lk(rp.mu);
        // (route_streams = 1: the steps live on the stream of the call that enqueued them -- rp.last)
        const bool own = rp.has_last;
        hipStream_t const cs = own ? rp.last : rp.compute, xs = own ? rp.last : rp.xchg;
        for (int i = 0; i < 2; i++)
        {
            RoutedStep & st = rp.step[(rp.calls + i) & 1]; // oldest first
            if (!st.pending)
                continue;
            st.pending = false;
            routed_back(rp, st, comm, cs, xs);
            MSVS_HIP(hipEventRecord(st.done_ev, cs));
        }
        if (rp.calls && cs != as_stream(hip_stream))
        {
            MSVS_HIP(hipEventRecord(rp.hand[1], cs));
            MSVS_HIP(hipStreamWaitEvent(as_stream(hip_stream), rp.hand[1], 0));
        }
    });
}

/// msvs_comm_free: the pipelined forms' state of a communicator goes with it (device synchronised first: nothing of it may be in flight).
static void forget_pipes(const msvs_comm_t * c)
{
    bool any = false;
    {
        std::lock_guard<std::mutex> lk(g_pipes_mu);
        any |= g_pipes.count(c) != 0;
    }
    {
        std::lock_guard<std::mutex> lk(g_route_pipes_mu);
        any |= g_route_pipes.count(c) != 0;
    }
    if (!any)
        return;
    (void)hipDeviceSynchronize();
    {
        std::lock_guard<std::mutex> lk(g_pipes_mu);
        auto it = g_pipes.find(c);
        if (it != g_pipes.end())
        {
            ShardPipe & p = *it->second;
            (void)hipStreamDestroy(p.compute);
            (void)hipStreamDestroy(p.xchg);
            for (int i = 0; i < 2; i++)
                for (hipEvent_t e : {p.in_ev[i], p.ab_ev[i], p.x1_ev[i], p.b_ev[i], p.done_ev[i]})
                    (void)hipEventDestroy(e);
            g_pipes.erase(it);
        }
    }
    std::lock_guard<std::mutex> lk(g_route_pipes_mu);
    auto it = g_route_pipes.find(c);
    if (it != g_route_pipes.end())
    {
        RoutePipe & p = *it->second;
        (void)hipStreamDestroy(p.compute);
        (void)hipStreamDestroy(p.xchg);
        for (hipEvent_t e : {p.hand[0], p.hand[1]})
            (void)hipEventDestroy(e);
        for (RoutedStep & st : p.step)
        {
            (void)hipFree(st.cmat);
            (void)hipHostFree(st.h_c);
            for (hipEvent_t e : {st.in_ev, st.a_ev, st.done_ev})
                (void)hipEventDestroy(e);
        }
        g_route_pipes.erase(it);
    }
}
