// fusion.hip -- msvs_hybrid_fuse_device (include/msvs.h): the fusion step of a hybrid search for a BATCH of queries on the
// device, straight from the two device searches' output arrays.
//
// Follows RankFusion / RelativeScoreFusion / computeNormalizedScore (src/VectorIndex/Utils/HybridSearchUtils.cpp:164-300) as
// MergeTreeHybridSearchManager::hybridSearch applies them -- the arithmetic and order of msvs_host_hybrid_search_batch
// (host/msvs_host.cpp), which tests hold against the map-based mirror: a label's contributions are applied in list order
// (RRF: 0 + 1/(k + rank_vector), then + 1/(k + rank_text); RSF: the text list ASSIGNS weight * norm, the vector list adds),
// output = descending fused score, ties by ascending label.  A hybrid batch of 64 queries spent 0.6 of its 2.7 ms in that
// host loop behind a stream synchronisation and four device-to-host copies; here the lists never leave the device and only
// the top-k rows are read back.
//
// One workgroup per query, thread i = entry i of (vector list | text list), both <= 256 long.  Every entry looks its label
// up in the other list (LDS, <= 256 compares); an entry is the OWNER of its label when it is the vector entry, or a text
// entry without a vector partner; owners rank themselves against each other by (score desc, label asc) and write themselves
// to their output slot.  The lists hold distinct labels each (row ids of one search).
#include "device_ops.hpp"

#pragma clang fp contract(off)

namespace msvs
{

constexpr uint32_t FUSE_MAX = 256; // entries per list (MSVS_MAX_K)

struct FuseParams
{
    const float * vec_dis;
    const int64_t * vec_ids;
    const float * txt_scores;
    const int64_t * txt_ids;
    uint32_t kv, kt, nq, topk;
    int rsf;
    uint64_t fusion_k;
    float weight;
    int direction;
    float * out_scores;
    int64_t * out_labels;
    uint32_t * n_out;
};

/// computeNormalizedScore: (score - min) / (max - min) with min / max = last / first entry (swapped when descending);
/// all equal: 1.
__device__ __forceinline__ float fuse_norm(const float s, const float first, const float last)
{
    float mn = last, mx = first;
    if (mn == mx)
        return 1.0f;
    if (mn > mx)
    {
        const float t = mn;
        mn = mx;
        mx = t;
    }
    return __fdiv_rn(__fsub_rn(s, mn), __fsub_rn(mx, mn));
}

static __global__ __launch_bounds__(2 * FUSE_MAX) void hybrid_fuse_kernel(const FuseParams p)
{
    __shared__ uint64_t s_label[2 * FUSE_MAX];
    __shared__ float s_value[2 * FUSE_MAX];
    __shared__ float s_score[2 * FUSE_MAX]; // fused score of the owners, NaN-free; non-owners: marked by s_owner = 0
    __shared__ uint8_t s_owner[2 * FUSE_MAX];
    __shared__ uint32_t s_n[3]; // nv, nt, owners
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const int64_t * vi = p.vec_ids + (size_t)q * p.kv, * ti = p.txt_ids + (size_t)q * p.kt;
    const float * vs = p.vec_dis + (size_t)q * p.kv, * ts = p.txt_scores + (size_t)q * p.kt;
    if (tid < 3)
        s_n[tid] = 0;
    __syncthreads();
    // list lengths: the first id < 0 ends a list (the host's `> -1` unpack; ids after it are not read)
    if (tid == 0)
    {
        uint32_t nv = 0, nt = 0;
        while (nv < p.kv && vi[nv] > -1)
            nv++;
        while (nt < p.kt && ti[nt] > -1)
            nt++;
        s_n[0] = nv;
        s_n[1] = nt;
    }
    __syncthreads();
    const uint32_t nv = s_n[0], nt = s_n[1];
    const bool is_vec = tid < FUSE_MAX;
    const uint32_t i = is_vec ? tid : tid - FUSE_MAX;
    const bool have = is_vec ? i < nv : i < nt;
    uint64_t label = 0;
    float value = 0.f;
    if (have)
    {
        label = (uint64_t)(is_vec ? vi[i] : ti[i]);
        if (!p.rsf)
            value = __fdiv_rn(1.0f, (float)(p.fusion_k + (uint64_t)(i + 1)));
        else if (is_vec)
        {
            const float n = fuse_norm(vs[i], vs[0], vs[nv - 1]);
            const float w1 = __fsub_rn(1.0f, p.weight);
            value = p.direction == -1 ? __fmul_rn(n, w1) : __fmul_rn(__fsub_rn(1.0f, n), w1);
        }
        else
            value = __fmul_rn(fuse_norm(ts[i], ts[0], ts[nt - 1]), p.weight);
    }
    s_label[tid] = label;
    s_value[tid] = value;
    __syncthreads();
    // the partner in the other list
    bool owner = have;
    float score = 0.f;
    if (have)
    {
        const uint32_t ob = is_vec ? FUSE_MAX : 0, on = is_vec ? nt : nv;
        int partner = -1;
        for (uint32_t j = 0; j < on; j++)
            if (s_label[ob + j] == label)
            {
                partner = (int)j;
                break;
            }
        if (is_vec)
        {
            // RRF: (0 + vector) + text; RSF: text assigns, vector adds
            if (p.rsf)
                score = partner >= 0 ? __fadd_rn(s_value[ob + partner], value) : __fadd_rn(0.0f, value);
            else
                score = partner >= 0 ? __fadd_rn(__fadd_rn(0.0f, value), s_value[ob + partner]) : __fadd_rn(0.0f, value);
        }
        else
        {
            owner = partner < 0;
            score = p.rsf ? value : __fadd_rn(0.0f, value);
        }
    }
    s_owner[tid] = owner ? 1 : 0;
    s_score[tid] = score;
    __syncthreads();
    if (owner)
    {
        uint32_t rank = 0;
        for (uint32_t j = 0; j < 2 * FUSE_MAX; j++)
        {
            if (!s_owner[j])
                continue;
            const float sj = s_score[j];
            rank += (sj > score || (sj == score && s_label[j] < label)) ? 1u : 0u;
        }
        atomicAdd(&s_n[2], 1u);
        if (rank < p.topk)
        {
            p.out_scores[(size_t)q * p.topk + rank] = score;
            p.out_labels[(size_t)q * p.topk + rank] = (int64_t)label;
        }
    }
    __syncthreads();
    const uint32_t n = s_n[2] < p.topk ? s_n[2] : p.topk;
    if (tid == 0)
        p.n_out[q] = n;
    if (tid >= n && tid < p.topk)
    {
        p.out_scores[(size_t)q * p.topk + tid] = 0.f;
        p.out_labels[(size_t)q * p.topk + tid] = -1;
    }
}

}

using namespace msvs;

extern "C" int msvs_hybrid_fuse_device(int fusion_type, const float * d_vec_dis, const int64_t * d_vec_ids, size_t kv,
                                       const float * d_txt_scores, const int64_t * d_txt_ids, size_t kt, size_t nq, uint64_t fusion_k,
                                       float fusion_weight, int vector_scan_direction, size_t topk, float * d_out_scores,
                                       int64_t * d_out_labels, uint32_t * d_n_out, void * hip_stream)
{
    return guarded([&] {
        if (nq == 0)
            return;
        if (!d_vec_dis || !d_vec_ids || !d_txt_scores || !d_txt_ids || !d_out_scores || !d_out_labels || !d_n_out || topk == 0)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null buffer or topk = 0");
        if (kv > FUSE_MAX || kt > FUSE_MAX || topk > 2 * FUSE_MAX)
            fail(MSVS_ERR_UNSUPPORTED_K, "the device fusion takes lists of at most %u rows", FUSE_MAX);
        if (fusion_type != 0 && fusion_type != 1)
            fail(MSVS_ERR_INVALID_ARGUMENT, "fusion_type: 0 = RRF, 1 = RSF");
        FuseParams p{};
        p.vec_dis = d_vec_dis;
        p.vec_ids = d_vec_ids;
        p.txt_scores = d_txt_scores;
        p.txt_ids = d_txt_ids;
        p.kv = (uint32_t)kv;
        p.kt = (uint32_t)kt;
        p.nq = (uint32_t)nq;
        p.topk = (uint32_t)topk;
        p.rsf = fusion_type == 1;
        p.fusion_k = fusion_k == 0 ? 60 : fusion_k;
        p.weight = fusion_weight;
        p.direction = vector_scan_direction;
        p.out_scores = d_out_scores;
        p.out_labels = d_out_labels;
        p.n_out = d_n_out;
        hipLaunchKernelGGL(hybrid_fuse_kernel, dim3((unsigned)nq), dim3(2 * FUSE_MAX), 0, reinterpret_cast<hipStream_t>(hip_stream), p);
        MSVS_HIP(hipGetLastError());
    });
}
