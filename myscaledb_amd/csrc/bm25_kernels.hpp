// bm25_kernels.hpp -- BM25 posting-list scorer (seam B), doc-partitioned so that scores accumulate in
// query-term order without atomics (bit-identical to tantivy's f32 arithmetic, see oracle_bm25_search):
// block b owns documents [b*DOCS, (b+1)*DOCS); for each query term in order it streams the slice of that term's
// (doc-sorted) posting list falling in its range -- coalesced 4-byte loads of doc ids and tfs -- and adds
//   weight_t * tf / (tf + norm_cache[fieldnorm_id[doc]])
// into an LDS accumulator (doc ids are unique inside one posting list, so no two lanes touch the same slot
// between barriers).  The block's hit documents then go through the wavefront top-k of scan_kernels.hpp.
#pragma once

#include "scan_kernels.hpp"

#pragma clang fp contract(off)

namespace msvs
{

constexpr uint32_t BM25_DOCS = 8192; // documents per block: 32 KiB of f32 scores + 8 KiB of hit flags in LDS
constexpr uint32_t BM25_MAX_TERMS = 64;

struct Bm25Params
{
    const int64_t * post_off;
    const uint32_t * doc_ids;
    const uint32_t * tfs;
    const uint8_t * fieldnorm_ids;
    const uint64_t * alive;
    uint64_t * partial; // [n_blocks][k]
    uint32_t num_docs;
    uint32_t nbits;
    uint32_t k;
    uint32_t n_terms;
    uint32_t qterms[BM25_MAX_TERMS];
    float weight[BM25_MAX_TERMS]; // idf * (1 + K1), computed on the host with libm logf like tantivy
    float norm_cache[256];        // K1 * (1 - B + B * fieldnorm / avg_fieldnorm)
};

template <int R>
__global__ __launch_bounds__(BLOCK) void bm25_score_kernel(const Bm25Params a)
{
    __shared__ float score[BM25_DOCS];
    __shared__ uint8_t hit[BM25_DOCS];
    __shared__ float cache[256];
    __shared__ int64_t range[BM25_MAX_TERMS][2];
    uint64_t * lds_merge = reinterpret_cast<uint64_t *>(msvs_smem);

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t base = blockIdx.x * BM25_DOCS;
    const uint32_t end = base + BM25_DOCS < a.num_docs ? base + BM25_DOCS : a.num_docs;
    for (uint32_t i = tid; i < BM25_DOCS; i += BLOCK)
    {
        score[i] = 0.f;
        hit[i] = 0;
    }
    cache[tid] = a.norm_cache[tid];
    if (tid < 2 * a.n_terms)
    {
        // all the posting-range searches of the block run side by side (one dependent-load chain of ~log2(df)
        // steps in total instead of one per term): first posting of term t with doc >= base (side 0) / end (side 1)
        const uint32_t t = tid >> 1, side = tid & 1;
        const uint32_t target = side == 0 ? base : end;
        int64_t lo = a.post_off[a.qterms[t]], hi = a.post_off[a.qterms[t] + 1];
        while (lo < hi)
        {
            int64_t mid = (lo + hi) >> 1;
            if (a.doc_ids[mid] < target)
                lo = mid + 1;
            else
                hi = mid;
        }
        range[t][side] = lo;
    }
    for (uint32_t t = 0; t < a.n_terms; t++)
    {
        __syncthreads(); // orders term t after term t-1 (and after the range searches): f32 sums in query-term order
        const float w = a.weight[t];
        for (int64_t p = range[t][0] + tid; p < range[t][1]; p += BLOCK)
        {
            const uint32_t doc = a.doc_ids[p];
            const float tf = (float)a.tfs[p];
            const float s = __fmul_rn(w, __fdiv_rn(tf, __fadd_rn(tf, cache[a.fieldnorm_ids[doc]])));
            score[doc - base] = __fadd_rn(score[doc - base], s);
            hit[doc - base] = 1;
        }
    }
    __syncthreads();
    WaveTopK<R> top;
    top.init();
    for (uint32_t i0 = wave * 64; i0 < BM25_DOCS; i0 += BLOCK)
    {
        const uint32_t i = i0 + lane, doc = base + i;
        bool ok = doc < end && hit[i];
        if (ok && a.alive)
            ok = doc < a.nbits && ((a.alive[doc >> 6] >> (doc & 63)) & 1);
        top.offer(ok ? make_key<M_IP>(score[i], doc) : KEY_NONE, a.k, lane);
    }
    top.store(lds_merge + wave * a.k, a.k, lane);
    __syncthreads();
    uint64_t * merged = lds_merge + 4 * a.k;
    block_rank_merge(lds_merge, a.k, merged, a.k, tid);
    for (uint32_t i = tid; i < a.k; i += BLOCK)
        a.partial[(size_t)blockIdx.x * a.k + i] = merged[i];
}

}
