// bm25_kernels.hpp -- BM25 posting-list scorer (seam B), batched.
//
// Replaces the scoring inside TANTIVY::ffi_bm25_search (src/Storages/MergeTree/TantivyIndexStore.cpp:900-954).  The
// arithmetic is tantivy's f32 Bm25Weight (k1 = 1.2, b = 0.75, idf = ln(1 + (N - n + 0.5) / (n + 0.5)), fieldnorm cache of
// 256 entries per field) and scores accumulate in QUERY-TERM ORDER without atomics, so they are bit-identical to the
// oracle's (and to the goldens of 00040 / 00041).
//
// Shape: HBM-bound integer / byte work (sum_t df_t * 8 B + touched fieldnorm bytes per query), far too little
// per query to fill the chip (3 mid-frequency terms over 10M documents = a few MB), so the unit of work is a BATCH:
//   1. bm25_bounds_kernel : one thread per (query term, document-block boundary): first posting of the term with
//      doc >= boundary (binary search).  All ~17-step dependent-load chains of the batch run side by side, once.
//   2. bm25_score_kernel  : block (b, y) owns documents [b D, (b + 1) D), D = 8192, and walks the batch's queries
//      y, y + Y, ...: for each query term IN ORDER it streams its slice of the posting list (coalesced 4-byte loads of
//      doc ids and tfs), adds weight * tf / (tf + cache[field][fieldnorm]) into an LDS score array, ORs the term's token
//      group into an LDS mask, and records first-touched documents in an LDS list; the block's top-k is then taken over
//      the TOUCHED documents only (a few hundred, not 8192 slots) and exactly those slots are zeroed again -- the score
//      array is cleared once per block, not once per query.
//   3. two levels of merge_kernel over the per-block lists.
// AND (operator_or = false): a document qualifies when every token GROUP of the query matched (a token searched in
// several fields is one group: tantivy's conjunction of per-field disjunctions); multi-field: every (field, token) is its
// own term with its own df, fieldnorms and average length.
#pragma once

#include "scan_kernels.hpp"

#pragma clang fp contract(off)

namespace msvs
{

constexpr uint32_t BM25_DOCS = 8192; // documents per block
constexpr uint32_t BM25_MAX_TERMS = 64;
constexpr uint32_t BM25_MAX_GROUPS = 16;
constexpr uint32_t BM25_MAX_FIELDS = 4;

struct Bm25Params
{
    const int64_t * post_off;
    const uint32_t * doc_ids;
    const uint32_t * tfs;
    const uint8_t * fieldnorm_ids; // [num_fields][num_docs]
    const uint8_t * term_field;    // nullable: field of term t (else 0)
    const uint64_t * alive;        // nullable
    uint32_t nbits;
    uint32_t num_docs, num_fields, n_blocks, n_pad;
    uint32_t k, nq;
    int operator_or;
    // the batch: query q owns flat terms [qoff[q], qoff[q + 1])
    const uint32_t * qoff;
    const uint32_t * qterms;
    const uint8_t * qgroup;   // token group of flat term j (0 .. 15)
    const uint16_t * qfull;   // [nq] mask of all groups of query q
    const float * weight;     // [flat terms] idf * (1 + k1), computed on the host with libm logf like tantivy
    const float * norm_cache; // [num_fields][256]: k1 * (1 - b + b * fieldnorm / avg_fieldnorm)
    int64_t * bounds;         // [flat terms][n_blocks + 1]
    uint64_t * partial;       // [nq][n_pad][k]
};

static __global__ void bm25_bounds_kernel(const Bm25Params a, uint32_t n_flat)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nb1 = a.n_blocks + 1;
    if (i >= (size_t)n_flat * nb1)
        return;
    const uint32_t j = (uint32_t)(i / nb1), b = (uint32_t)(i - (size_t)j * nb1);
    const uint32_t term = a.qterms[j];
    const uint64_t target = (uint64_t)b * BM25_DOCS;
    int64_t lo = a.post_off[term], hi = a.post_off[term + 1];
    while (lo < hi)
    {
        const int64_t mid = (lo + hi) >> 1;
        if (a.doc_ids[mid] < target)
            lo = mid + 1;
        else
            hi = mid;
    }
    a.bounds[i] = lo;
}

/// grid (n_blocks, Y); dynamic LDS: 5 * k * 8 bytes for the block merge.
template <int R>
__global__ __launch_bounds__(BLOCK) void bm25_score_kernel(const Bm25Params a)
{
    __shared__ float score[BM25_DOCS];
    __shared__ uint16_t mask[BM25_DOCS];
    __shared__ uint16_t touched[BM25_DOCS];
    __shared__ float cache[BM25_MAX_FIELDS * 256];
    __shared__ uint32_t ntouch;
    uint64_t * lds_merge = reinterpret_cast<uint64_t *>(msvs_smem);

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, k = a.k;
    const uint32_t base = blockIdx.x * BM25_DOCS;
    const uint32_t end = base + BM25_DOCS < a.num_docs ? base + BM25_DOCS : a.num_docs;
    const uint32_t nb1 = a.n_blocks + 1;
    for (uint32_t i = tid; i < BM25_DOCS; i += BLOCK)
    {
        score[i] = 0.f;
        mask[i] = 0;
    }
    for (uint32_t i = tid; i < a.num_fields * 256; i += BLOCK)
        cache[i] = a.norm_cache[i];
    if (tid == 0)
        ntouch = 0;
    __syncthreads();
    for (uint32_t q = blockIdx.y; q < a.nq; q += gridDim.y)
    {
        const uint32_t j0 = a.qoff[q], j1 = a.qoff[q + 1];
        for (uint32_t j = j0; j < j1; j++)
        {
            const float w = a.weight[j];
            const uint32_t field = a.term_field ? a.term_field[a.qterms[j]] : 0u;
            const uint8_t * fn = a.fieldnorm_ids + (size_t)field * a.num_docs;
            const float * fc = cache + field * 256;
            const uint16_t bit = (uint16_t)(1u << a.qgroup[j]);
            const int64_t p0 = a.bounds[(size_t)j * nb1 + blockIdx.x], p1 = a.bounds[(size_t)j * nb1 + blockIdx.x + 1];
            for (int64_t p = p0 + tid; p < p1; p += BLOCK)
            {
                // doc ids are unique inside one posting list: no two lanes touch the same slot between barriers
                const uint32_t doc = a.doc_ids[p], loc = doc - base;
                const float tf = (float)a.tfs[p];
                const float s = __fmul_rn(w, __fdiv_rn(tf, __fadd_rn(tf, fc[fn[doc]])));
                score[loc] = __fadd_rn(score[loc], s);
                const uint16_t old = mask[loc];
                mask[loc] = old | bit;
                if (old == 0)
                    touched[atomicAdd(&ntouch, 1u)] = (uint16_t)loc;
            }
            __syncthreads(); // term j + 1 after term j: f32 sums in query-term order
        }
        const uint32_t nt = ntouch;
        const uint16_t full = a.qfull[q];
        WaveTopK<R> top;
        top.init();
        for (uint32_t i0 = wave * 64; i0 < nt; i0 += BLOCK)
        {
            const uint32_t i = i0 + lane;
            uint64_t key = KEY_NONE;
            if (i < nt)
            {
                const uint32_t loc = touched[i], doc = base + loc;
                bool ok = doc < end && (a.operator_or || mask[loc] == full);
                if (ok && a.alive)
                    ok = doc < a.nbits && ((a.alive[doc >> 6] >> (doc & 63)) & 1);
                if (ok)
                    key = make_key<M_IP>(score[loc], doc);
            }
            top.offer(key, k, lane);
        }
        top.store(lds_merge + wave * k, k, lane);
        __syncthreads();
        // clear exactly what this query touched
        for (uint32_t i = tid; i < nt; i += BLOCK)
        {
            const uint32_t loc = touched[i];
            score[loc] = 0.f;
            mask[loc] = 0;
        }
        uint64_t * merged = lds_merge + 4 * k;
        block_rank_merge(lds_merge, k, merged, k, tid); // ends with a barrier
        uint64_t * out = a.partial + ((size_t)q * a.n_pad + blockIdx.x) * k;
        for (uint32_t i = tid; i < k; i += BLOCK)
            out[i] = merged[i];
        if (tid == 0)
            ntouch = 0;
        __syncthreads();
    }
}

}
