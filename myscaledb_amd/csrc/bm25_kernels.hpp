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
//      y, y + Y, ...: for each query term IN ORDER it streams its slice of the posting list (the first posting of up to
//      four terms per thread is in flight at once), adds weight * tf / (tf + cache[field][fieldnorm]) into an LDS score
//      array, ORs the term's token group into an LDS mask, and records first-touched documents in an LDS list, which is
//      also what gets cleared afterwards -- the score array is zeroed once per block, not once per query.
//      What leaves the block depends on the mode:
//        TOPK : the block's kk best (per-block lists, merged afterwards).  Used (a) over every 16th block with a short
//               list (the SAMPLE), (b) for small corpora, (c) as the exact fallback.
//        EMIT : every touched document scoring >= the query's cut is appended to the query's candidate list.  The cut
//               is the m-th best SAMPLE score, so about 16 m documents of the corpus pass it; when at least k did
//               (and the list did not overflow) the top-k of the candidates IS the top-k of the corpus -- every
//               document scoring at least the k-th best was emitted.  Otherwise the query goes to the fallback.  A
//               per-block top-100 of ~250 touched documents, its 800-byte list and the merges over 1221 of them cost
//               ten times the scoring itself (profiles/r02_bm25.txt).
//   3. bm25_select_kernel : top-k of a query's candidates; queries whose candidates do not prove the top-k are queued
//      for the fallback (TOPK over all blocks + bm25_fb_merge_kernel), launched unconditionally and empty when the
//      queue is.
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

constexpr uint32_t BM25_SAMPLE_STEP = 16; // every 16th document block is a sample block
constexpr uint32_t BM25_CAND_CAP = 2048;  // candidate slots per query (EMIT)

struct Bm25Params
{
    const int64_t * post_off;
    const uint32_t * doc_ids;
    const uint32_t * tfs;
    const uint8_t * fieldnorm_ids; // [num_fields][num_docs]
    const uint8_t * term_field;    // nullable: field of term t (else 0)
    const uint64_t * alive;        // nullable
    uint32_t nbits;
    uint32_t num_docs, num_fields, n_blocks;
    uint64_t last_posting; // num_postings - 1 (0 for an empty index): clamp of the speculative loads
    uint32_t nq, n_flat; // n_flat >= 1 (a dummy term when the batch has none)
    int operator_or;
    // the batch: query q owns flat terms [qoff[q], qoff[q + 1])
    const uint32_t * qoff;
    const uint32_t * qterms;
    const uint8_t * qgroup;   // token group of flat term j (0 .. 15)
    const uint8_t * qfield;   // field of flat term j
    const uint16_t * qfull;   // [nq] mask of all groups of query q
    const float * weight;     // [flat terms] idf * (1 + k1), computed on the host with libm logf like tantivy
    const float * norm_cache; // [num_fields][256]: k1 * (1 - b + b * fieldnorm / avg_fieldnorm)
    const int64_t * bounds;   // [flat terms][n_blocks + 1]: first posting of the term at or past each block boundary
    const int64_t * bounds_hi; // the same array shifted by one block, stored separately: one 16-byte load of the pair
                               // comes back in a register tuple that gets split with a copy -- and a copy waits for its load
    // TOPK: blockIdx.x-th list of slot s (= query, or position in the fallback queue) -> partial[(s * n_pad + x) * kk ..]
    uint64_t * partial;
    uint32_t kk, n_pad, bstep;   // document block = blockIdx.x * bstep
    const uint32_t * qsel;       // nullable: the slots are the first *nsel entries of this queue
    const uint32_t * nsel;
    // EMIT
    const uint64_t * cut_keys;   // [nq][cut_m] ascending keys of the sample's best; the cut is entry cut_m - 1
    uint32_t cut_m;
    uint64_t * cand;             // [nq][BM25_CAND_CAP], the first cand_cap slots in use
    uint32_t cand_cap;
    uint32_t * ccnt;             // [nq], zeroed by the caller
};

static __global__ void bm25_bounds_kernel(const Bm25Params a, int64_t * bounds, int64_t * bounds_hi, uint32_t n_flat)
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
    bounds[i] = lo;
    if (b > 0)
        bounds_hi[i - 1] = lo;
}

/// Workgroup barrier for LDS traffic only.  __syncthreads() also drains the wave's global loads (s_waitcnt vmcnt(0)),
/// which would serialise the software pipeline below at its first barrier: the loads issued for the NEXT queries must
/// stay in flight across the barriers of the current one.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

enum
{
    BM25_TOPK = 0,
    BM25_EMIT = 1
};

/// grid (document blocks, Y); dynamic LDS (TOPK only): 5 * kk * 8 bytes for the block merge.
///
/// The walk over the queries is a software pipeline: a (block, query) step moves a few hundred bytes, so what a step
/// costs is its chain of dependent loads (query -> slices -> postings -> fieldnorms), and the chain of query i + 1 ... i + 3
/// is in flight while query i is being scored:
///   step i issues  the term range of query i + 3            (registers),
///                  the posting-list slices of query i + 2   (registers of threads 0 .. 63, stored to LDS at the end),
///                  the first posting per thread of the first four terms of query i + 1 (slices already in LDS),
///                  ... and, after scoring, their fieldnorm bytes,
///   and scores query i from registers + LDS only (longer slices / more terms fall back to plain loads).
template <int MODE, int R>
__global__ __launch_bounds__(BLOCK) void bm25_score_kernel(const Bm25Params a)
{
    __shared__ float score[BM25_DOCS];
    __shared__ uint16_t mask[BM25_DOCS];
    __shared__ uint16_t touched[BM25_DOCS];
    __shared__ float cache[BM25_MAX_FIELDS * 256];
    __shared__ int64_t sl_p0[2][BM25_MAX_TERMS], sl_p1[2][BM25_MAX_TERMS];
    __shared__ float sl_w[2][BM25_MAX_TERMS];
    __shared__ uint16_t sl_bit[2][BM25_MAX_TERMS];
    __shared__ uint8_t sl_field[2][BM25_MAX_TERMS];
    __shared__ uint32_t ntouch2[2]; // alternates between consecutive queries (reset one barrier away from its last read)
    uint64_t * lds_merge = reinterpret_cast<uint64_t *>(msvs_smem);

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t blk = blockIdx.x * a.bstep;
    const uint32_t base = blk * BM25_DOCS;
    const uint32_t nb1 = a.n_blocks + 1;
    const uint32_t nslots = a.qsel ? *a.nsel : a.nq;
    if (blockIdx.y >= nslots)
        return;
    for (uint32_t i = tid; i < BM25_DOCS; i += BLOCK)
    {
        score[i] = 0.f;
        mask[i] = 0;
    }
    for (uint32_t i = tid; i < a.num_fields * 256; i += BLOCK)
        cache[i] = a.norm_cache[i];
    if (tid < 2)
        ntouch2[tid] = 0;

    // What the prefetches return stays RAW in registers until the step that needs it: any arithmetic on a loaded value
    // makes the compiler wait for the load where the arithmetic stands, i.e. right after the issue.
    struct QInfo
    {
        uint32_t q, j0, j1; // term range [j0, j1)
        uint16_t full;
        uint64_t ck; // EMIT: the cut as a key
        __device__ uint32_t nt() const { return j1 - j0; }
    };
    struct Slice
    {
        int64_t p0, p1;
        float w;
        uint8_t group, field;
    };
    struct First // the first posting of this thread in each of the first four terms
    {
        uint32_t doc[4], tf[4], fid[4];
        bool has[4];
    };
    auto load_qinfo = [&](uint32_t s) {
        QInfo r{0, 0, 0, 0, KEY_NONE};
        if (s < nslots) // uniform
        {
            r.q = a.qsel ? a.qsel[s] : s; // the fallback queue costs a dependent load here; it is rare
            r.j0 = a.qoff[r.q];
            r.j1 = a.qoff[r.q + 1];
            r.full = a.qfull[r.q];
            if (MODE == BM25_EMIT)
                r.ck = a.cut_keys[(size_t)r.q * a.cut_m + a.cut_m - 1];
        }
        return r;
    };
    auto load_slice = [&](const QInfo & qi) {
        // wavefront 0, lane = term; clamped addresses instead of a per-lane branch (a branch would merge the loaded
        // registers with defaults through copies, and a copy waits for its load)
        Slice r;
        if (wave == 0)
        {
            const uint32_t j = qi.j0 + tid < a.n_flat ? qi.j0 + tid : a.n_flat - 1;
            r.p0 = a.bounds[(size_t)j * nb1 + blk];
            r.p1 = a.bounds_hi[(size_t)j * nb1 + blk];
            r.w = a.weight[j];
            r.group = a.qgroup[j];
            r.field = a.qfield[j];
        }
        return r;
    };
    auto store_slice = [&](const Slice & r, const QInfo & qi, uint32_t buf) {
        if (wave == 0)
        {
            const bool on = tid < qi.nt(); // slots past the query's terms hold empty slices
            sl_p0[buf][tid] = on ? r.p0 : 0;
            sl_p1[buf][tid] = on ? r.p1 : 0;
            sl_w[buf][tid] = r.w;
            sl_bit[buf][tid] = (uint16_t)(1u << r.group);
            sl_field[buf][tid] = on ? r.field : (uint8_t)0;
        }
    };
    auto load_first = [&](uint32_t buf, uint32_t nt) {
        First r;
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            const int64_t p = sl_p0[buf][u] + tid; // slots past nt hold empty slices (p0 = p1 = 0)
            r.has[u] = (uint32_t)u < nt && p < sl_p1[buf][u];
            const uint64_t pc = (uint64_t)p < a.last_posting ? (uint64_t)p : a.last_posting;
            r.doc[u] = a.doc_ids[pc];
            r.tf[u] = a.tfs[pc];
            r.fid[u] = 0;
        }
        return r;
    };
    auto load_fids = [&](First & r, uint32_t buf) {
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            const uint32_t dc = r.doc[u] < a.num_docs ? r.doc[u] : 0;
            r.fid[u] = a.fieldnorm_ids[(size_t)sl_field[buf][u] * a.num_docs + dc];
        }
    };

    // prologue: fill the pipeline
    const uint32_t stride = gridDim.y;
    QInfo q0 = load_qinfo(blockIdx.y), q1 = load_qinfo(blockIdx.y + stride), q2 = load_qinfo(blockIdx.y + 2 * stride);
    uint32_t b0 = 0, b1 = 1; // LDS holds the slices of q0 and q1; those of q2 wait in registers until q0 is done
    store_slice(load_slice(q0), q0, b0);
    store_slice(load_slice(q1), q1, b1);
    lds_barrier();
    First fa = load_first(b0, q0.nt()), fb;
    load_fids(fa, b0);
    uint32_t par = 0;

    // one posting per lane into the block's score / mask / touched state (doc ids are unique inside one posting list: no
    // two lanes touch the same slot between barriers); first-touched documents join the list with one LDS atomic per
    // wavefront
    auto add_posting = [&](bool live, uint32_t d, uint32_t f, uint32_t id, float w, const float * fc, uint16_t bit, uint32_t & ntouch) {
        bool fresh = false;
        uint32_t loc = 0;
        if (live)
        {
            loc = d - base;
            const float tff = (float)f;
            const float sc = __fmul_rn(w, __fdiv_rn(tff, __fadd_rn(tff, fc[id])));
            score[loc] = __fadd_rn(score[loc], sc);
            const uint16_t old = mask[loc];
            mask[loc] = old | bit;
            fresh = old == 0;
        }
        const uint64_t fm = __ballot(fresh);
        if (fm)
        {
            uint32_t at = 0;
            if (lane == (uint32_t)__builtin_ctzll(fm))
                at = atomicAdd(&ntouch, (uint32_t)__builtin_popcountll(fm));
            at = __shfl(at, __builtin_ctzll(fm));
            if (fresh)
                touched[at + __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u))] = (uint16_t)loc;
        }
    };

    // one step: score the query whose first postings are in `cur`, prefetch into `nxt` (the two alternate by NAME in the
    // loop below: a register copy of a value still in flight would wait for it)
    auto step = [&](uint32_t s, First & cur, First & nxt) {
        uint32_t & ntouch = ntouch2[par];
        if (tid == 0)
            ntouch2[par ^ 1] = 0; // everyone passed the barrier that ended the previous query
        // ---- issue the loads of the queries behind this one
        const QInfo q3 = load_qinfo(s + 3 * stride);
        const Slice s2 = load_slice(q2);
        nxt = load_first(b1, q1.nt());
        __builtin_amdgcn_sched_barrier(0); // keep them ahead of the scoring below
        // ---- score query q0
        const uint32_t nt = q0.nt();
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            if ((uint32_t)u >= nt) // uniform
                break;
            const float w = sl_w[b0][u];
            const float * fc = cache + sl_field[b0][u] * 256;
            const uint8_t * fn = a.fieldnorm_ids + (size_t)sl_field[b0][u] * a.num_docs;
            const uint16_t bit = sl_bit[b0][u];
            const int64_t p1 = sl_p1[b0][u];
            int64_t p = sl_p0[b0][u] + tid;
            // the prefetched posting first, then (long slices only) the rest with plain loads: a loop carrying the
            // registers of both would make the compiler drain every load in flight at its top
            add_posting(cur.has[u], cur.doc[u], cur.tf[u], cur.fid[u], w, fc, bit, ntouch);
            for (p += BLOCK; __any(p < p1); p += BLOCK)
            {
                const bool live = p < p1;
                uint32_t d = 0, f = 0, id = 0;
                if (live)
                {
                    d = a.doc_ids[p];
                    f = a.tfs[p];
                    id = fn[d];
                }
                add_posting(live, d, f, id, w, fc, bit, ntouch);
            }
            lds_barrier(); // term t + 1 after term t: f32 sums in query-term order
        }
        for (uint32_t t = 4; t < nt; t++)
        {
            const float w = sl_w[b0][t];
            const float * fc = cache + sl_field[b0][t] * 256;
            const uint8_t * fn = a.fieldnorm_ids + (size_t)sl_field[b0][t] * a.num_docs;
            const uint16_t bit = sl_bit[b0][t];
            const int64_t p1 = sl_p1[b0][t];
            for (int64_t p = sl_p0[b0][t] + tid; __any(p < p1); p += BLOCK)
            {
                const bool live = p < p1;
                uint32_t d = 0, f = 0, id = 0;
                if (live)
                {
                    d = a.doc_ids[p];
                    f = a.tfs[p];
                    id = fn[d];
                }
                add_posting(live, d, f, id, w, fc, bit, ntouch);
            }
            lds_barrier();
        }
        load_fids(nxt, b1); // the postings of q1 have arrived by now
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t ntc = ntouch;
        const uint16_t full = q0.full;
        if (MODE == BM25_EMIT)
        {
            const float cut = q0.ck == KEY_NONE ? 0.f : key_value<M_IP>(q0.ck); // fewer than m sample hits: all pass
            for (uint32_t i = tid; i < ntc; i += BLOCK)
            {
                const uint32_t loc = touched[i], docid = base + loc;
                const float sc = score[loc];
                bool ok = sc >= cut && (a.operator_or || mask[loc] == full);
                score[loc] = 0.f;
                mask[loc] = 0;
                if (ok && a.alive)
                    ok = docid < a.nbits && ((a.alive[docid >> 6] >> (docid & 63)) & 1);
                if (ok)
                {
                    const uint32_t pos = atomicAdd(&a.ccnt[q0.q], 1u);
                    if (pos < a.cand_cap)
                        a.cand[(size_t)q0.q * BM25_CAND_CAP + pos] = make_key<M_IP>(sc, docid);
                }
            }
        }
        else
        {
            const uint32_t k = a.kk;
            WaveTopK<R> top;
            top.init();
            for (uint32_t i0 = wave * 64; i0 < ntc; i0 += BLOCK)
            {
                const uint32_t i = i0 + lane;
                uint64_t key = KEY_NONE;
                if (i < ntc)
                {
                    const uint32_t loc = touched[i], docid = base + loc;
                    bool ok = a.operator_or || mask[loc] == full;
                    if (ok && a.alive)
                        ok = docid < a.nbits && ((a.alive[docid >> 6] >> (docid & 63)) & 1);
                    if (ok)
                        key = make_key<M_IP>(score[loc], docid);
                }
                top.offer(key, k, lane);
            }
            top.store(lds_merge + wave * k, k, lane);
            lds_barrier();
            for (uint32_t i = tid; i < ntc; i += BLOCK) // clear exactly what this query touched
            {
                const uint32_t loc = touched[i];
                score[loc] = 0.f;
                mask[loc] = 0;
            }
            uint64_t * merged = lds_merge + 4 * k;
            block_rank_merge(lds_merge, k, merged, k, tid); // ends with a barrier
            uint64_t * out = a.partial + ((size_t)s * a.n_pad + blockIdx.x) * k;
            for (uint32_t i = tid; i < k; i += BLOCK)
                out[i] = merged[i];
        }
        // ---- rotate the pipeline (the asm pins the first use of the prefetched registers HERE, not at their issue)
        QInfo q3u = q3;
        Slice s2u = s2;
        asm volatile("" : "+v"(q3u.j0), "+v"(q3u.j1), "+v"(q3u.ck), "+v"(s2u.p0), "+v"(s2u.p1), "+v"(s2u.w));
        store_slice(s2u, q2, b0); // q0 is done: its buffer takes the slices of q2, which is the next q1
        const uint32_t t_ = b0;
        b0 = b1;
        b1 = t_;
        q0 = q1;
        q1 = q2;
        q2 = q3u;
        par ^= 1;
        lds_barrier(); // slices of the new q1 visible; clearing done before the next query's adds
    };
    for (uint32_t s = blockIdx.y;;)
    {
        step(s, fa, fb);
        s += stride;
        if (s >= nslots)
            break;
        step(s, fb, fa);
        s += stride;
        if (s >= nslots)
            break;
    }
}

/// One block per query: top-k of its candidates -> results; a query whose candidates cannot prove its top-k (list
/// overflowed, or a real cut let fewer than k through) joins the fallback queue instead.
template <int R>
__global__ __launch_bounds__(BLOCK) void bm25_select_kernel(const uint64_t * cand, const uint32_t * ccnt, uint32_t cand_cap,
                                                            const uint64_t * cut_keys, uint32_t cut_m, uint32_t k, int64_t * out_ids,
                                                            float * out_scores, uint32_t * failq, uint32_t * nfail,
                                                            unsigned long long * stat_fail)
{
    uint64_t * lds = reinterpret_cast<uint64_t *>(msvs_smem);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = blockIdx.x;
    const uint32_t cnt = ccnt[q];
    const bool real_cut = cut_keys[(size_t)q * cut_m + cut_m - 1] != KEY_NONE;
    if (cnt > cand_cap || (real_cut && cnt < k))
    {
        if (tid == 0)
        {
            failq[atomicAdd(nfail, 1u)] = q;
            atomicAdd(stat_fail, 1ull);
        }
        return;
    }
    const uint64_t * src = cand + (size_t)q * BM25_CAND_CAP;
    WaveTopK<R> top;
    top.init();
    for (uint32_t b = 0; b < cnt; b += 4 * BLOCK)
    {
        uint64_t key[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            const uint32_t i = b + u * BLOCK + tid;
            key[u] = i < cnt ? src[i] : KEY_NONE;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            top.offer(key[u], k, lane);
    }
    top.store(lds + wave * k, k, lane);
    __syncthreads();
    uint64_t * merged = lds + 4 * k;
    block_rank_merge(lds, k, merged, k, tid);
    for (uint32_t i = tid; i < k; i += BLOCK)
    {
        const uint64_t key = merged[i];
        out_ids[(size_t)q * k + i] = key == KEY_NONE ? -1 : (int64_t)(uint32_t)key;
        out_scores[(size_t)q * k + i] = key_value<M_IP>(key);
    }
}

/// Fallback merge: slot s of the queue (blocks past *nfail leave at once) -> top-k of its n_lists per-block lists.
template <int R>
__global__ __launch_bounds__(BLOCK) void bm25_fb_merge_kernel(const uint64_t * partial, uint32_t n_lists, uint32_t n_pad, uint32_t k,
                                                              const uint32_t * failq, const uint32_t * nfail, int64_t * out_ids,
                                                              float * out_scores)
{
    uint64_t * lds = reinterpret_cast<uint64_t *>(msvs_smem);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, s = blockIdx.x;
    if (s >= *nfail)
        return;
    const uint32_t q = failq[s];
    const uint64_t * src = partial + (size_t)s * n_pad * k;
    const uint64_t total = (uint64_t)n_lists * k;
    WaveTopK<R> top;
    top.init();
    for (uint64_t b = 0; b < total; b += 4 * BLOCK)
    {
        uint64_t key[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            const uint64_t i = b + u * BLOCK + tid;
            key[u] = i < total ? src[i] : KEY_NONE;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            top.offer(key[u], k, lane);
    }
    top.store(lds + wave * k, k, lane);
    __syncthreads();
    uint64_t * merged = lds + 4 * k;
    block_rank_merge(lds, k, merged, k, tid);
    for (uint32_t i = tid; i < k; i += BLOCK)
    {
        const uint64_t key = merged[i];
        out_ids[(size_t)q * k + i] = key == KEY_NONE ? -1 : (int64_t)(uint32_t)key;
        out_scores[(size_t)q * k + i] = key_value<M_IP>(key);
    }
}

}
