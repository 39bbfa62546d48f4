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
    // the first scorer launch behind the batch's tables tells the host that their pinned slot is free again (a word in pinned memory;
    // nullable): the tables are copied by the bounds launch in front of it (bm25_bounds8_kernel), complete when this launch starts
    uint32_t * slot_done = nullptr;
    uint32_t slot_seq = 0;
};

__device__ __forceinline__ void bm25_slot_signal(const Bm25Params & p)
{
    if (p.slot_done && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        __hip_atomic_store(p.slot_done, p.slot_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

/// The fills of the batch ride along (nullable): `zero` = the candidate counters + fail counter, `ones` = the sample's
/// per-query lists (KEY_NONE = all bits set) -- two launches less in front of the sample pass.
static __global__ void bm25_bounds_kernel(const Bm25Params a, int64_t * bounds, int64_t * bounds_hi, uint32_t n_flat,
                                          uint32_t docs_per_block, uint32_t * zero, size_t n_zero, uint64_t * ones, size_t n_ones)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    {
        const size_t gsz = (size_t)gridDim.x * blockDim.x;
        for (size_t j = i; j < n_zero; j += gsz)
            zero[j] = 0;
        for (size_t j = i; j < n_ones; j += gsz)
            ones[j] = KEY_NONE;
    }
    const uint32_t nb1 = a.n_blocks + 1;
    if (i >= (size_t)n_flat * nb1)
        return;
    const uint32_t j = (uint32_t)(i / nb1), b = (uint32_t)(i - (size_t)j * nb1);
    const uint32_t term = a.qterms[j];
    const uint64_t target = (uint64_t)b * docs_per_block;
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
    bm25_slot_signal(a);
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

// ------------------------------------------------------------------------------------------ wave-private streaming scorer
//
// bm25_score_kernel above moves ~2 KB per (block, query) step through 4 wavefronts and 5 workgroup barriers: 12.7 us per
// query at batch 64 over 10M documents whatever is prefetched.  Here the unit is ONE WAVEFRONT and there are no barriers:
//   * a work item = (chunk of `spi` consecutive 2048-document sub-ranges, query); persistent wavefronts take items
//     chunk-major (the 4 wavefronts of a block work on 4 queries of the same chunk: shared fieldnorm lines);
//   * the wavefront owns f32 scores, u16 token-group masks and a touched list for 2048 documents in LDS (16 KB) plus
//     the sub-range's fieldnorm bytes (2 KB per text column, loaded coalesced instead of gathered per posting);
//   * per sub-range and term IN ORDER the slice of the posting list (bounds precomputed at sub-range granularity) is
//     consumed 64 postings at a time: LDS ops of one wavefront execute in order, so term t + 1 sees term t's sums
//     without any barrier; first-touched documents join the list through ballot + mbcnt, no atomics;
//   * what the NEXT sub-range needs is in flight while this one is scored: its slice bounds are loaded two sub-ranges
//     ahead (lane t = term t), its fieldnorm slice and the first 64 postings of its first four terms one ahead;
//   * TOPK keeps one WaveTopK across the whole item (one sorted list per item), EMIT appends what passes the cut.
// Measured (profiles/r02_bm25.txt): 10.0 us per query at batch 64 (block scorer 12.7), 7.9 at 256 (10.9), 2.9 at 1024.
// Wall-clock stamps of one wavefront: 1.6 us per sub-range = prefetch issue 0.2 + terms 1.15 + touched 0.25 + waits 0.02:
// the loads ARE hidden; what is left is instruction issue -- ~550 wavefront instructions per sub-range whatever the
// lane use, and a mid-frequency term fills 20 of 64 lanes per 2048 documents.  Wider sub-ranges do not fit the LDS
// (8 B per document); the next step is a sparse (hashed) accumulator instead of dense arrays.
constexpr uint32_t BW_DOCS = 2048;
constexpr uint32_t BW_WAVES = 4; // wavefronts per workgroup
constexpr uint32_t BW_FNL = BW_DOCS / 64;   // fieldnorm bytes per lane and sub-range
constexpr uint32_t BW_FNV = BW_FNL / 16;    // ... as 16-byte loads
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Bm25WParams
{
    Bm25Params p;          // postings, batch, filter, cut / candidate buffers (bounds at BW_DOCS granularity: n_blocks = sub-ranges)
    uint32_t spi;          // sub-ranges per item
    uint32_t n_chunks;     // ceil(n_blocks / spi)
    uint32_t cstep;        // TOPK sample: item chunk = index * cstep
    uint32_t n_items_c;    // chunks this launch walks (n_chunks, or the sample's ceil(n_chunks / cstep))
    uint32_t lists;        // TOPK: lists per slot in `partial` (= n_items_c)
    const uint32_t * items; // bm25p_kernel: [n_items_tab][4] = query, first sub-range, end sub-range, list index (nullptr: (chunk, query) items)
    uint32_t n_items_tab;
    uint32_t dbg;          // experiment masks of bm25p_kernel (option bm25_dbg; results are wrong with any bit set)
    uint32_t sub_docs;     // documents per sub-range (BW_DOCS for bm25w_kernel; chosen per batch for bm25p_kernel)
};

template <int MODE, int R, int NF>
__global__ __launch_bounds__(64 * BW_WAVES) void bm25w_kernel(const Bm25WParams a)
{
    bm25_slot_signal(a.p);
    __shared__ float score[BW_WAVES][BW_DOCS];
    __shared__ uint16_t mask[BW_WAVES][BW_DOCS];
    __shared__ uint16_t touched[BW_WAVES][BW_DOCS];
    __shared__ __attribute__((aligned(16))) uint8_t fnw[BW_WAVES][NF][BW_DOCS];
    __shared__ float cache[NF * 256];
    const Bm25Params & p = a.p;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float * sc = score[wave];
    uint16_t * mk = mask[wave];
    uint16_t * tl = touched[wave];
    for (uint32_t i = lane; i < BW_DOCS; i += 64)
    {
        sc[i] = 0.f;
        mk[i] = 0;
    }
    for (uint32_t i = tid; i < p.num_fields * 256; i += 64 * BW_WAVES)
        cache[i] = p.norm_cache[i];
    __syncthreads(); // the only workgroup barrier: the fieldnorm cache
    const uint32_t nslots = p.qsel ? *p.nsel : p.nq;
    const uint32_t nb1 = p.n_blocks + 1;
    const uint64_t n_items = (uint64_t)a.n_items_c * nslots;
    const uint32_t waves_total = gridDim.x * BW_WAVES;
    for (uint64_t item = (uint64_t)blockIdx.x * BW_WAVES + wave; item < n_items; item += waves_total)
    {
        const uint32_t ci = (uint32_t)(item / nslots), slot = (uint32_t)(item - (uint64_t)ci * nslots);
        const uint32_t chunk = ci * a.cstep;
        const uint32_t q = p.qsel ? p.qsel[slot] : slot;
        const uint32_t j0 = p.qoff[q], nt = p.qoff[q + 1] - j0;
        const uint16_t full = p.qfull[q];
        float cut = 0.f;
        if (MODE == BM25_EMIT)
        {
            const uint64_t ck = p.cut_keys[(size_t)q * p.cut_m + p.cut_m - 1];
            cut = ck == KEY_NONE ? 0.f : key_value<M_IP>(ck); // fewer than m sample hits: everything passes
        }
        // lane t = term t of the query (<= 64 terms)
        const uint32_t jt = j0 + (lane < nt ? lane : 0);
        const float w_l = p.weight[jt];
        const uint32_t bit_l = 1u << p.qgroup[jt], field_l = p.qfield[jt];
        const uint32_t s_begin = chunk * a.spi, s_end = s_begin + a.spi < p.n_blocks ? s_begin + a.spi : p.n_blocks;
        auto load_bounds = [&](uint32_t s, int64_t & b0, int64_t & b1) {
            const uint32_t sx = s < p.n_blocks ? s : p.n_blocks - 1; // past the item: a valid address, never used
            b0 = p.bounds[(size_t)jt * nb1 + sx];
            b1 = p.bounds_hi[(size_t)jt * nb1 + sx];
        };
        struct Win // first 64 postings of terms 0..3 of a sub-range
        {
            uint32_t doc[4], tf[4];
        };
        auto load_win = [&](int64_t b0, Win & wd) {
#pragma unroll
            for (int u = 0; u < 4; u++)
            {
                const int64_t pp = (int64_t)readlane64((uint64_t)b0, u) + lane; // lanes past the slice read on: valid memory, masked later
                const uint64_t pc = (uint64_t)pp < p.last_posting ? (uint64_t)pp : p.last_posting;
                wd.doc[u] = p.doc_ids[pc];
                wd.tf[u] = p.tfs[pc];
            }
        };
        u32x4 f_next[NF == 1 ? BW_FNV : 1];
        auto load_fn = [&](uint32_t s) { // NF == 1: the sub-range's 2048 fieldnorm bytes, 32 per lane
            const uint32_t base = s * BW_DOCS;
            const size_t at = (size_t)base + lane * BW_FNL;
            const size_t lim = p.num_docs >= BW_FNL + 16 ? (p.num_docs - BW_FNL) & ~(size_t)15 : 0; // the corpus' tail: a clamped,
            const uint8_t * src = p.fieldnorm_ids + (at < lim ? at : lim); // aligned address (its bytes are re-read one by one)
#pragma unroll
            for (uint32_t v = 0; v < (NF == 1 ? BW_FNV : 1); v++)
                f_next[v] = *reinterpret_cast<const u32x4 *>(src + 16 * v);
        };
        int64_t b0c, b1c, b0n, b1n;
        load_bounds(s_begin, b0c, b1c);
        load_bounds(s_begin + 1, b0n, b1n);
        Win wc;
        load_win(b0c, wc);
        if (NF == 1)
            load_fn(s_begin);
        WaveTopK<R> top;
        top.init();
        for (uint32_t s = s_begin; s < s_end; s++)
        {
            const uint32_t base = s * BW_DOCS;
            // ---- land what was prefetched for this sub-range, issue the prefetches for the next ones
            if (NF == 1)
            {
                const size_t at = (size_t)base + lane * BW_FNL;
                const size_t lim = p.num_docs >= BW_FNL + 16 ? (p.num_docs - BW_FNL) & ~(size_t)15 : 0;
                if (at < lim)
                {
#pragma unroll
                    for (uint32_t v = 0; v < (NF == 1 ? BW_FNV : 1); v++)
                        *reinterpret_cast<u32x4 *>(&fnw[wave][0][lane * BW_FNL + 16 * v]) = f_next[v];
                }
                else
                    for (uint32_t i = 0; i < BW_FNL; i++) // the last lanes of the corpus' last sub-range only
                    {
                        const size_t d = at + i;
                        if (d < p.num_docs)
                            fnw[wave][0][lane * BW_FNL + i] = p.fieldnorm_ids[d];
                    }
            }
            else
                for (uint32_t f = 0; f < p.num_fields; f++)
                    for (uint32_t i = lane; i < BW_DOCS; i += 64)
                        fnw[wave][f][i] = (size_t)base + i < p.num_docs ? p.fieldnorm_ids[(size_t)f * p.num_docs + base + i] : (uint8_t)0;
            int64_t b0f, b1f;
            load_bounds(s + 2, b0f, b1f);
            Win wn;
            load_win(b0n, wn);
            if (NF == 1 && s + 1 < s_end)
                load_fn(s + 1);
            __builtin_amdgcn_sched_barrier(0);
            // ---- score the sub-range, term by term.  One posting = a chain of dependent LDS accesses (fieldnorm byte ->
            // cache entry -> score read -> write, mask read -> write): ~1000 cycles per term when walked one term at a time.
            // The lookups and the division of the first four terms' prefetched windows do not depend on each other: they
            // are issued together; only the read-modify-writes stay in term order.
            uint32_t ntouch = 0;
            auto add = [&](bool live, uint32_t loc, float sv, uint32_t bit) {
                bool fresh = false;
                if (live)
                {
                    sc[loc] = __fadd_rn(sc[loc], sv);
                    const uint16_t old = mk[loc];
                    mk[loc] = old | (uint16_t)bit;
                    fresh = old == 0;
                }
                const uint64_t fm = __ballot(fresh);
                if (fresh)
                    tl[ntouch + __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u))] = (uint16_t)loc;
                ntouch += (uint32_t)__popcll(fm);
            };
            float sv4[4];
            uint32_t loc4[4];
            bool live4[4];
#pragma unroll
            for (int u = 0; u < 4; u++)
            {
                const int64_t p0 = (int64_t)readlane64((uint64_t)b0c, u), p1 = (int64_t)readlane64((uint64_t)b1c, u);
                live4[u] = (uint32_t)u < nt && p0 + lane < p1;
                loc4[u] = live4[u] ? wc.doc[u] - base : 0u;
            }
            uint32_t fid4[4];
#pragma unroll
            for (int u = 0; u < 4; u++)
            {
                const uint32_t field = (uint32_t)__builtin_amdgcn_readlane((int)field_l, u);
                fid4[u] = fnw[wave][NF == 1 ? 0 : field][loc4[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
            {
                const uint32_t field = (uint32_t)__builtin_amdgcn_readlane((int)field_l, u);
                const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w_l), u));
                const float tff = (float)wc.tf[u];
                sv4[u] = __fmul_rn(w, __fdiv_rn(tff, __fadd_rn(tff, cache[field * 256 + fid4[u]])));
            }
            for (uint32_t t = 0; t < nt; t++)
            {
                // t is wave-uniform: v_readlane with a scalar lane index, not a trip through the LDS crossbar
                const int64_t p0 = (int64_t)readlane64((uint64_t)b0c, (int)t), p1 = (int64_t)readlane64((uint64_t)b1c, (int)t);
                const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w_l), (int)t));
                const uint32_t bit = (uint32_t)__builtin_amdgcn_readlane((int)bit_l, (int)t),
                               field = (uint32_t)__builtin_amdgcn_readlane((int)field_l, (int)t);
                const float * fc = cache + field * 256;
                const uint8_t * fb = fnw[wave][NF == 1 ? 0 : field];
                int64_t pp = p0;
                if (t < 4) // the prefetched window, already scored
                {
                    const bool lv = t == 0 ? live4[0] : (t == 1 ? live4[1] : (t == 2 ? live4[2] : live4[3]));
                    const uint32_t lc = t == 0 ? loc4[0] : (t == 1 ? loc4[1] : (t == 2 ? loc4[2] : loc4[3]));
                    const float sv = t == 0 ? sv4[0] : (t == 1 ? sv4[1] : (t == 2 ? sv4[2] : sv4[3]));
                    if (p0 < p1)
                        add(lv, lc, sv, bit);
                    pp = p0 + 64;
                }
                for (; pp < p1; pp += 64) // long slices, and the terms past the fourth: plain loads
                {
                    const bool live = pp + lane < p1;
                    const uint64_t pc = live ? (uint64_t)(pp + lane) : (uint64_t)p0;
                    const uint32_t d = p.doc_ids[pc], f = p.tfs[pc];
                    uint32_t loc = 0;
                    float sv = 0.f;
                    if (live)
                    {
                        loc = d - base;
                        const float tff = (float)f;
                        sv = __fmul_rn(w, __fdiv_rn(tff, __fadd_rn(tff, fc[fb[loc]])));
                    }
                    add(live, loc, sv, bit);
                }
            }
            // ---- the documents this sub-range touched
            for (uint32_t i0 = 0; i0 < ntouch; i0 += 64)
            {
                const uint32_t i = i0 + lane;
                uint64_t key = KEY_NONE;
                if (i < ntouch)
                {
                    const uint32_t loc = tl[i], docid = base + loc;
                    const float sv = sc[loc];
                    bool ok = (p.operator_or || mk[loc] == full) && (MODE != BM25_EMIT || sv >= cut);
                    sc[loc] = 0.f;
                    mk[loc] = 0;
                    if (ok && p.alive)
                        ok = docid < p.nbits && ((p.alive[docid >> 6] >> (docid & 63)) & 1);
                    if (ok)
                        key = make_key<M_IP>(sv, docid);
                }
                if (MODE == BM25_EMIT)
                {
                    if (key != KEY_NONE)
                    {
                        const uint32_t pos = atomicAdd(&p.ccnt[q], 1u);
                        if (pos < p.cand_cap)
                            p.cand[(size_t)q * BM25_CAND_CAP + pos] = key;
                    }
                }
                else
                    top.offer(key, p.kk, lane);
            }
            // ---- rotate
            b0c = b0n;
            b1c = b1n;
            b0n = b0f;
            b1n = b1f;
            wc = wn;
        }
        if (MODE == BM25_TOPK)
            top.store(p.partial + ((size_t)slot * a.lists + ci) * p.kk, p.kk, lane);
    }
}

/// One block of 1024 threads per query: top-k of its candidates -> results; a query whose candidates cannot prove its top-k
/// (list overflowed, or a real cut let fewer than k through) joins the fallback queue instead.
/// Selection by RANK: the <= 2048 candidate keys go to LDS and every thread counts the keys below its own (a key holds the
/// document id: no two are equal); rank < k is the output position.  cnt^2 / 1024 broadcast LDS reads and compares per
/// thread -- ~5 us at the usual 500-1000 candidates, where four wavefronts feeding WaveTopK insertions took 33 us.
constexpr uint32_t BM25_SELECT_THREADS = 1024;
static __global__ __launch_bounds__(BM25_SELECT_THREADS) void bm25_select_kernel(const uint64_t * cand, const uint32_t * ccnt, uint32_t cand_cap,
                                                                                 const uint64_t * cut_keys, uint32_t cut_m, uint32_t k,
                                                                                 int64_t * out_ids, float * out_scores, uint32_t * failq,
                                                                                 uint32_t * nfail, unsigned long long * stat_fail)
{
    __shared__ __attribute__((aligned(16))) uint64_t keys[BM25_CAND_CAP + 2];
    const uint32_t tid = threadIdx.x, q = blockIdx.x;
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
    for (uint32_t i = tid; i < cnt + 2; i += BM25_SELECT_THREADS)
        keys[i] = i < cnt ? src[i] : KEY_NONE; // two pad keys: the count loop reads pairs
    for (uint32_t i = cnt + tid; i < k; i += BM25_SELECT_THREADS) // fewer candidates than k: the tail is "no hit"
    {
        out_ids[(size_t)q * k + i] = -1;
        out_scores[(size_t)q * k + i] = key_value<M_IP>(KEY_NONE);
    }
    __syncthreads();
    for (uint32_t i = tid; i < cnt; i += BM25_SELECT_THREADS)
    {
        const uint64_t mine = keys[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < cnt; j += 2)
        {
            const uint64_t k0 = keys[j], k1 = keys[j + 1]; // the same address in every lane: one broadcast read of 16 bytes
            rank += (k0 < mine ? 1u : 0u) + (k1 < mine ? 1u : 0u); // the pad keys are KEY_NONE: never below a candidate
        }
        if (rank < k)
        {
            out_ids[(size_t)q * k + rank] = (int64_t)(uint32_t)mine;
            out_scores[(size_t)q * k + rank] = key_value<M_IP>(mine);
        }
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
