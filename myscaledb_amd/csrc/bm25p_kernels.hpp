// bm25p_kernels.hpp -- BM25 scorer with the POSTING as the unit of work (seam B, TantivyIndexStore.cpp:900-954).
//
// bm25w_kernel (bm25_kernels.hpp) keeps a dense per-document accumulator for 2048 documents per wavefront and walks the
// query's terms one after the other over it: ~550 wavefront instructions per 2048 documents whatever the lane use, and a
// mid-frequency term has ~10 postings there (profiles/r02_bm25.txt: 9.5 us per query at batch 64 where the bytes cost 0.15).
// Here nothing is indexed by document:
//   * a work item = (chunk of `spi` consecutive sub-ranges, query), one wavefront, as before; inside it the wavefront cuts
//     WINDOWS: as many whole sub-ranges as hold <= BP_CAP postings of all the query's terms together (bounds at sub-range
//     granularity from bm25_bounds_kernel, lane t = term t).  The sub-range size is chosen per batch from the posting
//     density so that a window is a few sub-ranges; a single sub-range over the cap (a local density spike) is cut by
//     document id with a per-lane binary search -- slow and rare, batches of frequent terms go to bm25w_kernel instead;
//   * the window's postings are ONE flat sequence (term 0's slice, term 1's, ...): lane l takes records l, l + 64, ...,
//     every lane busy but the last few.  Per record: (doc, tf) coalesced, the fieldnorm byte gathered, the record's
//     partial score w * tf / (tf + cache[field][fieldnorm]) computed with the scorer's f32 operations, (doc, partial) stored
//     to LDS in flat order -- the slice of a term is sorted by document;
//   * a document is OWNED by its record in the first term that has it.  For every other term u in term order each lane
//     binary-searches its records' documents in term u's LDS slice (all BP_RMAX records of the lane side by side: the LDS
//     round trips of a search step overlap): found in an earlier term -> not the owner; found in a later term -> add that
//     record's partial.  The owner therefore sums the present terms in QUERY-TERM ORDER starting from its own partial --
//     the same f32 additions as the dense accumulator's (0 + s is s), bit for bit; the token-group mask for AND queries
//     is collected on the way;
//   * owners that qualify (operator, cut, filter) are appended with ONE atomic per window (EMIT) or offered to the item's
//     WaveTopK (TOPK: sample, small corpora, exact fallback).
// Cost per window of ~300-500 postings: ~1000 wavefront instructions, three dependent memory round trips (bounds ->
// postings -> fieldnorms) hidden by 16-20 resident wavefronts per CU.
#pragma once
#include "bm25_kernels.hpp"

#pragma clang fp contract(off)

namespace msvs
{

constexpr uint32_t BP_WAVES = 4;             // wavefronts per workgroup
#ifndef MSVS_BP_BLOCKS_PER_CU
#define MSVS_BP_BLOCKS_PER_CU 4
#endif
constexpr uint32_t BP_BLOCKS_PER_CU = MSVS_BP_BLOCKS_PER_CU; // resident workgroups per CU the item tables are sized for (= waves per SIMD)
// (3: 135-140 VGPRs, no spills, scratch 0 -- measured round 4: 0.287 vs 0.283 ms per 64-query batch, 0.698 vs 0.628 at 256, 0.189 vs 0.199
//  at 16: the fourth wavefront per SIMD is worth more than the 2-8 spilled VGPRs cost)
constexpr uint32_t BP_RMAX = 8;              // records per lane and window
constexpr uint32_t BP_CAP = 64 * BP_RMAX;    // postings per window
constexpr uint32_t BP_SLOTS = 4096;          // hash slots of the shared-document filter (a power of two)
constexpr uint32_t BP_STAGE = 128;           // EMIT: staged keys per wavefront (>= 64 free slots after a flush)
constexpr uint32_t BP_MIN_DOCS = 512;        // sub-range sizes the host picks from (powers of two)
constexpr uint32_t BP_MAX_DOCS = 8192;

/// LDS traffic of ONE wavefront: its ds operations execute in order, so a wait + a compiler barrier is all a
/// write -> read-by-another-lane hand-over needs.
__device__ __forceinline__ void bp_wave_lds_fence()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint64_t bp_wave_sum(uint64_t v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        v += (uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)v, o) | ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o) << 32);
    return v;
}

template <int MODE, int R>
__global__ __launch_bounds__(64 * BP_WAVES, BP_BLOCKS_PER_CU) void bm25p_kernel(const Bm25WParams a)
{
    bm25_slot_signal(a.p);
    __shared__ uint32_t rdoc_s[BP_WAVES][BP_CAP];
    __shared__ float rsc_s[BP_WAVES][BP_CAP];
    __shared__ uint32_t bm_s[BP_WAVES][2 * BP_SLOTS / 32];   // word pairs: seen | dup bits of 32 hash slots
    __shared__ uint16_t flist_s[BP_WAVES][BP_CAP];           // flat positions of the records to look at one by one
    __shared__ uint64_t stg_key_s[BP_WAVES][BP_STAGE];       // EMIT: keys waiting for their flush ...
    __shared__ uint32_t stg_q_s[BP_WAVES][BP_STAGE];         // ... and their queries
    __shared__ uint64_t tbase_s[BP_WAVES][64]; // posting index of the term's first record minus its flat position
    __shared__ float tw_s[BP_WAVES][64];
    __shared__ uint32_t tfb_s[BP_WAVES][64];   // field | token-group bit << 8
    __shared__ float cache[BM25_MAX_FIELDS * 256];
    const Bm25Params & p = a.p;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t * const rdoc = rdoc_s[wave];
    float * const rsc = rsc_s[wave];
    uint64_t * const tbase = tbase_s[wave];
    float * const tw = tw_s[wave];
    uint32_t * const tfb = tfb_s[wave];
    uint32_t * const bm = bm_s[wave];
    uint16_t * const flist = flist_s[wave];
    uint64_t * const stg_key = stg_key_s[wave];
    uint32_t * const stg_q = stg_q_s[wave];
    for (uint32_t i = tid; i < p.num_fields * 256; i += 64 * BP_WAVES)
        cache[i] = p.norm_cache[i];
    for (uint32_t i = lane; i < 2 * BP_SLOTS / 32; i += 64)
        bm[i] = 0;
    __syncthreads(); // the only workgroup barrier
    // EMIT: passing keys wait in an LDS stage (with their query) and leave in rounds of 64: one returning atomic per
    // distinct query of a round instead of one per window -- a query's counter is ONE address for the whole chip, and
    // ~500 windows per query each taking their turn at it was a quarter of the kernel
    uint32_t stg_cnt = 0;
    auto flush = [&]() {
        bp_wave_lds_fence();
        for (uint32_t i0 = 0; i0 < stg_cnt; i0 += 64)
        {
            const bool have = i0 + lane < stg_cnt;
            const uint32_t qe = have ? stg_q[i0 + lane] : 0xFFFFFFFFu;
            const uint64_t ke = have ? stg_key[i0 + lane] : KEY_NONE;
            uint64_t rem = __ballot(have);
            uint32_t leader = lane, rank = 0, count = 0;
            while (rem)
            {
                const int lead = __builtin_ctzll(rem);
                const uint32_t q0 = (uint32_t)__builtin_amdgcn_readlane((int)qe, lead);
                const uint64_t m = __ballot(have && qe == q0);
                if (have && qe == q0)
                {
                    leader = (uint32_t)lead;
                    rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    count = (uint32_t)__popcll(m);
                }
                rem &= ~m;
            }
            uint32_t base = 0;
            if (have && leader == lane)
                base = atomicAdd(&p.ccnt[qe], count); // the leaders of all groups in one instruction
            base = (uint32_t)__shfl((int)base, (int)leader);
            if (have && base + rank < p.cand_cap)
                p.cand[(size_t)qe * BM25_CAND_CAP + base + rank] = ke;
        }
        stg_cnt = 0;
        bp_wave_lds_fence();
    };
    const uint32_t nslots = p.qsel ? *p.nsel : p.nq;
    const uint32_t nb1 = p.n_blocks + 1;
    const uint32_t sub_docs = a.sub_docs;
    const uint64_t n_items_u = (uint64_t)a.n_items_c * nslots;
    // Static stride over the items.  With (chunk, query) items and a wavefront count that is a multiple of the batch size every
    // wavefront would see the SAME query for the whole launch, and a query of frequent terms holds six times the postings of
    // a query of rare ones (the launch lasted six times its average wavefront; handing items out through global atomics is
    // worse: device-scope atomics of 4096 wavefronts on one line serialise beyond the XCDs' L2s, ~100 us per turn).  So:
    //   the EMIT and the SAMPLE launch walk a host-built item table (a.items: query, first sub-range, end, list index) whose
    //   items hold about the same number of postings whatever the query -- long document ranges for rare terms, short ones
    //   for frequent terms;
    //   the other TOPK launches keep (chunk, query) items and rotate the query with the chunk.
    const uint32_t waves_total = gridDim.x * BP_WAVES;
    const uint64_t n_items = a.items ? (uint64_t)a.n_items_tab : n_items_u;
    for (uint64_t item = (uint64_t)blockIdx.x * BP_WAVES + wave; item < n_items; item += waves_total)
    {
        uint32_t ci = 0, slot, s_begin, s_end;
        if (a.items)
        {
            const uint4 e = reinterpret_cast<const uint4 *>(a.items)[item];
            slot = e.x;
            s_begin = e.y;
            s_end = e.z;
            ci = e.w; // TOPK: which of the query's lists the item fills
        }
        else
        {
            ci = (uint32_t)(item / nslots);
            slot = (uint32_t)((item - (uint64_t)ci * nslots + ((uint64_t)ci * nslots) / waves_total) % nslots);
            const uint32_t chunk = ci * a.cstep;
            s_begin = chunk * a.spi;
            s_end = s_begin + a.spi < p.n_blocks ? s_begin + a.spi : p.n_blocks;
        }
        const uint32_t q = p.qsel ? p.qsel[slot] : slot;
        const uint32_t j0 = p.qoff[q], nt = p.qoff[q + 1] - j0;
        const uint32_t full = p.qfull[q];
        float cut = 0.f;
        if (MODE == BM25_EMIT)
        {
            const uint64_t ck = p.cut_keys[(size_t)q * p.cut_m + p.cut_m - 1];
            cut = ck == KEY_NONE ? 0.f : key_value<M_IP>(ck); // fewer than m sample hits: everything passes
        }
        // lane t = term t of the query (<= 64 terms)
        const bool has_term = lane < nt;
        const uint32_t jt = j0 + (has_term ? lane : 0);
        const float w_l = p.weight[jt];
        const uint32_t fb_l = (uint32_t)p.qfield[jt] | (1u << p.qgroup[jt]) << 8;
        bp_wave_lds_fence(); // the previous item's readers are done with the term tables
        tw[lane] = w_l;
        tfb[lane] = fb_l;
        const int64_t * const bnd = p.bounds + (size_t)jt * nb1;
        int64_t lo_l = has_term ? bnd[s_begin] : 0;
        WaveTopK<R> top;
        top.init();
        // what leaves a window, one key per lane: offered to the item's list (TOPK), or staged with its query (EMIT)
        auto out_one = [&](const bool ok, const uint64_t key) {
            if (MODE == BM25_TOPK)
            {
                top.offer(ok ? key : KEY_NONE, p.kk, lane);
                return;
            }
            const uint64_t m = __ballot(ok);
            if (!m)
                return;
            const uint32_t n = (uint32_t)__popcll(m);
            if (stg_cnt + n > BP_STAGE)
                flush();
            if (ok)
            {
                const uint32_t at = stg_cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                stg_key[at] = key;
                stg_q[at] = q;
            }
            stg_cnt += n;
        };
        // sub-ranges per window from the item's own density: up to 7/8 of the cap expected (an overflow halves the window)
        uint32_t W = s_end - s_begin;
        {
            const int64_t end_l = has_term ? bnd[s_end] : 0;
            const uint64_t total = bp_wave_sum((uint64_t)(end_l - lo_l));
            if (total)
            {
                const uint64_t w = (uint64_t)(BP_CAP * 7 / 8) * (s_end - s_begin) / total;
                W = w < 1 ? 1u : (w < W ? (uint32_t)w : W);
            }
            else
                W = 0; // nothing of this query in the item
        }
        uint32_t s = W ? s_begin : s_end;
        bool splitting = false;
        uint32_t d_lo = 0, d_end = 0;
        int64_t end_l = 0;
        // the end bound of the NEXT window's first guess is requested before the current window is processed
        uint32_t wc_next = s < s_end ? (s_end - s < W ? s_end - s : W) : 0;
        int64_t hi_pref = has_term && wc_next ? bnd[s + wc_next] : 0;
        while (s < s_end)
        {
            // ---- the next window: slices [lo_l, hi_l) of the terms, `tot` <= BP_CAP postings together
            int64_t hi_l = lo_l;
            uint32_t tot = 0;
            if (!splitting)
            {
                uint32_t wc = wc_next;
                hi_l = hi_pref;
                uint64_t t64;
                for (;;)
                {
                    t64 = bp_wave_sum((uint64_t)(hi_l - lo_l));
                    if (t64 <= BP_CAP || wc == 1)
                        break;
                    wc >>= 1;
                    hi_l = has_term ? bnd[s + wc] : 0;
                }
                if (t64 > BP_CAP)
                {
                    splitting = true; // one sub-range over the cap: cut it by document id
                    d_lo = s * sub_docs;
                    d_end = (uint64_t)d_lo + sub_docs < p.num_docs ? d_lo + sub_docs : p.num_docs;
                    end_l = hi_l;
                }
                else
                    s += wc;
                tot = (uint32_t)t64;
            }
            if (splitting)
            {
                uint32_t d_hi = d_end;
                hi_l = end_l;
                uint64_t t64 = bp_wave_sum((uint64_t)(hi_l - lo_l));
                while (t64 > BP_CAP && d_hi - d_lo > 1) // a single document holds <= nt <= 64 postings
                {
                    d_hi = d_lo + (d_hi - d_lo) / 2;
                    int64_t l2 = lo_l, h2 = hi_l;
                    while (l2 < h2)
                    {
                        const int64_t mid = (l2 + h2) >> 1;
                        if (p.doc_ids[mid] < d_hi)
                            l2 = mid + 1;
                        else
                            h2 = mid;
                    }
                    hi_l = l2;
                    t64 = bp_wave_sum((uint64_t)(hi_l - lo_l));
                }
                tot = (uint32_t)t64;
                d_lo = d_hi;
                if (d_lo >= d_end)
                {
                    splitting = false;
                    s += 1;
                }
            }
            if (!splitting)
            {
                wc_next = s < s_end ? (s_end - s < W ? s_end - s : W) : 0;
                hi_pref = has_term && wc_next ? bnd[s + wc_next] : 0;
            }
            const int64_t lo_w = lo_l;
            lo_l = hi_l;
            if (tot == 0 || (a.dbg & 8))
                continue;
            // ---- flat positions: exclusive prefix of the slice lengths over the lanes
            const uint32_t len_l = (uint32_t)(hi_l - lo_w);
            uint32_t inc = len_l;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1)
            {
                const uint32_t v = (uint32_t)__shfl_up((int)inc, o);
                if (lane >= (uint32_t)o)
                    inc += v;
            }
            const uint32_t pre_l = inc - len_l;
            bp_wave_lds_fence(); // the previous window's searches are done with rdoc / rsc / tbase
            tbase[lane] = (uint64_t)lo_w - pre_l;
            uint32_t t_r[BP_RMAX];
#pragma unroll
            for (uint32_t r = 0; r < BP_RMAX; r++)
                t_r[r] = 0;
            for (uint32_t u = 1; u < nt; u++)
            {
                const uint32_t pu = (uint32_t)__builtin_amdgcn_readlane((int)pre_l, (int)u);
#pragma unroll
                for (uint32_t r = 0; r < BP_RMAX; r++)
                    t_r[r] += r * 64 + lane >= pu ? 1u : 0u; // the LAST term whose slice starts at or before the record
            }
            bp_wave_lds_fence();
            // ---- the records: postings, fieldnorm bytes, partial scores (register rows past the window's last record are skipped)
            const uint32_t nr = (tot + 63) >> 6;
            uint32_t doc_r[BP_RMAX], tf_r[BP_RMAX];
#pragma unroll
            for (uint32_t r = 0; r < BP_RMAX; r++)
            {
                if (r >= nr)
                    break;
                const uint32_t f = r * 64 + lane;
                const bool live = f < tot;
                if (!live)
                    t_r[r] = 0;
                const uint64_t pp = live ? tbase[t_r[r]] + f : (uint64_t)lo_w; // idle lanes re-read a valid posting
                const uint64_t pc = pp < p.last_posting ? pp : p.last_posting;
                doc_r[r] = p.doc_ids[pc];
                tf_r[r] = p.tfs[pc];
            }
            uint32_t fn_r[BP_RMAX];
#pragma unroll
            for (uint32_t r = 0; r < BP_RMAX; r++)
                if (r < nr)
                    fn_r[r] = (a.dbg & 2) ? (doc_r[r] & 63u) : p.fieldnorm_ids[(size_t)(tfb[t_r[r]] & 0xffu) * p.num_docs + doc_r[r]];
            float s_r[BP_RMAX];
#pragma unroll
            for (uint32_t r = 0; r < BP_RMAX; r++)
            {
                if (r >= nr)
                    break;
                const uint32_t f = r * 64 + lane;
                const float tff = (float)tf_r[r];
                s_r[r] = __fmul_rn(tw[t_r[r]], __fdiv_rn(tff, __fadd_rn(tff, cache[(tfb[t_r[r]] & 0xffu) * 256 + fn_r[r]])));
                if (f < tot)
                {
                    rdoc[f] = doc_r[r];
                    rsc[f] = s_r[r];
                }
            }
            bp_wave_lds_fence();
            // ---- which records share their document with another record of the window?  A hashed bitmap says "maybe":
            // the first record of a slot sets `seen`, every later one sets `dup`; a record whose slot is not in `dup` is the
            // only posting of its document in the window -- owner, score = its own partial.
            uint32_t flags = 0; // bit r: record r of this lane may share its document
            if (nt > 1 && !(a.dbg & 1))
            {
#pragma unroll
                for (uint32_t r = 0; r < BP_RMAX; r++)
                    if (r * 64 + lane < tot)
                    {
                        const uint32_t slot = doc_r[r] & (BP_SLOTS - 1), bit = 1u << (slot & 31);
                        const uint32_t old = atomicOr(&bm[2 * (slot >> 5)], bit);
                        if (old & bit)
                            atomicOr(&bm[2 * (slot >> 5) + 1], bit);
                    }
                bp_wave_lds_fence();
#pragma unroll
                for (uint32_t r = 0; r < BP_RMAX; r++)
                {
                    if (r >= nr)
                        break;
                    const uint32_t slot = doc_r[r] & (BP_SLOTS - 1);
                    flags |= (r * 64 + lane < tot ? (bm[2 * (slot >> 5) + 1] >> (slot & 31)) & 1u : 0u) << r;
                }
                bp_wave_lds_fence();
#pragma unroll
                for (uint32_t r = 0; r < BP_RMAX; r++)
                    if (r * 64 + lane < tot)
                        *reinterpret_cast<uint2 *>(&bm[2 * ((doc_r[r] & (BP_SLOTS - 1)) >> 5)]) = make_uint2(0u, 0u);
            }
            // ---- the unshared records leave at once.  EMIT: few records of a window pass the cut, most windows have none:
            // the tests are collected as bits first and the rows nobody passes in are skipped
            uint32_t passbits = 0;
#pragma unroll
            for (uint32_t r = 0; r < BP_RMAX; r++)
            {
                if (r >= nr)
                    break;
                bool ok = !(a.dbg & 4) && r * 64 + lane < tot && !((flags >> r) & 1u) && (p.operator_or || (tfb[t_r[r]] >> 8) == full)
                    && (MODE != BM25_EMIT || s_r[r] >= cut);
                if (MODE == BM25_TOPK && ok && p.alive) // (EMIT tests the few records that pass the cut below)
                    ok = doc_r[r] < p.nbits && ((p.alive[doc_r[r] >> 6] >> (doc_r[r] & 63)) & 1);
                passbits |= (ok ? 1u : 0u) << r;
            }
            // TOPK, the item's list not full yet: offering a window's ~400 records one by one costs ~100 list insertions.  The
            // kk-th largest of the 64 lanes' BEST scores is a floor -- kk records at or above it exist -- and what lies below
            // it cannot be among the window's kk best: ~30 insertions instead
            if (MODE == BM25_TOPK && top.thr == KEY_NONE)
            {
                float best = -1.f; // scores are >= 0
#pragma unroll
                for (uint32_t r = 0; r < BP_RMAX; r++)
                    if (r < nr && ((passbits >> r) & 1u))
                        best = fmaxf(best, s_r[r]);
                if ((uint32_t)__popcll(__ballot(best >= 0.f)) >= p.kk)
                {
                    uint32_t rank = 0;
                    for (int j = 0; j < 64; j++)
                    {
                        const float sj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best), j));
                        rank += (sj > best || (sj == best && (uint32_t)j < lane)) ? 1u : 0u;
                    }
                    const uint64_t at = __ballot(rank == p.kk - 1);
                    const float floor_s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best), __builtin_ctzll(at)));
#pragma unroll
                    for (uint32_t r = 0; r < BP_RMAX; r++)
                        if (r < nr && s_r[r] < floor_s)
                            passbits &= ~(1u << r);
                }
            }
            if (MODE == BM25_TOPK || __ballot(passbits != 0))
            {
#pragma unroll
                for (uint32_t r = 0; r < BP_RMAX; r++)
                {
                    if (r >= nr)
                        break;
                    bool ok = (passbits >> r) & 1u;
                    if (MODE == BM25_EMIT && !__ballot(ok))
                        continue;
                    const uint32_t docid = doc_r[r];
                    if (MODE == BM25_EMIT && ok && p.alive)
                        ok = docid < p.nbits && ((p.alive[docid >> 6] >> (docid & 63)) & 1);
                    out_one(ok, make_key<M_IP>(s_r[r], docid));
                }
            }
            uint32_t nfl = 0;
            if (__ballot(flags != 0))
            {
#pragma unroll
                for (uint32_t r = 0; r < BP_RMAX; r++)
                {
                    if (r >= nr)
                        break;
                    const bool fl = (flags >> r) & 1u;
                    const uint64_t fm = __ballot(fl);
                    if (fl)
                        flist[nfl + __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u))] = (uint16_t)(r * 64 + lane);
                    nfl += (uint32_t)__popcll(fm);
                }
            }
            // ---- the shared ones (and the hash's false alarms), one per lane: the record is the OWNER of its document when no
            // earlier term has it; the owner adds the later terms' partials in term order
            if (nfl)
                bp_wave_lds_fence();
            for (uint32_t i0 = 0; i0 < nfl; i0 += 64)
            {
                const bool have = i0 + lane < nfl;
                const uint32_t f = have ? flist[i0 + lane] : 0u;
                const uint32_t docid = rdoc[f];
                float acc = rsc[f];
                uint32_t t = 0;
                for (uint32_t u = 1; u < nt; u++)
                    t += f >= (uint32_t)__builtin_amdgcn_readlane((int)pre_l, (int)u) ? 1u : 0u;
                uint32_t mask = tfb[t] >> 8;
                bool dead = !have;
                for (uint32_t u = 0; u < nt; u++)
                {
                    const uint32_t lu = (uint32_t)__builtin_amdgcn_readlane((int)len_l, (int)u);
                    if (lu == 0)
                        continue;
                    const uint32_t pu = (uint32_t)__builtin_amdgcn_readlane((int)pre_l, (int)u);
                    const uint32_t bit_u = (uint32_t)__builtin_amdgcn_readlane((int)fb_l, (int)u) >> 8;
                    uint32_t pos = 0; // entries of term u's slice below the document
                    for (uint32_t b = 1u << (31 - __builtin_clz(lu)); b; b >>= 1)
                    {
                        const uint32_t np = pos + b;
                        const uint32_t v = rdoc[pu + (np < lu ? np : lu) - 1];
                        if (np <= lu && v < docid)
                            pos = np;
                    }
                    const uint32_t at = pu + (pos < lu ? pos : lu - 1);
                    const bool found = pos < lu && rdoc[at] == docid;
                    const float sv = rsc[at];
                    if (found && u != t)
                    {
                        if (u < t)
                            dead = true;
                        else
                        {
                            acc = __fadd_rn(acc, sv);
                            mask |= bit_u;
                        }
                    }
                }
                bool ok = !(a.dbg & 4) && !dead && (p.operator_or || mask == full) && (MODE != BM25_EMIT || acc >= cut);
                if (ok && p.alive)
                    ok = docid < p.nbits && ((p.alive[docid >> 6] >> (docid & 63)) & 1);
                out_one(ok, make_key<M_IP>(acc, docid));
            }
        }
        if (MODE == BM25_TOPK)
            top.store(p.partial + ((size_t)slot * a.lists + ci) * p.kk, p.kk, lane);
    }
    if (MODE == BM25_EMIT)
        flush();
}

}
