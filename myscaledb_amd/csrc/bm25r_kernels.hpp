// bm25r_kernels.hpp -- BM25 posting scorer over SCORE-READY RECORDS (seam B, TantivyIndexStore.cpp:900-954), round 5.
//
// bm25p_kernel (bm25p_kernels.hpp) made the posting the unit of work; what it still paid per posting was (a) a third
// dependent memory round trip -- the fieldnorm byte of the posting's document, gathered -- followed by a table lookup and an
// f32 division, (b) every record written to LDS so that the few shared documents could be binary-searched in the other terms'
// slices, and (c) 128 VGPRs with spills.  ~1300 wavefront instructions per 512-posting window, 0.043 of HBM.
//
// Here:
//   * the postings are read as RECORDS (doc, tfn): tfn = tf / (tf + cache[field][fieldnorm[doc]]) with the scorer's own f32
//     operations, precomputed once per (posting set, fieldnorm cache) by bm25_rec_build_kernel -- derived data like the fp16
//     shadow of the vector side, 8 B per posting, rebuilt when the corpus statistics (average field length) change.  A record is
//     ONE coalesced 8-byte load; its partial score is one multiplication, w * tfn: the same f32 operations in the same order as
//     tantivy's Bm25Weight::score, bit for bit;
//   * records stay in REGISTERS; the next window's records are in flight while the current window is processed (two named
//     register sets, always 8 loads per window so that the waits are counted, not drained);
//   * shared documents: the hashed seen / dup bitmap pair flags every record whose slot holds a second record (all records of
//     a shared document are flagged, plus the hash's false alarms); only the FLAGGED records are written to LDS (in flat =
//     term order) and resolved among themselves: a flagged record with an earlier flagged record of the same document is not
//     the owner; the owner adds its later partners in list order -- the dense accumulator's additions in query-term order,
//     starting from its own partial (0 + s = s);
//   * everything else -- items, windows, staged EMIT appends, the TOPK floor -- is bm25p_kernel's.
#pragma once
#include "bm25p_kernels.hpp"
#include "mfma_scan_kernels.hpp" // wave_kth_word_radix

#pragma clang fp contract(off)

namespace msvs
{

struct Bm25RParams
{
    Bm25WParams w;
    const uint2 * rec; // [postings] (doc, bits of tf / (tf + cache[field][fieldnorm[doc]]))
};

/// rec[p] = (doc, tfn) for every posting.  term_field != nullptr: the posting's term (and so its text column) by a binary search
/// of the posting offset table (~18 steps through L2; the build runs once per statistics change).
static __global__ void bm25_rec_build_kernel(const uint32_t * doc_ids, const uint32_t * tfs, const uint8_t * fieldnorm_ids,
                                             const int64_t * post_off, const uint8_t * term_field, uint32_t num_terms,
                                             uint32_t num_docs, const float * cache, uint2 * rec, uint64_t n)
{
    const uint64_t gsz = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gsz)
    {
        uint32_t field = 0;
        if (term_field)
        {
            uint32_t lo = 0, hi = num_terms; // the last term whose first posting is at or before p
            while (hi - lo > 1)
            {
                const uint32_t mid = (lo + hi) >> 1;
                if ((uint64_t)post_off[mid] <= p)
                    lo = mid;
                else
                    hi = mid;
            }
            field = term_field[lo];
        }
        const uint32_t d = doc_ids[p];
        const float tff = (float)tfs[p];
        const uint32_t fn = fieldnorm_ids[(size_t)field * num_docs + d];
        rec[p] = make_uint2(d, __float_as_uint(__fdiv_rn(tff, __fadd_rn(tff, cache[field * 256 + fn]))));
    }
}

/// Inclusive prefix sum over the lanes of a wavefront: inside the 16-lane rows by DPP row_shr, across them by readlane.
__device__ __forceinline__ uint32_t br_wave_incl_scan(uint32_t v, uint32_t lane)
{
    v += dpp32<0x111>(v);
    v += dpp32<0x112>(v);
    v += dpp32<0x114>(v);
    v += dpp32<0x118>(v);
    const uint32_t r0 = __builtin_amdgcn_readlane((int)v, 15), r1 = __builtin_amdgcn_readlane((int)v, 31),
                   r2 = __builtin_amdgcn_readlane((int)v, 47);
    return v + (lane < 16 ? 0u : lane < 32 ? r0 : lane < 48 ? r0 + r1 : r0 + r1 + r2);
}

/// Postings of all the query's terms between two bounds, saturated: exact while <= 2^26 (a window holds <= BP_CAP), uniform.
__device__ __forceinline__ uint32_t br_wave_sum_sat(int64_t len)
{
    return wave_sum_u32(len > (int64_t)(1 << 20) ? (1u << 20) : (uint32_t)len);
}

#define BR_LOG2(x) (31 - __builtin_clz((unsigned)(x)))

template <int MODE, int R, int SLOTS>
__global__ __launch_bounds__(64 * BP_WAVES, 4) void bm25r_kernel(const Bm25RParams ar)
{
    bm25_slot_signal(ar.w.p);
    constexpr uint32_t BMW = 2 * SLOTS / 32;
    __shared__ __attribute__((aligned(16))) uint32_t bm_s[BP_WAVES][BMW]; // word pairs: seen | dup bits of 32 hash slots
    __shared__ uint32_t fdoc_s[BP_WAVES][BP_CAP];     // the flagged records of a window, in flat (= term) order: document ...
    __shared__ float fsc_s[BP_WAVES][BP_CAP];         // ... partial score ...
    __shared__ uint8_t ft_s[BP_WAVES][BP_CAP];        // ... term
    __shared__ uint64_t stg_key_s[BP_WAVES][BP_STAGE]; // EMIT: keys waiting for their flush ...
    __shared__ uint32_t stg_q_s[BP_WAVES][BP_STAGE];   // ... and their queries
    __shared__ uint64_t tbase_s[BP_WAVES][64];        // posting index of the term's first record minus its flat position
    __shared__ float tw_s[BP_WAVES][64];
    __shared__ uint32_t tfb_s[BP_WAVES][64];          // token-group bit << 8
    const Bm25WParams & a = ar.w;
    const Bm25Params & p = a.p;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t * const fdoc = fdoc_s[wave];
    float * const fsc = fsc_s[wave];
    uint8_t * const ft = ft_s[wave];
    uint64_t * const tbase = tbase_s[wave];
    float * const tw = tw_s[wave];
    uint32_t * const tfb = tfb_s[wave];
    uint32_t * const bm = bm_s[wave];
    uint64_t * const stg_key = stg_key_s[wave];
    uint32_t * const stg_q = stg_q_s[wave];
    for (uint32_t i = lane; i < BMW; i += 64)
        bm[i] = 0;
    bp_wave_lds_fence(); // wave-private LDS only: no workgroup barrier in this kernel
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
    const uint32_t waves_total = gridDim.x * BP_WAVES;
    const uint64_t n_items = a.items ? (uint64_t)a.n_items_tab : n_items_u;
    for (uint64_t item = (uint64_t)blockIdx.x * BP_WAVES + wave; item < n_items; item += waves_total)
    {
        uint32_t ci = 0, slot, s_begin, s_end;
        if (a.items)
        {
            const uint4 e = reinterpret_cast<const uint4 *>(a.items)[item];
            slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.x);
            s_begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.y);
            s_end = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.z);
            ci = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.w); // TOPK: which of the query's lists the item fills
        }
        else
        {
            ci = (uint32_t)(item / nslots);
            slot = (uint32_t)((item - (uint64_t)ci * nslots + ((uint64_t)ci * nslots) / waves_total) % nslots);
            const uint32_t chunk = ci * a.cstep;
            s_begin = chunk * a.spi;
            s_end = s_begin + a.spi < p.n_blocks ? s_begin + a.spi : p.n_blocks;
            ci = (uint32_t)__builtin_amdgcn_readfirstlane((int)ci);
            slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
            s_begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_begin);
            s_end = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_end);
        }
        const uint32_t q = (uint32_t)__builtin_amdgcn_readfirstlane((int)(p.qsel ? p.qsel[slot] : slot));
        const uint32_t j0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.qoff[q]);
        const uint32_t nt = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.qoff[q + 1]) - j0;
        const uint32_t full = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)p.qfull[q]);
        float cut = 0.f;
        if (MODE == BM25_EMIT)
        {
            const uint64_t ck = p.cut_keys[(size_t)q * p.cut_m + p.cut_m - 1];
            const float c = ck == KEY_NONE ? 0.f : key_value<M_IP>(ck); // fewer than m sample hits: everything passes
            cut = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(c)));
        }
        // lane t = term t of the query (<= 64 terms)
        const bool has_term = lane < nt;
        const uint32_t jt = j0 + (has_term ? lane : 0);
        const float w_l = p.weight[jt];
        const uint32_t fb_l = (1u << p.qgroup[jt]) << 8;
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
            const int64_t end0 = has_term ? bnd[s_end] : 0;
            const uint64_t total = bp_wave_sum((uint64_t)(end0 - lo_l));
            if (total)
            {
                const uint64_t w = (uint64_t)(BP_CAP * 7 / 8) * (s_end - s_begin) / total;
                W = w < 1 ? 1u : (w < W ? (uint32_t)w : W);
            }
            else
                W = 0; // nothing of this query in the item
            W = (uint32_t)__builtin_amdgcn_readfirstlane((int)W);
        }
        // ---- the window generator (bm25p_kernel's loop head as a function: it runs one window AHEAD of the scoring)
        uint32_t s = W ? s_begin : s_end;
        bool splitting = false;
        uint32_t d_lo = 0, d_end = 0;
        int64_t end_l = 0;
        uint32_t wc_next = s < s_end ? (s_end - s < W ? s_end - s : W) : 0;
        int64_t hi_pref = has_term && wc_next ? bnd[s + wc_next] : 0; // the end bound of the NEXT window's first guess
        // -> false: the item is exhausted (tot = 0, len_l = 0)
        auto next_window = [&](int64_t & lo_w, uint32_t & len_l, uint32_t & tot) -> bool {
            while (s < s_end)
            {
                int64_t hi_l = lo_l;
                tot = 0;
                if (!splitting)
                {
                    uint32_t wc = wc_next;
                    hi_l = hi_pref;
                    uint32_t t32;
                    for (;;)
                    {
                        t32 = br_wave_sum_sat(hi_l - lo_l);
                        if (t32 <= BP_CAP || wc == 1)
                            break;
                        wc >>= 1;
                        hi_l = has_term ? bnd[s + wc] : 0;
                    }
                    if (t32 > BP_CAP)
                    {
                        splitting = true; // one sub-range over the cap: cut it by document id
                        d_lo = s * sub_docs;
                        d_end = (uint64_t)d_lo + sub_docs < p.num_docs ? d_lo + sub_docs : p.num_docs;
                        end_l = hi_l;
                    }
                    else
                        s += wc;
                    tot = t32;
                }
                if (splitting)
                {
                    uint32_t d_hi = d_end;
                    hi_l = end_l;
                    uint32_t t32 = br_wave_sum_sat(hi_l - lo_l);
                    while (t32 > BP_CAP && d_hi - d_lo > 1) // a single document holds <= nt <= 64 postings
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
                        t32 = br_wave_sum_sat(hi_l - lo_l);
                    }
                    tot = t32;
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
                lo_w = lo_l;
                lo_l = hi_l;
                if (tot == 0 || (a.dbg & 8))
                    continue;
                len_l = (uint32_t)(hi_l - lo_w);
                return true;
            }
            lo_w = 0;
            len_l = 0;
            tot = 0;
            return false;
        };
        // ---- issue the 8 record loads of a window (always 8: rows past the window re-read record 0) and pack the records' terms
        auto issue = [&](const int64_t lo_w, const uint32_t len_l, const uint32_t tot, uint2 (&rb)[BP_RMAX], uint32_t (&tp)[2]) {
            const uint32_t pre_l = br_wave_incl_scan(len_l, lane) - len_l; // flat position of the term's first record
            bp_wave_lds_fence(); // the previous issue's readers are done with tbase
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
            tp[0] = tp[1] = 0;
#pragma unroll
            for (uint32_t r = 0; r < BP_RMAX; r++)
            {
                const uint32_t f = r * 64 + lane;
                const bool live = f < tot;
                const uint32_t t = live ? t_r[r] : 0u;
                const uint64_t pp = tbase[t] + f;
                const uint64_t pc = live ? (pp < p.last_posting ? pp : p.last_posting) : 0ull;
                rb[r] = ar.rec[pc];
                tp[r >> 2] |= t << (8 * (r & 3));
            }
        };
        // ---- score a window whose records have arrived
        auto process = [&](const uint32_t tot, const uint2 (&rb)[BP_RMAX], const uint32_t (&tp)[2]) {
            const uint32_t nr = (tot + 63) >> 6;
            float s_r[BP_RMAX];
#pragma unroll
            for (uint32_t r = 0; r < BP_RMAX; r++)
            {
                if (r >= nr)
                    break;
                const uint32_t t = (tp[r >> 2] >> (8 * (r & 3))) & 0xffu;
                s_r[r] = __fmul_rn(tw[t], __uint_as_float(rb[r].y));
            }
            // which records share their document with another record of the window?  A hashed bitmap says "maybe": the first
            // record of a slot sets `seen`, every later one sets `dup`; a record whose slot is not in `dup` is the only posting
            // of its document in the window -- owner, score = its own partial.
            uint32_t flags = 0; // bit r: record r of this lane may share its document
            if (nt > 1 && !(a.dbg & 1))
            {
#pragma unroll
                for (uint32_t r = 0; r < BP_RMAX; r++)
                    if (r < nr && r * 64 + lane < tot)
                    {
                        const uint32_t hs = rb[r].x & (SLOTS - 1), bit = 1u << (hs & 31);
                        const uint32_t old = atomicOr(&bm[2 * (hs >> 5)], bit);
                        if (old & bit)
                            atomicOr(&bm[2 * (hs >> 5) + 1], bit);
                    }
                bp_wave_lds_fence();
#pragma unroll
                for (uint32_t r = 0; r < BP_RMAX; r++)
                {
                    if (r >= nr)
                        break;
                    const uint32_t hs = rb[r].x & (SLOTS - 1);
                    flags |= (r * 64 + lane < tot ? (bm[2 * (hs >> 5) + 1] >> (hs & 31)) & 1u : 0u) << r;
                }
                bp_wave_lds_fence();
#pragma unroll
                for (uint32_t r = 0; r < BP_RMAX; r++)
                    if (r < nr && r * 64 + lane < tot)
                        *reinterpret_cast<uint2 *>(&bm[2 * ((rb[r].x & (SLOTS - 1)) >> 5)]) = make_uint2(0u, 0u);
            }
            // the unshared records leave at once.  EMIT: few records of a window pass the cut, most windows have none: the tests
            // are collected as bits first and the rows nobody passes in are skipped
            uint32_t passbits = 0;
#pragma unroll
            for (uint32_t r = 0; r < BP_RMAX; r++)
            {
                if (r >= nr)
                    break;
                const uint32_t t = (tp[r >> 2] >> (8 * (r & 3))) & 0xffu;
                bool ok = !(a.dbg & 4) && r * 64 + lane < tot && !((flags >> r) & 1u) && (p.operator_or || (tfb[t] >> 8) == full)
                    && (MODE != BM25_EMIT || s_r[r] >= cut);
                if (MODE == BM25_TOPK && ok && p.alive) // (EMIT tests the few records that pass the cut below)
                    ok = rb[r].x < p.nbits && ((p.alive[rb[r].x >> 6] >> (rb[r].x & 63)) & 1);
                passbits |= (ok ? 1u : 0u) << r;
            }
            // TOPK, the item's list not full yet: the kk-th largest of the 64 lanes' BEST scores is a floor -- kk records at or
            // above it exist -- and what lies below it cannot be among the window's kk best
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
                    const uint32_t docid = rb[r].x;
                    if (MODE == BM25_EMIT && ok && p.alive)
                        ok = docid < p.nbits && ((p.alive[docid >> 6] >> (docid & 63)) & 1);
                    out_one(ok, make_key<M_IP>(s_r[r], docid));
                }
            }
            // ---- the flagged ones: compacted to LDS in flat order (ascending term) and resolved among themselves
            uint32_t nfl = 0;
            if (__ballot(flags != 0))
            {
                bp_wave_lds_fence(); // the previous window's resolution is done with the lists
#pragma unroll
                for (uint32_t r = 0; r < BP_RMAX; r++)
                {
                    if (r >= nr)
                        break;
                    const bool fl = (flags >> r) & 1u;
                    const uint64_t fm = __ballot(fl);
                    if (fl)
                    {
                        const uint32_t at = nfl + __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                        fdoc[at] = rb[r].x;
                        fsc[at] = s_r[r];
                        ft[at] = (uint8_t)((tp[r >> 2] >> (8 * (r & 3))) & 0xffu);
                    }
                    nfl += (uint32_t)__popcll(fm);
                }
                bp_wave_lds_fence();
            }
            // ---- second look at the flagged few, one per lane: a DIFFERENT hash (of the whole document id) through the same bitmap
            // pair (cleared above).  Two records of one document meet again whatever the hash; the first filter's false alarms --
            // records whose documents only share their low bits: ~6 % of a window that spans 4 x SLOTS documents -- do not, and
            // leave with their own partial like the unshared ones.  What is flagged twice (true shares; a false alarm per ~30
            // windows) is resolved pairwise below.
            uint32_t f2 = 0; // bit c: record c * 64 + lane of the flagged list is flagged again
            const bool second = nfl != 0 && nfl <= 4 * 64 && !(a.dbg & 2);
            if (second)
            {
#pragma unroll
                for (uint32_t c = 0; c < 4; c++)
                    if (c * 64 < nfl && c * 64 + lane < nfl)
                    {
                        const uint32_t hs = (fdoc[c * 64 + lane] * 0x9E3779B1u) >> (32 - BR_LOG2(SLOTS)), bit = 1u << (hs & 31);
                        const uint32_t old = atomicOr(&bm[2 * (hs >> 5)], bit);
                        if (old & bit)
                            atomicOr(&bm[2 * (hs >> 5) + 1], bit);
                    }
                bp_wave_lds_fence();
#pragma unroll
                for (uint32_t c = 0; c < 4; c++)
                    if (c * 64 < nfl && c * 64 + lane < nfl)
                    {
                        const uint32_t hs = (fdoc[c * 64 + lane] * 0x9E3779B1u) >> (32 - BR_LOG2(SLOTS));
                        f2 |= ((bm[2 * (hs >> 5) + 1] >> (hs & 31)) & 1u) << c;
                    }
                bp_wave_lds_fence();
#pragma unroll
                for (uint32_t c = 0; c < 4; c++)
                    if (c * 64 < nfl && c * 64 + lane < nfl)
                    {
                        const uint32_t hs = (fdoc[c * 64 + lane] * 0x9E3779B1u) >> (32 - BR_LOG2(SLOTS));
                        *reinterpret_cast<uint2 *>(&bm[2 * (hs >> 5)]) = make_uint2(0u, 0u);
                    }
            }
            const bool any2 = !second || __ballot(f2 != 0) != 0; // uniform
            for (uint32_t i0 = 0; i0 < nfl; i0 += 64)
            {
                const uint32_t me = i0 + lane;
                const bool have = me < nfl;
                const uint32_t mi = have ? me : 0u;
                const uint32_t docid = fdoc[mi];
                float acc = fsc[mi];
                uint32_t mask = tfb[ft[mi]] >> 8;
                bool dead = !have;
                // the records flagged twice (all of them without the second look): their partners are flagged twice as well
                const bool mine2 = have && (!second || ((f2 >> (i0 >> 6)) & 1u));
                if (any2 && __ballot(mine2))
                    for (uint32_t j0 = 0; j0 < nfl; j0 += 64)
                    {
                        uint64_t cand_j = second ? __ballot((f2 >> (j0 >> 6)) & 1u) : __ballot(j0 + lane < nfl);
                        while (cand_j)
                        {
                            const uint32_t j = j0 + (uint32_t)__builtin_ctzll(cand_j);
                            cand_j &= cand_j - 1;
                            const uint32_t dj = fdoc[j]; // the same address in every lane: a broadcast read
                            const bool match = mine2 && dj == docid && j != me;
                            if (!__ballot(match))
                                continue;
                            const float sj = fsc[j];
                            const uint32_t bj = tfb[ft[j]] >> 8;
                            if (match)
                            {
                                if (j < me)
                                    dead = true; // an earlier term has the document: not the owner
                                else
                                {
                                    acc = __fadd_rn(acc, sj); // later terms in term order, onto the owner's own partial
                                    mask |= bj;
                                }
                            }
                        }
                    }
                bool ok = !(a.dbg & 4) && !dead && (p.operator_or || mask == full) && (MODE != BM25_EMIT || acc >= cut);
                if (ok && p.alive)
                    ok = docid < p.nbits && ((p.alive[docid >> 6] >> (docid & 63)) & 1);
                out_one(ok, make_key<M_IP>(acc, docid));
            }
        };
        // ---- two windows in flight, named register sets (a copy of a register waits for its load)
        int64_t lwA, lwB;
        uint32_t lenA, lenB, totA, totB;
        uint2 rbA[BP_RMAX], rbB[BP_RMAX];
        uint32_t tpA[2], tpB[2];
        bool more = next_window(lwA, lenA, totA);
        issue(lwA, lenA, totA, rbA, tpA);
        while (more)
        {
            more = next_window(lwB, lenB, totB);
            issue(lwB, lenB, totB, rbB, tpB);
            process(totA, rbA, tpA);
            if (!more)
                break;
            more = next_window(lwA, lenA, totA);
            issue(lwA, lenA, totA, rbA, tpA);
            process(totB, rbB, tpB);
        }
        if (MODE == BM25_TOPK)
            top.store(p.partial + ((size_t)slot * a.lists + ci) * p.kk, p.kk, lane);
    }
    if (MODE == BM25_EMIT)
        flush();
}

/// The cut of every query: the m-th best score word among its sample lists (n keys, KEY_NONE = unused), one wavefront per query,
/// words in registers, radix select (mfma_scan_kernels.hpp).  Only the SCORE of the m-th best key is ever read from cut_keys
/// (entry cut_m - 1), so the low word is zero; fewer than m real keys: KEY_NONE (everything passes, `real_cut` false).
/// Replaces a 64-block list merge (18 us per batch) by ~5 us.
template <int NW>
__global__ __launch_bounds__(BLOCK) void bm25_cut_kernel(const uint64_t * sample, uint32_t n, uint32_t nq, uint32_t m, uint64_t * cut_keys)
{
    __shared__ uint32_t hist_s[BLOCK / 64][256];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t q = blockIdx.x * (BLOCK / 64) + wave;
    if (q >= nq)
        return;
    const uint64_t * src = sample + (size_t)q * n;
    uint32_t word[NW];
#pragma unroll
    for (int u = 0; u < NW; u++)
    {
        const uint32_t i = (uint32_t)u * 64 + lane;
        word[u] = i < n ? (uint32_t)(src[i] >> 32) : 0xFFFFFFFFu;
    }
    const uint32_t H = wave_kth_word_radix<NW>(word, m, hist_s[wave], lane);
    if (lane == 0)
        cut_keys[(size_t)q * m + m - 1] = H == 0xFFFFFFFFu ? KEY_NONE : (uint64_t)H << 32;
}

/// bm25_cut_kernel for small batches: a WORKGROUP per query (one wavefront per query leaves a 64-query batch on 16 workgroups,
/// 64 words per lane: 12.7 us).  Each of the four wavefronts selects the m best of its quarter (their m-th best word H: the words
/// below it, then copies of H up to m), the first one selects the m-th best of the 4 m.
template <int NW>
__global__ __launch_bounds__(BLOCK) void bm25_cut_block_kernel(const uint64_t * sample, uint32_t n, uint32_t nq, uint32_t m, uint64_t * cut_keys)
{
    __shared__ uint32_t hist_s[BLOCK / 64][256];
    __shared__ uint32_t best_s[BLOCK / 64][64];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = blockIdx.x;
    const uint64_t * src = sample + (size_t)q * n;
    uint32_t word[NW];
#pragma unroll
    for (int u = 0; u < NW; u++)
    {
        const uint32_t i = (uint32_t)u * BLOCK + threadIdx.x;
        word[u] = i < n ? (uint32_t)(src[i] >> 32) : 0xFFFFFFFFu;
    }
    const uint32_t H = wave_kth_word_radix<NW>(word, m, hist_s[wave], lane); // 0xFFFFFFFF: fewer than m real keys in this quarter
    best_s[wave][lane] = H; // (m <= 64: lanes past m are not read)
    __builtin_amdgcn_wave_barrier();
    uint32_t run = 0;
#pragma unroll
    for (int u = 0; u < NW; u++)
    {
        const bool take = word[u] < H;
        const uint64_t mask = __ballot(take);
        if (take)
            best_s[wave][run + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] = word[u];
        run += (uint32_t)__popcll(mask); // < m in total: H is the m-th smallest
    }
    __syncthreads();
    if (wave != 0)
        return;
    uint32_t w4[BLOCK / 64];
#pragma unroll
    for (uint32_t v = 0; v < BLOCK / 64; v++)
        w4[v] = lane < m ? best_s[v][lane] : 0xFFFFFFFFu;
    const uint32_t G = wave_kth_word_radix<BLOCK / 64>(w4, m, hist_s[0], lane);
    if (lane == 0)
        cut_keys[(size_t)q * m + m - 1] = G == 0xFFFFFFFFu ? KEY_NONE : (uint64_t)G << 32;
}

/// bm25_select_kernel by SELECTION instead of ranking every candidate against every other (~500 candidates per query: 14 us per launch,
/// as long as the sample pass of a 64-query batch): the four wavefronts of a workgroup each select the k smallest score words of a
/// quarter of the keys (register radix select), the first one the k-th smallest of those 4 k -- the k-th smallest word G of all; the keys
/// with a word <= G (k of them plus the ties of the last score) are ranked among themselves.  More than 1024 such keys (a score
/// shared by a thousand documents at the cut): the full ranking.  Same outputs, same failure rule.
static __global__ __launch_bounds__(BLOCK) void bm25_select2_kernel(const uint64_t * cand, const uint32_t * ccnt, uint32_t cand_cap,
                                                                     const uint64_t * cut_keys, uint32_t cut_m, uint32_t k, int64_t * out_ids,
                                                                     float * out_scores, uint32_t * failq, uint32_t * nfail,
                                                                     unsigned long long * stat_fail)
{
    constexpr uint32_t NW = BM25_CAND_CAP / BLOCK; // 8 keys per thread
    __shared__ __attribute__((aligned(16))) uint64_t keys[BM25_CAND_CAP + 2];
    __shared__ uint64_t sel[1024 + 2];
    __shared__ uint32_t hist_s[BLOCK / 64][256];
    __shared__ uint32_t best_s[BLOCK / 64][256];
    __shared__ uint32_t s_cnt, s_G;
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
    uint64_t mine[NW];
    uint32_t word[NW];
#pragma unroll
    for (uint32_t u = 0; u < NW; u++)
    {
        const uint32_t i = (u * (BLOCK / 64) + wave) * 64 + lane; // wavefront w holds every fourth run of 64 keys
        mine[u] = i < cnt ? src[i] : KEY_NONE;
        word[u] = (uint32_t)(mine[u] >> 32);
    }
    for (uint32_t i = cnt + tid; i < k; i += BLOCK) // fewer candidates than k: the tail is "no hit"
    {
        out_ids[(size_t)q * k + i] = -1;
        out_scores[(size_t)q * k + i] = key_value<M_IP>(KEY_NONE);
    }
    if (tid == 0)
        s_cnt = 0;
    uint32_t G = 0xFFFFFFFFu;
    if (cnt > k)
    {
        const uint32_t H = wave_kth_word_radix<(int)NW>(word, k, hist_s[wave], lane); // 0xFFFFFFFF: fewer than k real keys in this quarter
        for (uint32_t i = lane; i < k; i += 64)
            best_s[wave][i] = H;
        __builtin_amdgcn_wave_barrier();
        uint32_t run = 0;
#pragma unroll
        for (uint32_t u = 0; u < NW; u++)
        {
            const bool take = word[u] < H;
            const uint64_t mask = __ballot(take);
            if (take)
                best_s[wave][run + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] = word[u];
            run += (uint32_t)__popcll(mask); // < k in total: H is the k-th smallest
        }
        __syncthreads();
        if (wave == 0)
        {
            uint32_t w4[16]; // 4 k <= 1024 words
#pragma unroll
            for (uint32_t v = 0; v < 16; v++)
            {
                const uint32_t i = v * 64 + lane, w = i / k, j = i - w * k; // (k >= 1)
                w4[v] = i < 4 * k ? best_s[w][j] : 0xFFFFFFFFu;
            }
            const uint32_t g = wave_kth_word_radix<16>(w4, k, hist_s[0], lane);
            if (lane == 0)
                s_G = g;
        }
        __syncthreads();
        G = s_G;
    }
    else
        __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < NW; u++)
        if (mine[u] != KEY_NONE && word[u] <= G)
        {
            const uint32_t at = atomicAdd(&s_cnt, 1u);
            if (at < 1024)
                sel[at] = mine[u];
        }
    __syncthreads();
    const uint32_t s = s_cnt;
    if (s <= 1024)
    {
        if (tid == 0)
            sel[s] = sel[s + 1] = KEY_NONE; // pad: the count loop reads pairs
        __syncthreads();
        for (uint32_t i = tid; i < s; i += BLOCK)
        {
            const uint64_t me = sel[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < s; j += 2)
                rank += (sel[j] < me ? 1u : 0u) + (sel[j + 1] < me ? 1u : 0u);
            if (rank < k)
            {
                out_ids[(size_t)q * k + rank] = (int64_t)(uint32_t)me;
                out_scores[(size_t)q * k + rank] = key_value<M_IP>(me);
            }
        }
        return;
    }
    // a thousand ties at the cut: rank everything
#pragma unroll
    for (uint32_t u = 0; u < NW; u++)
    {
        const uint32_t i = (u * (BLOCK / 64) + wave) * 64 + lane;
        if (i < cnt + 2)
            keys[i] = mine[u];
    }
    if (tid < 2)
        keys[cnt + tid] = KEY_NONE;
    __syncthreads();
    for (uint32_t i = tid; i < cnt; i += BLOCK)
    {
        const uint64_t me = keys[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < cnt; j += 2)
            rank += (keys[j] < me ? 1u : 0u) + (keys[j + 1] < me ? 1u : 0u);
        if (rank < k)
        {
            out_ids[(size_t)q * k + rank] = (int64_t)(uint32_t)me;
            out_scores[(size_t)q * k + rank] = key_value<M_IP>(me);
        }
    }
}

constexpr uint32_t BM25_SKIP_DOCS = 2048; // documents per stretch of the skip table (the sub-range sizes 2048 / 4096 / 8192 are multiples: their
                                          // bounds are table entries, no posting is read)

/// Skip table of a posting set: tab[row][c] = postings of term sel[row] with a document id below c * BM25_SKIP_DOCS (c = 0 .. n_c).
static __global__ void bm25_skip_build_kernel(const int64_t * post_off, const uint32_t * doc_ids, const uint32_t * sel, uint32_t n_rows,
                                              uint32_t n_c, uint32_t * tab)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_rows * (n_c + 1))
        return;
    const uint32_t row = (uint32_t)(i / (n_c + 1)), c = (uint32_t)(i - (size_t)row * (n_c + 1));
    const uint32_t term = sel[row];
    const uint64_t target = (uint64_t)c * BM25_SKIP_DOCS;
    const int64_t base = post_off[term];
    int64_t lo = base, hi = post_off[term + 1];
    while (lo < hi)
    {
        const int64_t mid = (lo + hi) >> 1;
        if (doc_ids[mid] < target)
            lo = mid + 1;
        else
            hi = mid;
    }
    tab[i] = (uint32_t)(lo - base);
}

/// bm25_bounds_kernel with an 8-ary search: the chain of dependent loads is what the launch costs (21 steps for a list of 2M
/// postings); seven independent probes per step cut it to 7.
typedef uint32_t bm25_u32x4 __attribute__((ext_vector_type(4)));
static __global__ void bm25_bounds8_kernel(const Bm25Params a, int64_t * bounds, int64_t * bounds_hi, uint32_t n_flat,
                                           uint32_t docs_per_block, uint32_t * zero, size_t n_zero, uint64_t * ones, size_t n_ones,
                                           const int32_t * skip_row, const uint32_t * skip_tab, uint32_t skip_n,
                                           bm25_u32x4 * tables_dst = nullptr, const bm25_u32x4 * tables_pinned = nullptr, size_t tables_n16 = 0)
{
    // Round 6: the batch's tables ride along too -- this launch copies them from their pinned slot to the device (every thread a
    // grid-stride share; the scorers behind it read the device copy) and reads the ONE table it needs itself, the flat terms, straight
    // from pinned memory (a.qterms points there for this launch).  A copy of its own in front cost a launch and the gap behind it.
    {
        const size_t gsz = (size_t)gridDim.x * gridDim.y * blockDim.x;
        for (size_t i = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < tables_n16; i += gsz)
            tables_dst[i] = __builtin_nontemporal_load(&tables_pinned[i]);
    }
    // grid: x over the boundaries of a term, y over the flat terms (the term and its list ends are uniform: scalar loads, and no
    // 64-bit division per thread -- the flat index form spent more on `i / nb1` than on its search)
    {
        const size_t gsz = (size_t)gridDim.x * gridDim.y * blockDim.x;
        const size_t i = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        for (size_t j = i; j < n_zero; j += gsz)
            zero[j] = 0;
        for (size_t j = i; j < n_ones; j += gsz)
            ones[j] = KEY_NONE;
    }
    const uint32_t nb1 = a.n_blocks + 1;
    for (uint32_t j = blockIdx.y; j < n_flat; j += gridDim.y)
    {
    if (skip_row && docs_per_block % BM25_SKIP_DOCS == 0)
    {
        // a frequent term whose boundaries are stretch boundaries: the table holds the answers.  The launch is a chain of dependent
        // loads per thread (term -> list ends / table row -> entry) at 8 wavefronts per SIMD: four entries per thread, their loads side
        // by side, quarter the chains
        const uint32_t term = a.qterms[j];
        const int32_t row = skip_row[term];
        if (row >= 0)
        {
            const uint32_t * const tab = skip_tab + (size_t)row * (skip_n + 1);
            const int64_t base = a.post_off[term];
            const uint32_t mul = docs_per_block / BM25_SKIP_DOCS;
            const uint32_t b0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
            uint32_t v[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++)
            {
                const uint64_t c = (uint64_t)(b0 + u) * mul;
                v[u] = b0 + u < nb1 ? tab[c < skip_n ? (uint32_t)c : skip_n] : 0u;
            }
#pragma unroll
            for (uint32_t u = 0; u < 4; u++)
                if (b0 + u < nb1)
                {
                    const size_t i = (size_t)j * nb1 + b0 + u;
                    bounds[i] = base + v[u];
                    if (b0 + u > 0 && bounds_hi)
                        bounds_hi[i - 1] = base + v[u];
                }
            continue;
        }
    }
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < nb1; b += gridDim.x * blockDim.x)
    {
    const size_t i = (size_t)j * nb1 + b;
    const uint32_t term = a.qterms[j];
    const uint64_t target = (uint64_t)b * docs_per_block;
    int64_t lo = a.post_off[term], hi = a.post_off[term + 1]; // the answer lies in [lo, hi]
    if (skip_row)
    {
        // a frequent term: the stretch of 8192 documents that holds the target (the posting set's skip table) -- two 8-ary steps
        // through a few cache lines instead of six through the whole list
        const int32_t row = skip_row[term];
        if (row >= 0)
        {
            const uint32_t * const tab = skip_tab + (size_t)row * (skip_n + 1);
            if (docs_per_block % BM25_SKIP_DOCS == 0)
            {
                // the boundary IS a stretch boundary: the table holds the answer (the doc ids of the list are not touched -- with
                // ~2.5 postings per sub-range the searches of a batch read every cache line of its terms' doc ids: 600 MB at 1024 queries)
                const uint64_t c = target / BM25_SKIP_DOCS;
                const int64_t v = lo + tab[c < skip_n ? (uint32_t)c : skip_n];
                bounds[i] = v;
                if (b > 0 && bounds_hi)
                    bounds_hi[i - 1] = v;
                continue;
            }
            const uint64_t c = target / BM25_SKIP_DOCS;
            const uint32_t c0 = c < skip_n ? (uint32_t)c : skip_n, c1 = c0 < skip_n ? c0 + 1 : skip_n;
            const int64_t base = lo;
            lo = base + tab[c0];
            hi = base + tab[c1];
        }
    }
    while (hi - lo > 7)
    {
        const int64_t step = (hi - lo) >> 3; // >= 1; probes lo + step .. lo + 7 step, all < hi
        uint32_t d[7];
#pragma unroll
        for (int u = 0; u < 7; u++)
            d[u] = a.doc_ids[lo + (u + 1) * step];
        int below = 0; // probes are ascending: the ones below the target form a prefix
#pragma unroll
        for (int u = 0; u < 7; u++)
            below += d[u] < target ? 1 : 0;
        // doc[lo + below step] < target (or below = 0) and doc[lo + (below + 1) step] >= target (or below = 7)
        const int64_t nlo = below ? lo + below * step + 1 : lo;
        const int64_t nhi = below < 7 ? lo + (below + 1) * step : hi;
        lo = nlo;
        hi = nhi;
    }
    {
        // at most 7 postings left: all of them side by side, the ones below the target are a prefix
        const int64_t n = hi - lo;
        int below = 0;
#pragma unroll
        for (int u = 0; u < 7; u++)
        {
            const uint32_t d = a.doc_ids[u < n ? lo + u : (lo > 0 ? lo - 1 : 0)]; // (a readable posting; its value is not used)
            below += u < n && d < target ? 1 : 0;
        }
        lo += below;
    }
    bounds[i] = lo;
    if (b > 0 && bounds_hi)
        bounds_hi[i - 1] = lo;
    }
    }
}

}
