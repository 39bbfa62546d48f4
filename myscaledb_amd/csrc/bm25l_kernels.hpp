// bm25l_kernels.hpp -- BM25 posting scorer for queries of at most FOUR terms (seam B, TantivyIndexStore.cpp:900-954), round 5.
//
// bm25r_kernel (bm25r_kernels.hpp) takes any query of up to 64 terms: lane t = term t, the window's slices are found with
// wave-wide sums and prefix scans, every record looks its term up.  Counted on the device (profiles/r05_bm25_pmc.txt) that is
// ~1850 wavefront instructions per 512-posting window -- 910 vector, 630 scalar, 100 LDS -- for 8 coalesced loads: the kernel
// is issue-bound at 0.18 of HBM.  Nearly every text query has one to four terms, and for those the per-term state fits in
// SCALAR registers:
//   * the window generator looks 16 steps AHEAD for all four terms with ONE vector load (lane = (step, term): the bounds of
//     the sub-ranges s + stride, s + 2 stride, ...), sums the four terms of a step with two DPP quad additions, and takes the
//     last step whose postings fit the window: a ballot and a count of trailing ones.  The load for the next window is issued
//     before the current one is scored.  `stride` follows the item's density (sparse queries take many sub-ranges per window);
//   * a ROW of 64 records that lies inside one term's slice -- all rows but the nt - 1 that hold a slice boundary -- is loaded
//     through a uniform base pointer and scored with a uniform weight: no per-record term arithmetic at all;
//   * conditions live as lane masks (scalar register pairs), the shared-document filter ORs unconditionally (a zero bit for a
//     dead lane, `old & bit` into the dup word), so that no execution mask is saved or restored per row;
//   * the flagged records get the second look of bm25r_kernel (a hash of the whole document id); what is flagged twice is
//     resolved pairwise.
// Same windows-of-whole-sub-ranges contract, same hashed seen / dup filter, same staged EMIT appends and TOPK floor as
// bm25r_kernel: the sums are the dense accumulator's additions in query-term order, bit for bit.
#pragma once
#include "bm25r_kernels.hpp"

#pragma clang fp contract(off)

namespace msvs
{

constexpr uint32_t BL_SLOTS2 = 4096; // hash slots of the second look (a few dozen records per window reach it)
#ifndef MSVS_BL_WPS
#define MSVS_BL_WPS 4
#endif
constexpr uint32_t BL_WAVES_PER_SIMD = MSVS_BL_WPS; // resident wavefronts per SIMD the register budget is cut for (= workgroups per CU)
constexpr uint32_t BL_NT = 4; // terms per query this kernel takes (the look-ahead is 64 lanes = 16 steps x 4 terms)

typedef const __attribute__((address_space(1))) uint2 * bl_gptr_u2;

template <typename T>
__device__ __forceinline__ T bl_sel4(const T & a0, const T & a1, const T & a2, const T & a3, uint32_t t)
{
    return t == 0 ? a0 : t == 1 ? a1 : t == 2 ? a2 : a3;
}

template <int MODE, int R, int SLOTS>
__global__ __launch_bounds__(64 * BP_WAVES, BL_WAVES_PER_SIMD) void bm25l_kernel(const Bm25RParams ar)
{
    bm25_slot_signal(ar.w.p);
    constexpr uint32_t BMW = 2 * SLOTS / 32;
    __shared__ __attribute__((aligned(16))) uint32_t bm_s[BP_WAVES][BMW]; // word pairs: seen | dup bits of 32 hash slots
    __shared__ __attribute__((aligned(16))) uint32_t bm2_s[BP_WAVES][2 * BL_SLOTS2 / 32]; // the second look's seen | dup
    __shared__ uint32_t fdoc_s[BP_WAVES][BP_CAP];     // the flagged records of a window, in flat (= term) order: document ...
    __shared__ float fsc_s[BP_WAVES][BP_CAP];         // ... partial score ...
    __shared__ uint8_t ft_s[BP_WAVES][BP_CAP];        // ... term
    __shared__ uint64_t stg_key_s[BP_WAVES][BP_STAGE]; // EMIT: keys waiting for their flush ...
    __shared__ uint32_t stg_q_s[BP_WAVES][BP_STAGE];   // ... and their queries
    __shared__ __attribute__((aligned(16))) uint4 tt_s[BP_WAVES][BL_NT]; // per term: record base (lo, hi), d, weight
    const Bm25WParams & a = ar.w;
    const Bm25Params & p = a.p;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t * const fdoc = fdoc_s[wave];
    float * const fsc = fsc_s[wave];
    uint8_t * const ft = ft_s[wave];
    uint32_t * const bm = bm_s[wave];
    uint32_t * const bm2 = bm2_s[wave];
    uint64_t * const stg_key = stg_key_s[wave];
    uint32_t * const stg_q = stg_q_s[wave];
    uint4 * const tt = tt_s[wave];
    for (uint32_t i = lane; i < BMW; i += 64)
        bm[i] = 0;
    for (uint32_t i = lane; i < 2 * BL_SLOTS2 / 32; i += 64)
        bm2[i] = 0;
    bp_wave_lds_fence(); // wave-private LDS only: no workgroup barrier in this kernel
    uint32_t vzero; // a zero the compiler cannot see through: the dead rows' loads stay vector loads (8 per window, counted waits)
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
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
    const uint32_t tl = lane & 3, jl = lane >> 2; // the look-ahead's lane layout: step jl of term tl
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
        const uint32_t nt = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.qoff[q + 1]) - j0; // <= BL_NT (the host routes the rest to bm25r_kernel)
        const uint32_t full = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)p.qfull[q]);
        float cut = 0.f;
        if (MODE == BM25_EMIT)
        {
            const uint64_t ck = p.cut_keys[(size_t)q * p.cut_m + p.cut_m - 1];
            const float c = ck == KEY_NONE ? 0.f : key_value<M_IP>(ck); // fewer than m sample hits: everything passes
            cut = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(c)));
        }
        WaveTopK<R> top;
        top.init();
        // ---- per-term state.  In lanes (step jl, term tl) for the generator, in scalar registers for the rows
        const bool has_l = tl < nt;
        const uint32_t jt = has_l ? j0 + tl : (nt ? j0 : 0u); // a term-less query reads flat term 0 (its lengths are forced to 0)
        const int64_t * const bnd_l = p.bounds + (size_t)jt * nb1;
        const int64_t B0_l = bnd_l[s_begin];
        const uint2 * const recp_l = ar.rec + B0_l; // the lane's term, first record of the item
        const uint32_t endrel_l = has_l ? (uint32_t)(bnd_l[s_end] - B0_l) : 0u;
        const float wl_ = has_l ? p.weight[jt] : 0.f;
        const uint32_t gl_ = has_l ? 1u << p.qgroup[jt] : 0u;
        uint64_t gpack = 0; // the four terms' token-group bits, 16 each (an array indexed by a lane's term would live in scratch)
#pragma unroll
        for (uint32_t t = 0; t < BL_NT; t++)
        {
            gpack |= (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)gl_, (int)t) << (16 * t);
        }
        auto group_of = [&](const uint32_t t) -> uint32_t { return (uint32_t)(gpack >> (16 * t)) & 0xffffu; };
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
        // ---- the window generator.  Sub-ranges per step from the item's own density: a full window is ~12 steps, the
        // look-ahead reaches 16 (an item that cannot fill a step falls back to single sub-ranges, and those to a split by id)
        uint32_t stride = 1;
        {
            uint32_t len = endrel_l > (1u << 24) ? (1u << 24) : endrel_l;
            len += dpp32<0xB1>(len);
            len += dpp32<0x4E>(len);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readfirstlane((int)len);
            if (total == 0 || s_begin >= s_end)
            {
                if (MODE == BM25_TOPK)
                    top.store(p.partial + ((size_t)slot * a.lists + ci) * p.kk, p.kk, lane);
                continue; // nothing of this query in the item
            }
            const uint64_t w = (uint64_t)(BP_CAP * 7 / 8) * (s_end - s_begin) / total; // sub-ranges per full window
            stride = w >= 18 ? (uint32_t)((w + 6) / 12) : 1u;
        }
        uint32_t s = s_begin;
        bool splitting = false;
        uint32_t d_lo = 0, d_end = 0, end_l = 0;
        uint32_t lo_l = 0; // the lane's term: first record not yet in a window (relative to the item's first)
        auto look_load = [&]() -> int64_t {
            const uint64_t idx = (uint64_t)s + (uint64_t)(jl + 1) * stride;
            return bnd_l[idx < s_end ? (uint32_t)idx : s_end];
        };
        // the look-ahead of the NEXT call, issued at ONE place (the end of next_window): a value defined at several places is a
        // copy, and a copy waits for its load.  The rare turns inside a call (an empty window, a stride that does not fit, the
        // end of a split) load what they need on the spot.
        int64_t braw = look_load();
        // -> false: the item is exhausted (tot = 0).  wlo_l / whi_l: the window's slice of the lane's term
        auto next_window = [&](uint32_t & wlo_l, uint32_t & whi_l, uint32_t & tot) -> bool {
            int64_t cur = braw;
            bool got = false;
            for (;;)
            {
                if (!splitting)
                {
                    if (s >= s_end)
                        break;
                    const uint32_t bl = has_l ? (uint32_t)(cur - B0_l) : 0u; // (waits for the look-ahead)
                    uint32_t tj = bl - lo_l;
                    tj = tj > (1u << 24) ? (1u << 24) : tj;
                    tj += dpp32<0xB1>(tj); // the four terms of the step
                    tj += dpp32<0x4E>(tj);
                    const uint64_t okm = __ballot(tj <= BP_CAP); // steps ascend, totals do not descend: a prefix
                    const uint32_t cnt = okm == ~0ull ? 16u : (uint32_t)__builtin_ctzll(~okm) >> 2;
                    if (cnt != 0)
                    {
                        const uint32_t jw = (cnt - 1) * 4;
                        tot = (uint32_t)__builtin_amdgcn_readlane((int)tj, (int)jw);
                        wlo_l = lo_l;
                        whi_l = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((jw + tl) * 4), (int)bl);
                        lo_l = whi_l;
                        const uint64_t ns = (uint64_t)s + (uint64_t)cnt * stride;
                        s = ns < s_end ? (uint32_t)ns : s_end;
                        if (tot == 0 || (a.dbg & 8))
                        {
                            cur = look_load();
                            continue;
                        }
                        got = true;
                        break;
                    }
                    if (stride > 1)
                    {
                        stride = 1; // a dense stretch of a sparse item: single sub-ranges from here on
                        cur = look_load();
                        continue;
                    }
                    splitting = true; // one sub-range over the cap: cut it by document id
                    d_lo = s * sub_docs;
                    d_end = (uint64_t)d_lo + sub_docs < p.num_docs ? d_lo + sub_docs : p.num_docs;
                    end_l = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(tl * 4), (int)bl); // step 0 at stride 1: the bound at s + 1
                }
                {
                    uint32_t d_hi = d_end;
                    uint32_t hi_l = end_l;
                    auto total_of = [&](const uint32_t h) -> uint32_t {
                        uint32_t len = h - lo_l;
                        len = len > (1u << 24) ? (1u << 24) : len;
                        len += dpp32<0xB1>(len);
                        len += dpp32<0x4E>(len);
                        return (uint32_t)__builtin_amdgcn_readfirstlane((int)len);
                    };
                    uint32_t t32 = total_of(hi_l);
                    while (t32 > BP_CAP && d_hi - d_lo > 1) // a single document holds <= nt <= 4 postings
                    {
                        d_hi = d_lo + (d_hi - d_lo) / 2;
                        uint32_t l2 = lo_l, h2 = hi_l; // the lane's term: first record at or past document d_hi
                        while (l2 < h2)
                        {
                            const uint32_t mid = l2 + ((h2 - l2) >> 1);
                            if (recp_l[mid].x < d_hi)
                                l2 = mid + 1;
                            else
                                h2 = mid;
                        }
                        hi_l = l2;
                        t32 = total_of(hi_l);
                    }
                    tot = t32;
                    wlo_l = lo_l;
                    whi_l = hi_l;
                    lo_l = hi_l;
                    d_lo = d_hi;
                    if (d_lo >= d_end)
                    {
                        splitting = false;
                        s += 1;
                    }
                    if (tot == 0 || (a.dbg & 8))
                    {
                        if (!splitting)
                            cur = look_load();
                        continue;
                    }
                    got = true;
                    break;
                }
            }
            if (!got)
            {
                tot = 0;
                wlo_l = whi_l = lo_l;
            }
            braw = look_load(); // (past the item's end: its last bound again)
            return got;
        };
        // ---- issue the 8 record loads of a window (always 8: counted waits), note every record's weight and term.  Straight-line:
        // record f of the window belongs to the LAST term whose slice starts at or before f (empty slices are skipped, their
        // starts coincide) and is record f + d_t of that term.  The term of all 8 records of a lane comes out of one packed
        // word: rows r >= ceil((pre_k - lane) / 64) are past the start of term k, +1 in the row's 2-bit field for k = 1, 2, 3;
        // (base, d, weight) of the four terms wait in a 64-byte LDS table, one 16-byte read per row.  Dead lanes (f >= tot) take
        // term 3 and re-read the window's last record -- or, behind an empty slice, the record in front of it: the record array
        // is padded by one record at either end (records_for).
        auto issue = [&](const uint32_t wlo_l, const uint32_t whi_l, const uint32_t tot, uint2 (&rb)[BP_RMAX], float (&wv)[BP_RMAX],
                         uint32_t & tpk) {
            const uint32_t len = whi_l - wlo_l;
            // inclusive sum over the terms of the step (the four lanes of a quad): +[t - 1] for t >= 1, then +[t - 2] for t >= 2
            // (the DPP moves run in EVERY lane, the selection follows: a move under a condition reads switched-off lanes)
            const uint32_t sh1 = dpp32<0x90>(len); // quad_perm [0, 0, 1, 2]
            uint32_t inc = len + (tl >= 1 ? sh1 : 0u);
            const uint32_t sh2 = dpp32<0x40>(inc); // quad_perm [0, 0, 0, 1]
            inc += tl >= 2 ? sh2 : 0u;
            const uint32_t d_l = wlo_l - (inc - len);
            tt[tl] = make_uint4((uint32_t)(uintptr_t)recp_l, (uint32_t)((uintptr_t)recp_l >> 32), d_l, __float_as_uint(wl_));
            const uint32_t pre1 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 0), pre2 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 1),
                           pre3 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 2);
            auto rows_from = [&](const uint32_t pre) -> uint32_t { // 0x5555 << 2 * (first row of the lane at or past `pre`), 16 bits
                const int32_t r0 = ((int32_t)(pre - lane) + 63) >> 6;
                const uint32_t rc = (uint32_t)(r0 < 0 ? 0 : r0 > 8 ? 8 : r0);
                return (0x5555u << (2 * rc)) & 0xffffu;
            };
            tpk = rows_from(pre1) + rows_from(pre2) + rows_from(pre3); // fields never carry: at most 3 each
            const uint32_t last = tot ? tot - 1 : 0u;
#pragma unroll
            for (uint32_t h = 0; h < BP_RMAX; h += 4) // four table reads side by side (one wait), then their four loads
            {
                uint4 e_r[4];
#pragma unroll
                for (uint32_t r = 0; r < 4; r++)
                    e_r[r] = tt[(tpk >> (2 * (h + r))) & 3u];
#pragma unroll
                for (uint32_t r = 0; r < 4; r++)
                {
                    const uint32_t f = min((h + r) * 64 + lane, last);
                    // (the pointer comes out of LDS as two integers: said to be GLOBAL, or the load is a flat one -- which waits for everything)
                    const bl_gptr_u2 base = reinterpret_cast<bl_gptr_u2>((uintptr_t)e_r[r].y << 32 | e_r[r].x);
                    const uint64_t rv = *reinterpret_cast<const __attribute__((address_space(1))) uint64_t *>(base + (int64_t)(int32_t)(f + e_r[r].z));
                    rb[h + r] = make_uint2((uint32_t)rv, (uint32_t)(rv >> 32));
                    wv[h + r] = __uint_as_float(e_r[r].w);
                }
            }
        };
        // ---- score a window whose records have arrived
        auto process = [&](const uint32_t tot, const uint2 (&rb)[BP_RMAX], const float (&wv)[BP_RMAX], const uint32_t tpk) {
            const uint32_t nr = (tot + 63) >> 6;
            float s_r[BP_RMAX];
#pragma unroll
            for (uint32_t r = 0; r < BP_RMAX; r++)
                s_r[r] = __fmul_rn(wv[r], __uint_as_float(rb[r].y));
            // which records share their document with another record of the window?  A hashed bitmap says "maybe": the first
            // record of a slot sets `seen`, every later one sets `dup`; a record whose slot is not in `dup` is the only posting
            // of its document in the window -- owner, score = its own partial.
            bool fl_r[BP_RMAX];
#pragma unroll
            for (uint32_t r = 0; r < BP_RMAX; r++)
                fl_r[r] = false;
            if (nt > 1 && !(a.dbg & 1))
            {
                // the eight rows side by side in every phase (one wait per phase, not per row); a dead lane ORs nothing.  `seen` and
                // `dup` are separate arrays (a word pair per slot group would use every other LDS bank only); the dup OR runs for
                // the few lanes that found their slot taken (an unconditional one doubled the atomic traffic: measured slower);
                // both arrays are cleared as a whole, 2 x 16 bytes per lane, not slot by slot.
                uint32_t bit_r[BP_RMAX], wa_r[BP_RMAX], old_r[BP_RMAX];
#pragma unroll
                for (uint32_t r = 0; r < BP_RMAX; r++)
                {
                    const uint32_t hs = rb[r].x & (SLOTS - 1);
                    wa_r[r] = hs >> 5;
                    bit_r[r] = r * 64 + lane < tot ? 1u << (hs & 31) : 0u;
                }
#pragma unroll
                for (uint32_t r = 0; r < BP_RMAX; r++)
                    old_r[r] = atomicOr(&bm[wa_r[r]], bit_r[r]);
#pragma unroll
                for (uint32_t r = 0; r < BP_RMAX; r++)
                    if (old_r[r] & bit_r[r])
                        atomicOr(&bm[SLOTS / 32 + wa_r[r]], bit_r[r]);
                bp_wave_lds_fence();
#pragma unroll
                for (uint32_t r = 0; r < BP_RMAX; r++)
                    old_r[r] = bm[SLOTS / 32 + wa_r[r]];
#pragma unroll
                for (uint32_t r = 0; r < BP_RMAX; r++)
                    fl_r[r] = (old_r[r] & bit_r[r]) != 0 && !(a.dbg & 16);
                bp_wave_lds_fence();
#pragma unroll
                for (uint32_t i = 0; i < BMW / 256; i++)
                    *reinterpret_cast<uint4 *>(&bm[(i * 64 + lane) * 4]) = make_uint4(0u, 0u, 0u, 0u);
            }
            // second look at the flagged few, still in their registers: a DIFFERENT hash (of the whole document id) through a small
            // bitmap pair of its own, only the flagged lanes switched on.  Two records of one document meet again whatever the hash;
            // the first filter's false alarms -- documents that only share their low bits, ~6 % of a window that spans 4 x SLOTS
            // documents -- do not, and leave with their own partial like the unshared ones.  What is flagged twice (true shares; a
            // false alarm per ~7 windows) goes to LDS and is resolved pairwise below.
            {
                bool anyf = false;
#pragma unroll
                for (uint32_t r = 0; r < BP_RMAX; r++)
                    anyf = anyf || fl_r[r];
                if (__ballot(anyf) && !(a.dbg & 2))
                {
                    uint32_t bit2_r[BP_RMAX], wa2_r[BP_RMAX], got_r[BP_RMAX];
#pragma unroll
                    for (uint32_t r = 0; r < BP_RMAX; r++)
                    {
                        const uint32_t h2 = (rb[r].x * 0x9E3779B1u) >> (32 - BR_LOG2(BL_SLOTS2));
                        wa2_r[r] = h2 >> 5;
                        bit2_r[r] = fl_r[r] ? 1u << (h2 & 31) : 0u;
                        got_r[r] = 0;
                    }
                    // (every phase issues its eight operations before the first result is looked at: a wait per phase, not per row)
#pragma unroll
                    for (uint32_t r = 0; r < BP_RMAX; r++)
                        if (fl_r[r])
                            got_r[r] = atomicOr(&bm2[wa2_r[r]], bit2_r[r]);
#pragma unroll
                    for (uint32_t r = 0; r < BP_RMAX; r++)
                        if (got_r[r] & bit2_r[r])
                            atomicOr(&bm2[BL_SLOTS2 / 32 + wa2_r[r]], bit2_r[r]);
                    bp_wave_lds_fence();
#pragma unroll
                    for (uint32_t r = 0; r < BP_RMAX; r++)
                        got_r[r] = 0;
#pragma unroll
                    for (uint32_t r = 0; r < BP_RMAX; r++)
                        if (fl_r[r])
                            got_r[r] = bm2[BL_SLOTS2 / 32 + wa2_r[r]];
#pragma unroll
                    for (uint32_t r = 0; r < BP_RMAX; r++)
                        fl_r[r] = (got_r[r] & bit2_r[r]) != 0;
                    bp_wave_lds_fence();
#pragma unroll
                    for (uint32_t i = 0; i < 2 * BL_SLOTS2 / 32 / 256; i++)
                        *reinterpret_cast<uint4 *>(&bm2[(i * 64 + lane) * 4]) = make_uint4(0u, 0u, 0u, 0u);
                }
            }
            // the unshared records leave at once.  EMIT: few records of a window pass the cut, most rows have none
            bool pass_r[BP_RMAX];
            bool any = false;
#pragma unroll
            for (uint32_t r = 0; r < BP_RMAX; r++)
            {
                bool ok = !(a.dbg & 4) && r * 64 + lane < tot && !fl_r[r] && (MODE != BM25_EMIT || s_r[r] >= cut);
                if (!p.operator_or)
                {
                    const uint32_t t = (tpk >> (2 * r)) & 3u;
                    ok = ok && group_of(t) == full;
                }
                if (MODE == BM25_TOPK && ok && p.alive) // (EMIT tests the few records that pass the cut below)
                    ok = rb[r].x < p.nbits && ((p.alive[rb[r].x >> 6] >> (rb[r].x & 63)) & 1);
                pass_r[r] = ok;
                any = any || ok;
            }
            // TOPK, the item's list not full yet: the kk-th largest of the 64 lanes' BEST scores is a floor -- kk records at or
            // above it exist -- and what lies below it cannot be among the window's kk best
            if (MODE == BM25_TOPK && top.thr == KEY_NONE)
            {
                float best = -1.f; // scores are >= 0
#pragma unroll
                for (uint32_t r = 0; r < BP_RMAX; r++)
                    if (pass_r[r])
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
                        pass_r[r] = pass_r[r] && !(s_r[r] < floor_s);
                }
            }
            if (MODE == BM25_TOPK || __ballot(any))
            {
#pragma unroll
                for (uint32_t r = 0; r < BP_RMAX; r++)
                {
                    if (r >= nr)
                        break;
                    bool ok = pass_r[r];
                    if (MODE == BM25_EMIT && !__ballot(ok))
                        continue;
                    const uint32_t docid = rb[r].x;
                    if (MODE == BM25_EMIT && ok && p.alive)
                        ok = docid < p.nbits && ((p.alive[docid >> 6] >> (docid & 63)) & 1);
                    out_one(ok, make_key<M_IP>(s_r[r], docid));
                }
            }
            // ---- the records flagged twice: compacted to LDS in flat order (ascending term) and resolved among themselves
            bool anyf = false;
#pragma unroll
            for (uint32_t r = 0; r < BP_RMAX; r++)
                anyf = anyf || fl_r[r];
            if (!__ballot(anyf))
                return;
            uint32_t nfl = 0;
            bp_wave_lds_fence(); // the previous window's resolution is done with the lists
#pragma unroll
            for (uint32_t r = 0; r < BP_RMAX; r++)
            {
                if (r >= nr)
                    break;
                const bool fl = fl_r[r];
                const uint64_t fm = __ballot(fl);
                if (fl)
                {
                    const uint32_t at = nfl + __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                    fdoc[at] = rb[r].x;
                    fsc[at] = s_r[r];
                    ft[at] = (uint8_t)((tpk >> (2 * r)) & 3u);
                }
                nfl += (uint32_t)__popcll(fm);
            }
            bp_wave_lds_fence();
            for (uint32_t i0 = 0; i0 < nfl; i0 += 64)
            {
                const uint32_t me = i0 + lane;
                const bool have = me < nfl;
                const uint32_t mi = have ? me : 0u;
                const uint32_t docid = fdoc[mi];
                float acc = fsc[mi];
                uint32_t mask = group_of((uint32_t)ft[mi]);
                bool dead = !have;
                for (uint32_t j = 0; j < nfl; j++)
                {
                    const uint32_t dj = fdoc[j]; // the same address in every lane: a broadcast read
                    const bool match = have && dj == docid && j != me;
                    if (!__ballot(match))
                        continue;
                    const float sj = fsc[j];
                    const uint32_t bj = group_of((uint32_t)ft[j]);
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
                bool ok = !(a.dbg & 4) && !dead && (p.operator_or || mask == full) && (MODE != BM25_EMIT || acc >= cut);
                if (ok && p.alive)
                    ok = docid < p.nbits && ((p.alive[docid >> 6] >> (docid & 63)) & 1);
                out_one(ok, make_key<M_IP>(acc, docid));
            }
        };
        // ---- two windows in flight, named register sets (a copy of a register waits for its load)
        uint32_t loA, hiA, loB, hiB, totA, totB, tpA, tpB;
        uint2 rbA[BP_RMAX], rbB[BP_RMAX];
        float wvA[BP_RMAX], wvB[BP_RMAX];
        bool more = next_window(loA, hiA, totA);
        issue(loA, hiA, totA, rbA, wvA, tpA);
        while (more)
        {
            more = next_window(loB, hiB, totB);
            issue(loB, hiB, totB, rbB, wvB, tpB);
            process(totA, rbA, wvA, tpA);
            if (!more)
                break;
            more = next_window(loA, hiA, totA);
            issue(loA, hiA, totA, rbA, wvA, tpA);
            process(totB, rbB, wvB, tpB);
        }
        if (MODE == BM25_TOPK)
            top.store(p.partial + ((size_t)slot * a.lists + ci) * p.kk, p.kk, lane);
    }
    if (MODE == BM25_EMIT)
        flush();
}

}
