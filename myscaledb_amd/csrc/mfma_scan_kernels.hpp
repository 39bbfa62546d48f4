// mfma_scan_kernels.hpp -- matrix-core candidate pass of batched searches + canonical re-rank.
//
// When many queries of a batch scan the same rows (>= ~4 per IVF list, or a batch against a whole table) the canonical
// scan of scan_kernels.hpp is VALU-bound: the parity contract forbids fma, so every (query, row, element) costs three
// separately rounded VALU ops.  The query-tile x row-tile inner products are a GEMM, so this path runs them on the
// matrix cores as a PRE-FILTER and keeps the returned numbers canonical:
//
//   1. ivf_mfma_scan_big_kernel : approximate distance a(q,x) = |x|^2 - 2<q,x> + |q|^2 (L2) or <q,x> (IP) for every
//      (query, probed row), inner products in split bf16 on v_mfma_f32_32x32x16_bf16; every (query, 128-row slice)
//      appends its <= 16 best rows by a() to the query's candidate buffer (keys carry row POSITIONS).
//   2. cand_select_kernel : per query, the KC (32 for k <= 12, else 64) best candidates, ascending, and
//      bound = the smallest key any slice may have cut.
//   3. ivf_rerank_kernel  : canonical (bit-exact, scan_kernels.hpp arithmetic) distance of the KC candidates, exact
//      top-k among them, and a CERTIFICATE: every row that is not a candidate has a() >= min(a_KC, bound), and
//      |a - canonical| <= eps for a rigorous rounding-error bound eps, so if min(a_KC, bound) - eps > e_k (the exact
//      k-th distance found) no excluded row can enter the top-k and the result equals the exhaustive one.
//   4. queries whose certificate fails (near-ties wider than the candidate lists, huge norms, NaN, an overflowed
//      buffer) are re-run through the canonical one-query-per-block scan (ivf_scan_subset_kernel /
//      ivf_merge_subset_kernel), stream-ordered, no host sync.
//
// The result is therefore ALWAYS identical to the canonical scan; eps only decides how often step 4 has work.
// The same four steps serve the IVF list scan, the coarse quantiser (centroid table = one list every query probes,
// result = the probe lists) and FLAT indexes (single_list_plan_kernel).
// Error bound (u = 2^-24), relative to |x||q| <= (|x|+|q|)^2 / 4 with |x| <= sqrt(max row norm of the table):
//   c_dot   split bf16: x = xh + xl + rx with |xl| <= 2^-8 |x|, |rx| <= 2^-16 |x| (same for q); the products kept are
//           xh*qh + xh*ql + xl*qh (each exact in f32), the dropped xl*ql, xh*rq, rx*q sum to <= 3.1 * 2^-16 |x||q|;
//           the MFMA accumulation of n = 3d terms, whatever its internal order and rounding mode, stays within
//           n * 2^-23 |x||q|;
//   c_norm  the fma-accumulated f32 norms: (d + 8) u;
//   c_canon the canonical result itself (d separately rounded terms, depth <= 18, + the sub/mul of L2): 32u;
//   L2:  eps = 2 c_dot |x||q| + c_norm (|x|^2 + |q|^2) + c_canon (|x|+|q|)^2 (+ 2 roundings of the final a);
//   IP:  eps = (c_dot + c_canon) |x||q|;   everything times 1.05.
#pragma once

#include "scan_kernels.hpp"

namespace msvs
{

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MF_LDS = 128 + 4; // LDS row stride (floats) of the per-query distance tile

/// out[r] = |X[r]|^2 (fma, 16 lanes per row; NOT the canonical order: only feeds the approximate pass and its error
/// bound).  max_bits (nullable): running maximum of the float bit patterns (norms are >= 0; NaN compares largest).
static __global__ __launch_bounds__(BLOCK) void row_sqnorm16_kernel(const float4 * X, float * out, size_t n,
                                                                     uint32_t ld4, uint32_t * max_bits)
{
    const size_t r = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const uint32_t g = threadIdx.x & 15;
    float s = 0.f;
    if (r < n)
        for (uint32_t c = g; c < ld4; c += 16)
        {
            const float4 v = X[r * ld4 + c];
            s = fmaf(v.x, v.x, s);
            s = fmaf(v.y, v.y, s);
            s = fmaf(v.z, v.z, s);
            s = fmaf(v.w, v.w, s);
        }
    s = row16_tree_sum(s);
    if (r < n && g == 0)
    {
        out[r] = s;
        if (max_bits)
            atomicMax(max_bits, __float_as_uint(s));
    }
}

/// One (query, float4 column) update of the canonical accumulators (same arithmetic as scan_rows).
template <int METRIC>
__device__ __forceinline__ void canonical_update(float4 & s, const float4 q, const float4 y)
{
    if (METRIC == M_L2)
    {
        float dx = __fsub_rn(q.x, y.x), dy = __fsub_rn(q.y, y.y), dz = __fsub_rn(q.z, y.z), dw = __fsub_rn(q.w, y.w);
        s.x = __fadd_rn(s.x, __fmul_rn(dx, dx));
        s.y = __fadd_rn(s.y, __fmul_rn(dy, dy));
        s.z = __fadd_rn(s.z, __fmul_rn(dz, dz));
        s.w = __fadd_rn(s.w, __fmul_rn(dw, dw));
    }
    else
    {
        s.x = __fadd_rn(s.x, __fmul_rn(q.x, y.x));
        s.y = __fadd_rn(s.y, __fmul_rn(q.y, y.y));
        s.z = __fadd_rn(s.z, __fmul_rn(q.z, y.z));
        s.w = __fadd_rn(s.w, __fmul_rn(q.w, y.w));
    }
}

// ------------------------------------------------------------------------------------------ the candidate pass
//
// Design of ivf_mfma_scan_big_kernel (measured alternatives: profiles/r01_candidate_pass_notes.txt):
//   * a 128-row x 128-query tile per workgroup: each of the 4 wavefronts owns 32 rows x 128 queries = 4 MFMA
//     accumulators, so one LDS read of its rows feeds up to 4 products; 32-query column blocks without queries are
//     skipped.  A list is read ONCE for up to 128 probing queries.  71 KB of LDS -> two workgroups per CU, one
//     selecting while the other multiplies;
//   * SPLIT BF16 ("bf16x3"): every f32 operand is staged in LDS as hi = bf16(v) and lo = bf16(v - hi) (v - hi is
//     exact in f32), and <x,q> ~= <xh,qh> + <xh,ql> + <xl,qh> on v_mfma_f32_32x32x16_bf16 with f32 accumulation:
//     6 bf16 MFMAs (32 cycles each) per 32 reduction elements where the f32 MFMA needs 16 x 64 cycles, i.e. 5.3x less
//     matrix-core time, which turns the kernel from MFMA-bound into memory-bound;
//   * no per-query list: every (query, 128-row slice) appends its own 16 best rows (radix select over the wavefront's
//     128 keys by ballots, then a 16-lane DPP bitonic sort) to the query's candidate buffer in global memory (atomic
//     cursor).  A slice can only hide a row from the result if 16 better rows sit in the same slice; the re-rank
//     certifies against bound = min(KC-th candidate, 16th key of every FULL slice list) -- the second term is
//     exactly qthr below;
//   * ONE number per query shared across the whole grid: qthr[q] = the smallest 16th-key distance any full slice list
//     of the query has emitted so far (atomicMin).  A row whose approximate distance is not below it can be dropped
//     at once -- it is no better than a key the bound already accounts for -- so after the first few slices of a
//     query almost every later slice has fewer than 16 survivors and skips the selection altogether.  Which rows get
//     dropped depends on timing; the certified result does not.

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

/// Two f32 -> two bf16 (round to nearest even) packed in one dword, `a` in the low half.
__device__ __forceinline__ uint32_t pack_bf16(const float a, const float b)
{
    bf16x2 t;
    t[0] = (__bf16)a;
    t[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, t);
}

constexpr int BG_ROWS = 128;
constexpr int BG_TQ = 128;
constexpr int BG_SLICE_K = 16; // keys emitted per (query, slice)
constexpr int BG_KC = 32;      // candidates per query after the merge

/// The 16 smallest of the wavefront's 128 keys (k0 of every lane, then k1; KEY_NONE = absent) in ascending order in
/// lanes 0..15 of the return value (KEY_NONE padded; other lanes KEY_NONE).  Keys are unique except KEY_NONE.
/// scratch: 16 u64 of LDS private to the wavefront.
__device__ __forceinline__ uint64_t wave_select16(const uint64_t k0, const uint64_t k1, volatile uint64_t * scratch,
                                                  const uint32_t lane)
{
    uint64_t alive0 = __ballot(k0 != KEY_NONE), alive1 = __ballot(k1 != KEY_NONE);
    uint64_t sel0 = 0, sel1 = 0;
    uint32_t need = BG_SLICE_K;
    uint32_t cnt = __popcll(alive0) + __popcll(alive1);
    if (cnt <= need)
    {
        sel0 = alive0;
        sel1 = alive1;
    }
    else
    {
        // MSB-first radix select of the need-th smallest key: `alive` = keys still tied with it on the bits seen so far
        const uint32_t h0 = (uint32_t)(k0 >> 32), l0 = (uint32_t)k0, h1 = (uint32_t)(k1 >> 32), l1 = (uint32_t)k1;
        for (int bit = 63; bit >= 0; bit--)
        {
            const uint32_t sh = bit & 31;
            const uint32_t w0 = bit >= 32 ? h0 : l0, w1 = bit >= 32 ? h1 : l1;
            const uint64_t one0 = __ballot((w0 >> sh) & 1u), one1 = __ballot((w1 >> sh) & 1u);
            const uint64_t z0 = alive0 & ~one0, z1 = alive1 & ~one1;
            const uint32_t c = __popcll(z0) + __popcll(z1);
            if (c >= need)
            {
                alive0 = z0;
                alive1 = z1;
                cnt = c;
            }
            else
            {
                need -= c; // every tied key with a 0 here is smaller: selected
                sel0 |= z0;
                sel1 |= z1;
                alive0 &= one0;
                alive1 &= one1;
                cnt -= c;
            }
            if (cnt == need)
            {
                sel0 |= alive0;
                sel1 |= alive1;
                break;
            }
        }
    }
    // compact the selected keys into 16 LDS slots, reload one per lane
    if (lane < (uint32_t)BG_SLICE_K)
        scratch[lane] = KEY_NONE;
    const uint32_t s0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(sel0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sel0, 0u));
    const uint32_t s1 = __popcll(sel0)
        + __builtin_amdgcn_mbcnt_hi((uint32_t)(sel1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sel1, 0u));
    if ((sel0 >> lane) & 1)
        scratch[s0] = k0;
    if ((sel1 >> lane) & 1)
        scratch[s1] = k1;
    __threadfence_block();
    uint64_t v = lane < (uint32_t)BG_SLICE_K ? scratch[lane] : KEY_NONE;
    // bitonic sort of the 16 lanes of DPP row 0 (flip + half-cleaner form; every exchange is a DPP move)
    auto cx = [&](const uint64_t partner, const bool keep_min) {
        const bool take = keep_min ? partner < v : partner > v;
        v = take ? partner : v;
    };
    cx(dpp64<0xB1>(v), !(lane & 1));  // blocks of 2: lane ^ 1
    cx(dpp64<0x1B>(v), !(lane & 2));  // blocks of 4: lane ^ 3 (quad reversed)
    cx(dpp64<0xB1>(v), !(lane & 1));
    cx(dpp64<0x141>(v), !(lane & 4)); // blocks of 8: lane ^ 7 (row_half_mirror)
    cx(dpp64<0x4E>(v), !(lane & 2));  //              lane ^ 2
    cx(dpp64<0xB1>(v), !(lane & 1));
    cx(dpp64<0x140>(v), !(lane & 8)); // blocks of 16: lane ^ 15 (row_mirror)
    {
        const uint64_t up = dpp64<0x104>(v), dn = dpp64<0x114>(v); // row_shl:4 = lane + 4, row_shr:4 = lane - 4
        cx((lane & 4) ? dn : up, !(lane & 4));                      //               lane ^ 4
    }
    cx(dpp64<0x4E>(v), !(lane & 2));
    cx(dpp64<0xB1>(v), !(lane & 1));
    return v;
}

/// Work item = (list, tile of <= 128 * NQG probing queries, segment of a.rows_per_block rows (a multiple of 128)); plan
/// built with T = BG_TQ * NQG.  Every (query, 128-row slice) appends its <= 16 best (approximate key, row position) at
/// a.partial[q * a.cand_cap + atomicAdd(a.qcnt[q], n)].  a.cand_cap may be smaller than 16 * (slices the query can
/// meet): keys past the capacity are dropped, qcnt keeps counting, and cand_select_kernel turns qcnt > cap into a
/// failed certificate (canonical fallback) -- with the cut at work a query appends a few hundred keys.
///
/// NQG = 1: 4 wavefronts, 128-query tiles, 71 KB of LDS, two workgroups per CU.
/// NQG = 2: 8 wavefronts, 256-query tiles: wavefronts 0-3 and 4-7 multiply the SAME staged rows by the first / second
///          128 queries, so a list probed by up to 256 queries of the batch is read once (at ~200 queries per list the
///          128-query tiles read it 1.9 times); 102 KB of LDS, one workgroup per CU, same 8 wavefronts per CU.
///
/// PHASE only names the launch (0: sample phase / table passes, 1: main phase of the list scan) so that profilers list the
/// two phases of a search step as two kernels; the code is the same.
template <int METRIC, int NQG, int PHASE>
__global__ __launch_bounds__(BLOCK * NQG) __attribute__((amdgpu_waves_per_eu(2, 2))) void ivf_mfma_scan_big_kernel(
    const ScanParams a)
{
    constexpr uint32_t THREADS = BLOCK * NQG, NWAVE = 4 * NQG, TQ = BG_TQ * NQG;
    constexpr uint32_t RS = THREADS / 8;       // rows covered by one pass of the loader (8 threads per 128-byte row)
    constexpr int XPT = BG_ROWS / RS;          // row float4 per thread and step: 4 (NQG = 1) / 2 (NQG = 2)
    constexpr int QPT = TQ / RS;               // query float4 per thread and step: 4
    // Operand stage: 2 buffers x {rows hi, rows lo: 128 x 64 B; queries hi, queries lo: TQ x 64 B} (32 bf16 of the
    // reduction step per row).  A row's four 16-byte chunks are XOR-swizzled by (row >> 2) & 3, which makes both the
    // 8-byte staging writes and the 16-lane ds_read_b128 operand reads bank-conflict free without padding.
    constexpr uint32_t XPLANE = BG_ROWS * 64, QPLANE = TQ * 64, BUF = 2 * XPLANE + 2 * QPLANE;
    constexpr uint32_t STAGE_BYTES = 2 * BUF > BG_TQ * MF_LDS * 4 ? 2 * BUF : BG_TQ * MF_LDS * 4;
    __shared__ __attribute__((aligned(16))) unsigned char stage[STAGE_BYTES];
    __shared__ uint64_t sel_s[NWAVE][BG_SLICE_K];
    __shared__ uint32_t thr_s[TQ];
    __shared__ float xn_s[BG_ROWS];
    __shared__ float qn_s[TQ];
    __shared__ uint32_t qrow_s[TQ];
    __shared__ uint32_t qpair_s[TQ];
    float * const Ss = reinterpret_cast<float *>(stage); // distance tile [128 queries][128 rows (+4)]: reuses the stage

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r32 = lane & 31, h = lane >> 5;
    const uint32_t wr = wave & 3, qg = wave >> 2; // wavefront tile: rows 32*wr .. +31 x queries 128*qg .. +127
    const uint32_t ld4 = a.ld4;
    const uint32_t nk = (ld4 + 7) / 8;
    const uint32_t lc = tid & 7, lr = tid >> 3; // loader: float4 column lc of rows / queries lr + RS*i
    const uint32_t total = a.work_off[a.nlist];
    const uint32_t per_xcd = (total + 7) / 8;
    for (uint32_t s = blockIdx.x; s < 8 * per_xcd; s += gridDim.x)
    {
        const uint32_t w = a.xcd_order ? (s & 7) * per_xcd + (s >> 3) : s;
        if (w >= total || (a.xcd_order && (s >> 3) >= per_xcd))
            continue;
        uint32_t lo = 0, hi = a.nlist;
        while (hi - lo > 1)
        {
            uint32_t mid = (lo + hi) >> 1;
            if (a.work_off[mid] <= w)
                lo = mid;
            else
                hi = mid;
        }
        const uint32_t l = lo;
        const int64_t lbeg = a.list_off[l], lend = a.list_end ? a.list_end[l] : a.list_off[l + 1];
        const uint32_t local = w - a.work_off[l];
        const uint32_t pe = a.pair_off[l + 1];
        const uint32_t tq = a.tile_q ? a.tile_q : TQ; // smaller tiles = more work items for small tables / batches
        const uint32_t ntile = (pe - a.pair_off[l] + tq - 1) / tq;
        const uint32_t seg = local / ntile, tile = local - seg * ntile;
        const uint32_t pb = a.pair_off[l] + tile * tq;
        const uint32_t nvalid = pe - pb < tq ? pe - pb : tq;
        const uint32_t ncb = (nvalid + 31) >> 5; // 32-query column blocks in use (of 4 * NQG)
        const int64_t rb = lbeg + (int64_t)seg * a.rows_per_block;
        const int64_t re = rb + a.rows_per_block < lend ? rb + a.rows_per_block : lend;

        __syncthreads(); // the previous work item is done with the tables and the stage
        if (tid < TQ)
        {
            const uint32_t pi = pb + tid < pe ? pb + tid : pe - 1; // short tiles repeat their last pair (never selected)
            const uint32_t qp = a.pairs[pi];
            const uint32_t q = qp / a.nprobe;
            qrow_s[tid] = q;
            qpair_s[tid] = qp;
            qn_s[tid] = METRIC == M_L2 ? a.qnorm[q] : 0.f;
        }
        if (METRIC == M_L2 && tid < BG_ROWS)
            xn_s[tid] = a.xnorm[rb + tid < re ? rb + tid : re - 1];
        __syncthreads();

        const float4 * qsrc[QPT];
        bool qld[QPT];
#pragma unroll
        for (int i = 0; i < QPT; i++)
        {
            qsrc[i] = a.Qsplit + (size_t)qrow_s[lr + RS * i] * nk * 8; // 8 x 16 B per query and reduction step
            qld[i] = (lr + RS * i) / 32 < ncb; // the column block of query row lr + RS*i is in use
        }
        float qn[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
            qn[i] = qn_s[128 * qg + 32 * i + r32];
        // this wavefront's column blocks (global index 4*qg + i) in use
        const uint32_t mycb = ncb > 4 * qg ? (ncb - 4 * qg < 4 ? ncb - 4 * qg : 4) : 0;

        int64_t pf_sub = rb; // prefetch position of the software pipeline over (sub-tile, reduction step)
        uint32_t pf_ki = 0;
        const float4 * xsrc[XPT];
        float4 px[XPT], pq[QPT];
        auto set_rows = [&](int64_t sub) {
#pragma unroll
            for (int i = 0; i < XPT; i++)
            {
                int64_t row = sub + lr + RS * i;
                if (row >= re)
                    row = re - 1; // rows past the segment repeat its last row; they are never offered
                xsrc[i] = a.Y + (size_t)row * ld4;
            }
        };
        auto gload = [&]() {
            const uint32_t c = pf_ki * 8 + lc;
            const bool in = c < ld4;
#pragma unroll
            for (int i = 0; i < XPT; i++)
                px[i] = in ? xsrc[i][c] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < QPT; i++)
                pq[i] = qld[i] ? qsrc[i][c] : make_float4(0.f, 0.f, 0.f, 0.f); // already split and zero padded
        };
        auto advance = [&]() {
            if (++pf_ki == nk)
            {
                pf_ki = 0;
                pf_sub += BG_ROWS;
                if (pf_sub < re)
                    set_rows(pf_sub);
            }
        };
        // float4 -> 4 bf16 hi + 4 bf16 lo, written to the swizzled position of (row, float4 column lc); the lo plane
        // follows the hi plane at `plane` bytes
        auto split_store = [&](unsigned char * hi_plane, const uint32_t plane, const uint32_t row, const float4 v) {
            const uint32_t h01 = pack_bf16(v.x, v.y), h23 = pack_bf16(v.z, v.w);
            const float fx = __uint_as_float(h01 << 16), fy = __uint_as_float(h01 & 0xffff0000u);
            const float fz = __uint_as_float(h23 << 16), fw = __uint_as_float(h23 & 0xffff0000u);
            const uint32_t l01 = pack_bf16(v.x - fx, v.y - fy), l23 = pack_bf16(v.z - fz, v.w - fw);
            const uint32_t off = row * 64 + ((((lc >> 1) ^ (row >> 2)) & 3) << 4) + ((lc & 1) << 3);
            *reinterpret_cast<uint2 *>(hi_plane + off) = make_uint2(h01, h23);
            *reinterpret_cast<uint2 *>(hi_plane + plane + off) = make_uint2(l01, l23);
        };
        auto sstore = [&](int buf) {
            unsigned char * base = stage + buf * BUF;
#pragma unroll
            for (int i = 0; i < XPT; i++)
                split_store(base, XPLANE, lr + RS * i, px[i]);
#pragma unroll
            for (int i = 0; i < QPT; i++)
                if (qld[i]) // thread lc carries chunk lc & 3 of plane lc >> 2 (hi / lo): a plain 16-byte copy
                {
                    const uint32_t row = lr + RS * i;
                    *reinterpret_cast<float4 *>(base + 2 * XPLANE + (lc >> 2) * QPLANE + row * 64
                                                + ((((lc & 3) ^ (row >> 2)) & 3) << 4))
                        = pq[i];
                }
        };
        set_rows(rb);
        gload();
        advance();
        sstore(0);
        __syncthreads();
        int cur = 0;

        for (int64_t sub = rb; sub < re; sub += BG_ROWS)
        {
            const bool has_next = sub + BG_ROWS < re;
            float xn_next = 0.f;
            if (METRIC == M_L2 && has_next && tid < BG_ROWS)
                xn_next = a.xnorm[sub + BG_ROWS + tid < re ? sub + BG_ROWS + tid : re - 1];
            if (tid < TQ)
                thr_s[tid] = a.qthr[qrow_s[tid]]; // read by the selection, at least one barrier from here
            bool okrow[2];
#pragma unroll
            for (int u = 0; u < 2; u++)
            {
                const int64_t row = sub + lane + 64 * u;
                bool ok = row < re;
                if (ok && a.alive)
                {
                    const uint32_t id = a.ids ? a.ids[row] : (uint32_t)row + a.id_base;
                    ok = id < a.nbits && ((a.alive[id >> 6] >> (id & 63)) & 1);
                }
                okrow[u] = ok;
            }

            f32x16 acc0, acc1, acc2, acc3; // one per 32-query column block of this wavefront's 128 queries
#pragma unroll
            for (int r = 0; r < 16; r++)
            {
                acc0[r] = 0.f;
                acc1[r] = 0.f;
                acc2[r] = 0.f;
                acc3[r] = 0.f;
            }
            for (uint32_t ki = 0; ki < nk; ki++)
            {
                const bool more = pf_sub < re;
                if (more)
                    gload();
                if (mycb)
                {
                    // lane (r32, h) feeds row / query r32 with reduction elements 16*j + 8*h .. +7 (chunk 2j + h) of
                    // each 32x32x16 product; both operands use the same positions, so the element order inside the
                    // instruction does not matter
                    const unsigned char * base = stage + cur * BUF;
                    const uint32_t sw = (r32 >> 2) & 3;
                    const uint32_t o0 = ((h ^ sw) & 3) << 4, o1 = (((2 + h) ^ sw) & 3) << 4;
                    const unsigned char * xa = base + (32 * wr + r32) * 64;
                    const unsigned char * qb = base + 2 * XPLANE + (128 * qg + r32) * 64;
                    const bf16x8 ah0 = *reinterpret_cast<const bf16x8 *>(xa + o0);
                    const bf16x8 ah1 = *reinterpret_cast<const bf16x8 *>(xa + o1);
                    const bf16x8 al0 = *reinterpret_cast<const bf16x8 *>(xa + XPLANE + o0);
                    const bf16x8 al1 = *reinterpret_cast<const bf16x8 *>(xa + XPLANE + o1);
                    auto block = [&](f32x16 & acc, const int cb) {
                        const unsigned char * q = qb + cb * 32 * 64;
                        const bf16x8 bh0 = *reinterpret_cast<const bf16x8 *>(q + o0);
                        const bf16x8 bh1 = *reinterpret_cast<const bf16x8 *>(q + o1);
                        const bf16x8 bl0 = *reinterpret_cast<const bf16x8 *>(q + QPLANE + o0);
                        const bf16x8 bl1 = *reinterpret_cast<const bf16x8 *>(q + QPLANE + o1);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh0, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl0, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh1, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl1, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh0, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh1, acc, 0, 0, 0);
                    };
                    block(acc0, 0);
                    if (mycb > 1)
                        block(acc1, 1);
                    if (mycb > 2)
                        block(acc2, 2);
                    if (mycb > 3)
                        block(acc3, 3);
                }
                if (ki + 1 < nk)
                {
                    sstore(cur ^ 1);
                    __syncthreads();
                    cur ^= 1;
                }
                if (more)
                    advance();
            }

            // one selection round per 128-query group: its wavefronts publish their distances, ALL wavefronts select
            auto put = [&](const f32x16 & acc, const float qnv, const uint32_t n0) {
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++)
                {
                    const uint32_t m0 = 32 * wr + 8 * g4 + 4 * h;
                    float4 o;
                    if (METRIC == M_L2)
                    {
                        o.x = fmaf(-2.f, acc[4 * g4 + 0], xn_s[m0 + 0]) + qnv;
                        o.y = fmaf(-2.f, acc[4 * g4 + 1], xn_s[m0 + 1]) + qnv;
                        o.z = fmaf(-2.f, acc[4 * g4 + 2], xn_s[m0 + 2]) + qnv;
                        o.w = fmaf(-2.f, acc[4 * g4 + 3], xn_s[m0 + 3]) + qnv;
                    }
                    else
                        o = make_float4(acc[4 * g4 + 0], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3]);
                    *reinterpret_cast<float4 *>(&Ss[(n0 + r32) * MF_LDS + m0]) = o;
                }
            };
#pragma unroll
            for (uint32_t g = 0; g < (uint32_t)NQG; g++)
            {
                const uint32_t gvalid = nvalid > 128 * g ? (nvalid - 128 * g < 128 ? nvalid - 128 * g : 128) : 0;
                if (gvalid == 0)
                    break; // uniform over the workgroup
                __syncthreads(); // operand stage (g = 0) / previous group's tile (g = 1) fully consumed
                if (qg == g)
                {
                    put(acc0, qn[0], 0);
                    if (mycb > 1)
                        put(acc1, qn[1], 32);
                    if (mycb > 2)
                        put(acc2, qn[2], 64);
                    if (mycb > 3)
                        put(acc3, qn[3], 96);
                }
                __syncthreads();
                for (uint32_t n = wave; n < gvalid; n += NWAVE)
                {
                    const uint32_t qi = 128 * g + n;
                    const uint32_t cut = thr_s[qi];
                    uint64_t key[2];
#pragma unroll
                    for (int u = 0; u < 2; u++)
                    {
                        const float v = Ss[n * MF_LDS + lane + 64 * u];
                        const uint64_t kk
                            = okrow[u] ? make_key<METRIC>(v, (uint32_t)(sub + lane + 64 * u)) : KEY_NONE;
                        key[u] = (uint32_t)(kk >> 32) < cut ? kk : KEY_NONE;
                    }
                    if (__ballot(key[0] != KEY_NONE) | __ballot(key[1] != KEY_NONE))
                    {
                        const uint64_t best = wave_select16(key[0], key[1], sel_s[wave], lane);
                        const uint32_t q = qrow_s[qi];
                        const uint32_t nsel = __popcll(__ballot(best != KEY_NONE)); // lanes 0 .. nsel-1, ascending
                        uint32_t pos = 0;
                        if (lane == 0)
                            pos = atomicAdd(&a.qcnt[q], nsel);
                        pos = __builtin_amdgcn_readfirstlane(pos);
                        if (lane < nsel && pos + lane < a.cand_cap)
                            a.partial[(size_t)q * a.cand_cap + pos + lane] = best;
                        if (lane == (uint32_t)BG_SLICE_K - 1 && best != KEY_NONE) // a full list: its last key cuts
                            atomicMin(&a.qthr[q], (uint32_t)(best >> 32));
                    }
                }
            }
            if (has_next)
            {
                __syncthreads(); // distance tile consumed: restage the first step of the next sub-tile
                sstore(0);
                if (METRIC == M_L2 && tid < BG_ROWS)
                    xn_s[tid] = xn_next;
                __syncthreads();
                cur = 0;
            }
        }
    }
}

/// Queries -> the split-bf16 layout the candidate pass stages verbatim: per query and 32-element reduction step 128 B =
/// {hi: 4 chunks of 8 bf16 | lo: 4 chunks of 8 bf16}, zero padded to whole steps (same bytes as the f32 rows; done once
/// per search instead of once per (work item, sub-tile)).  One thread per 16-byte chunk.
static __global__ void split_queries_kernel(const float4 * Q, uint32_t nq, uint32_t ld4, uint32_t nk, uint4 * out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nq * nk * 8)
        return;
    const uint32_t c = (uint32_t)(i & 7), ki = (uint32_t)((i >> 3) % nk), q = (uint32_t)((i >> 3) / nk);
    const uint32_t col = ki * 8 + (c & 3) * 2; // float4 column of the chunk's first 4 elements
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 v0 = col < ld4 ? Q[(size_t)q * ld4 + col] : zero, v1 = col + 1 < ld4 ? Q[(size_t)q * ld4 + col + 1] : zero;
    uint32_t h0 = pack_bf16(v0.x, v0.y), h1 = pack_bf16(v0.z, v0.w), h2 = pack_bf16(v1.x, v1.y), h3 = pack_bf16(v1.z, v1.w);
    if (c >> 2) // lo plane: bf16(v - hi), v - hi exact in f32
    {
        auto lo2 = [](const uint32_t h, const float a, const float b) {
            return pack_bf16(a - __uint_as_float(h << 16), b - __uint_as_float(h & 0xffff0000u));
        };
        h0 = lo2(h0, v0.x, v0.y);
        h1 = lo2(h1, v0.z, v0.w);
        h2 = lo2(h2, v1.x, v1.y);
        h3 = lo2(h3, v1.z, v1.w);
    }
    out[i] = make_uint4(h0, h1, h2, h3);
}

/// The one-list "plan" that lets the candidate pass run over a plain row table (the coarse quantiser's centroids, a
/// FLAT index): every query probes list 0 = rows [row_begin, row_end).
static __global__ void single_list_plan_kernel(uint32_t nq, uint32_t row_begin, uint32_t row_end,
                                               uint32_t rows_per_block, uint32_t tq, uint32_t * pairs,
                                               int32_t * probes0, int64_t * list_off, uint32_t * pair_off,
                                               uint32_t * work_off)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nq)
    {
        pairs[i] = i;
        probes0[i] = 0;
    }
    if (i == 0)
    {
        list_off[0] = row_begin;
        list_off[1] = row_end;
        pair_off[0] = 0;
        pair_off[1] = nq;
        work_off[0] = 0;
        work_off[1] = ((nq + tq - 1) / tq) * ((row_end - row_begin + rows_per_block - 1) / rows_per_block);
    }
}

/// Long tables: after the candidate pass over a SAMPLE of the rows, the m-th best sample candidate of a query becomes
/// its cut for the rest of the table (about m * n / sample rows of the whole table lie below it), so the main pass
/// appends a few dozen keys per query instead of a fixed fraction of the table.  The cut is an ordinary qthr value:
/// the certificate accounts for it, and a cut that turns out too tight only sends the query to the fallback.
static __global__ void sample_cut_kernel(const uint64_t * cand, uint32_t kc, uint32_t m, uint32_t nq, uint32_t * qthr)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq)
        return;
    const uint64_t key = cand[(size_t)q * kc + (m - 1)];
    if (key != KEY_NONE)
        atomicMin(&qthr[q], (uint32_t)(key >> 32));
}

/// Wave-wide selection without insertion: the kc (<= 64) smallest of the wavefront's keys (NW per lane as
/// (hi, lo) words; hi = 0xFFFFFFFF, lo = 0xFFFFFFFF = absent) by a bitwise search for H = the kc-th smallest HIGH word
/// (32 rounds of NW compares + scalar popcounts: no cross-lane traffic, no serial inserts), then two ballot
/// compactions: out[0..c) = the keys with hi < H (c < kc), out[c..kc) = keys with hi == H in (register, lane) order.
/// out[kc-1] always holds a key whose high word is H = the largest approximate value among the candidates, which is all
/// the certificate reads from it (ivf_rerank_kernel: `last`); the candidates are NOT sorted (the re-rank sorts the
/// canonical keys).  Which of several rows with the same approximate value H are taken is arbitrary: the untaken
/// ones are covered by the certificate's strict inequality, like any other row at the cut.
/// The kc-th smallest of the wavefront's words (NW per lane; 0xFFFFFFFF when there are fewer than kc below it): the
/// largest H with count(word < H) < kc, built bit by bit from the top.  32 rounds of 2 NW + 12 instructions: the
/// wavefronts of a SIMD share its VALU, so a kernel of 4 of them per SIMD spends 4 x 1400 x 4 clocks = 11 us in here.
template <int NW>
__device__ inline uint32_t wave_kth_word_bits(const uint32_t (&hi)[NW], uint32_t kc)
{
    uint32_t H = 0;
#pragma unroll 1
    for (int b = 31; b >= 0; b--)
    {
        const uint32_t c = H | (1u << b);
        uint32_t cnt = 0;
#pragma unroll
        for (int u = 0; u < NW; u++)
            cnt += hi[u] < c ? 1u : 0u;
        if (wave_sum_u32(cnt) < kc) // fewer than kc words below c: the kc-th smallest is >= c
            H = c;
    }
    return H;
}

/// The same value by a radix select through a wave-private 256-bin histogram in LDS (hist: 256 words owned by this
/// wavefront): the bits the words do not all share (from the highest bit in which minimum and maximum differ) are fixed 8 at a
/// time -- one LDS atomic per word still inside the current prefix, an in-register prefix sum over the bins (4 per
/// lane), the bin that holds the rank.  Distances of one query differ in ~24 bits: 3 passes of ~5 NW + 60 instructions
/// against 32 rounds.  LDS operations of one wavefront execute in program order: no barrier.
template <int NW>
__device__ inline uint32_t wave_kth_word_radix(const uint32_t (&hi)[NW], uint32_t kc, uint32_t * hist, uint32_t lane)
{
    uint32_t mn = 0xFFFFFFFFu, mx = 0u, real = 0;
#pragma unroll
    for (int u = 0; u < NW; u++)
    {
        mn = min(mn, hi[u]);
        mx = max(mx, hi[u] == 0xFFFFFFFFu ? 0u : hi[u]);
        real += hi[u] != 0xFFFFFFFFu ? 1u : 0u;
    }
    real = wave_sum_u32(real);
    if (real < kc)
        return 0xFFFFFFFFu;
    mn = wave_min_u32(mn);
    mx = wave_max_u32(mx);
    if (mn >= mx) // all present words equal
        return mn;
    int hi_bit = 31 - __builtin_clz(mn ^ mx); // highest bit that is not common
    uint32_t H = hi_bit == 31 ? 0u : mn & ~((2u << hi_bit) - 1u);
    uint32_t need = kc; // rank (1-based) among the words that share the prefix above hi_bit
#pragma unroll 1
    while (hi_bit >= 0)
    {
        const int lo_bit = hi_bit >= 7 ? hi_bit - 7 : 0;
        const uint32_t dmask = (1u << (hi_bit - lo_bit + 1)) - 1u;
        *reinterpret_cast<uint4 *>(hist + 4 * lane) = make_uint4(0u, 0u, 0u, 0u);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < NW; u++)
            if ((((hi[u] ^ H) >> hi_bit) >> 1) == 0) // shares the prefix (absent words that do sort last: never reached)
                atomicAdd(hist + ((hi[u] >> lo_bit) & dmask), 1u);
        __builtin_amdgcn_wave_barrier();
        const uint4 b = *reinterpret_cast<const uint4 *>(hist + 4 * lane);
        const uint32_t own = b.x + b.y + b.z + b.w;
        uint32_t incl = own; // inclusive prefix sum over the lanes: inside the 16-lane rows by DPP, across them by readlane
        incl += dpp32<0x111>(incl);
        incl += dpp32<0x112>(incl);
        incl += dpp32<0x114>(incl);
        incl += dpp32<0x118>(incl);
        const uint32_t r0 = __builtin_amdgcn_readlane((int)incl, 15), r1 = __builtin_amdgcn_readlane((int)incl, 31),
                       r2 = __builtin_amdgcn_readlane((int)incl, 47);
        incl += lane < 16 ? 0u : lane < 32 ? r0 : lane < 48 ? r0 + r1 : r0 + r1 + r2;
        const uint64_t reach = __ballot(incl >= need);
        const int L = __builtin_ctzll(reach); // reach != 0: need <= the words in the prefix group
        uint32_t below = (uint32_t)__builtin_amdgcn_readlane((int)(incl - own), L);
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)b.x, L), b1 = (uint32_t)__builtin_amdgcn_readlane((int)b.y, L),
                       b2 = (uint32_t)__builtin_amdgcn_readlane((int)b.z, L);
        uint32_t digit = 4u * (uint32_t)L;
        if (below + b0 < need)
        {
            below += b0;
            digit++;
            if (below + b1 < need)
            {
                below += b1;
                digit++;
                if (below + b2 < need)
                {
                    below += b2;
                    digit++;
                }
            }
        }
        need -= below;
        H |= digit << lo_bit;
        hi_bit = lo_bit - 1;
    }
    return H;
}

/// hist != nullptr: the radix form (hist = 256 LDS words owned by the calling wavefront); else the bitwise search.
template <int NW>
__device__ inline uint32_t wave_kth_word(const uint32_t (&hi)[NW], uint32_t kc, uint32_t * hist, uint32_t lane)
{
    return hist ? wave_kth_word_radix<NW>(hi, kc, hist, lane) : wave_kth_word_bits<NW>(hi, kc);
}

template <int NW>
__device__ inline void wave_select_words(const uint32_t (&hi)[NW], const uint32_t (&lo)[NW], uint32_t kc, uint64_t * out, uint32_t lane,
                                         uint32_t * hist, uint64_t * stage /* LDS [64], wave-private */, bool sort)
{
    const uint32_t H = wave_kth_word<NW>(hi, kc, hist, lane);
    stage[lane] = KEY_NONE;
    __builtin_amdgcn_wave_barrier();
    uint32_t run = 0;
#pragma unroll
    for (int u = 0; u < NW; u++)
    {
        const bool take = hi[u] < H;
        const uint64_t mask = __ballot(take);
        if (take)
            stage[run + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))]
                = (uint64_t)hi[u] << 32 | lo[u];
        run += (uint32_t)__popcll(mask);
    }
#pragma unroll
    for (int u = 0; u < NW; u++)
    {
        const bool tie = hi[u] == H && run < kc;
        const uint64_t mask = __ballot(tie);
        const uint32_t pos = run + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (tie && pos < kc)
            stage[pos] = (uint64_t)hi[u] << 32 | lo[u];
        run += (uint32_t)__popcll(mask);
    }
    __builtin_amdgcn_wave_barrier();
    const uint64_t mine = stage[lane];
    if (!sort) // only the largest value last (all the certificate reads)
    {
        if (lane < kc)
            out[lane] = mine;
        return;
    }
    // ascending order for the early exit of the re-rank: lane i ranks its key among the kc selected (keys are distinct except
    // KEY_NONE padding, which ranks itself by its slot)
    uint32_t rank = 0;
    for (uint32_t j = 0; j < kc; j++)
    {
        const uint64_t other = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine >> 32), j) << 32
            | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, j);
        rank += other < mine || (other == mine && j < lane) ? 1u : 0u;
    }
    if (lane < kc)
        out[rank] = mine;
}

template <int NW>
__device__ inline void cand_select_wave(const uint64_t * src, uint32_t n, uint32_t kc, uint64_t * out, uint32_t lane, uint32_t * hist,
                                        uint64_t * stage, bool sort)
{
    uint32_t hi[NW], lo[NW];
#pragma unroll
    for (int u = 0; u < NW; u++)
    {
        const uint32_t i = u * WAVE + lane;
        const uint64_t key = i < n ? src[i] : KEY_NONE;
        hi[u] = (uint32_t)(key >> 32);
        lo[u] = (uint32_t)key;
    }
    wave_select_words<NW>(hi, lo, kc, out, lane, hist, stage, sort);
}

constexpr uint32_t CAND_SELECT_WAVE_CAP = 2048; // 32 keys per lane

/// cand_select_kernel for buffers of at most CAND_SELECT_WAVE_CAP keys: ONE wavefront per query holds the whole
/// buffer in registers (a query of the bench keeps ~250 keys below its cut: 4 per lane) and selects by
/// wave_select_words.  out[q][kc]: the candidates (ascending when `sort`, else only the largest value last), KEY_NONE padded.
static __global__ __launch_bounds__(BLOCK) void cand_select_wave_kernel(const uint64_t * buf, const uint32_t * qcnt,
                                                                         const uint32_t * qthr, uint32_t cap, uint32_t nq,
                                                                         uint32_t kc, uint64_t * out, uint64_t * bound, int radix, int sort)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[BLOCK / WAVE][256];
    const uint32_t q = blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nq)
        return;
    __shared__ uint64_t s_stage[BLOCK / WAVE][WAVE];
    uint32_t * hist = radix ? s_hist[threadIdx.x >> 6] : nullptr;
    uint64_t * stage = s_stage[threadIdx.x >> 6];
    const uint32_t n = qcnt[q] < cap ? qcnt[q] : cap;
    const uint64_t * src = buf + (size_t)q * cap;
    uint64_t * dst = out + (size_t)q * kc;
    if (n <= 4 * WAVE)
        cand_select_wave<4>(src, n, kc, dst, lane, hist, stage, sort != 0);
    else if (n <= 8 * WAVE)
        cand_select_wave<8>(src, n, kc, dst, lane, hist, stage, sort != 0);
    else if (n <= 16 * WAVE)
        cand_select_wave<16>(src, n, kc, dst, lane, hist, stage, sort != 0);
    else
        cand_select_wave<32>(src, n, kc, dst, lane, hist, stage, sort != 0);
    if (lane == 0) // an overflowed buffer dropped unknown keys: bound 0 = nothing can be certified
        bound[q] = qcnt[q] > cap ? 0 : (qthr[q] == 0xFFFFFFFFu ? KEY_NONE : (uint64_t)qthr[q] << 32);
}

/// The kc (<= 64 R) best of the keys a query's slices appended (unsorted runs), ascending, one block per query (the
/// 4 wavefronts take interleaved 256-key chunks, then a rank merge): out[q][kc] (KEY_NONE padded);
/// bound[q] = the smallest key any slice may have cut (from qthr; KEY_NONE if none).
template <int R> // kc <= 64 R
static __global__ __launch_bounds__(BLOCK) void cand_select_kernel(const uint64_t * buf, const uint32_t * qcnt,
                                                                    const uint32_t * qthr, uint32_t cap, uint32_t nq,
                                                                    uint32_t kc, uint64_t * out, uint64_t * bound)
{
    __shared__ uint64_t lds[5 * 64 * R];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = blockIdx.x;
    const uint32_t n = qcnt[q] < cap ? qcnt[q] : cap;
    const uint64_t * src = buf + (size_t)q * cap;
    WaveTopK<R> top;
    top.init();
    for (uint32_t base = wave * 4 * WAVE; base < n; base += 4 * 4 * WAVE)
    {
        uint64_t key[4]; // 4 independent loads in flight before the first (serialising) offer
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            const uint32_t i = base + u * WAVE + lane;
            key[u] = i < n ? src[i] : KEY_NONE;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            top.offer(key[u], kc, lane);
    }
    top.store(lds + wave * kc, kc, lane);
    __syncthreads();
    uint64_t * merged = lds + 4 * kc;
    block_rank_merge(lds, kc, merged, kc, tid);
    if (tid < kc)
        out[(size_t)q * kc + tid] = merged[tid];
    if (tid == 0) // an overflowed buffer dropped unknown keys: bound 0 = nothing can be certified
        bound[q] = qcnt[q] > cap ? 0 : (qthr[q] == 0xFFFFFFFFu ? KEY_NONE : (uint64_t)qthr[q] << 32);
}

/// The kc (<= 256) best of a query's candidate buffer by a BLOCK-wide radix select (one block per query, any buffer
/// size): four passes over the high words -- an LDS histogram of the current byte among the keys still inside the
/// prefix, a 256-thread prefix sum, the bin that holds the kc-th -- then the keys below the value and the ties are
/// collected in LDS and ranked against each other, so out[q][kc] is ascending (KEY_NONE padded).  The insertion kernel
/// above needs ~kc (1 + ln(n / kc)) serial list insertions per wavefront: 1.03 ms per 4096 queries at kc = 256,
/// n ~ 2500 (k = 100), against ~0.03 ms here.
static __global__ __launch_bounds__(BLOCK) void cand_select_block_kernel(const uint64_t * buf, const uint32_t * qcnt,
                                                                          const uint32_t * qthr, uint32_t cap, uint32_t nq,
                                                                          uint32_t kc, uint64_t * out, uint64_t * bound)
{
    __shared__ uint32_t hist[256];
    __shared__ uint32_t wsum[BLOCK / WAVE];
    __shared__ uint32_t s_digit, s_below, s_cnt;
    __shared__ uint64_t sel[256];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = blockIdx.x;
    const uint32_t n = qcnt[q] < cap ? qcnt[q] : cap;
    const uint64_t * src = buf + (size_t)q * cap;
    const bool all = n <= kc; // every key is a candidate
    uint32_t H = 0, need = kc;
    if (!all)
        for (int shift = 24; shift >= 0; shift -= 8)
        {
            hist[tid] = 0;
            __syncthreads();
            const uint32_t pmask = shift == 24 ? 0u : ~0u << (shift + 8); // the bits fixed so far
            for (uint32_t i = tid; i < n; i += BLOCK)
            {
                const uint32_t hi = (uint32_t)(src[i] >> 32);
                if ((hi & pmask) == H)
                    atomicAdd(&hist[(hi >> shift) & 255u], 1u);
            }
            __syncthreads();
            const uint32_t own = hist[tid]; // thread t owns bin t: inclusive prefix sum over the block
            uint32_t incl = own;
            incl += dpp32<0x111>(incl);
            incl += dpp32<0x112>(incl);
            incl += dpp32<0x114>(incl);
            incl += dpp32<0x118>(incl);
            const uint32_t r0 = __builtin_amdgcn_readlane((int)incl, 15), r1 = __builtin_amdgcn_readlane((int)incl, 31),
                           r2 = __builtin_amdgcn_readlane((int)incl, 47);
            incl += lane < 16 ? 0u : lane < 32 ? r0 : lane < 48 ? r0 + r1 : r0 + r1 + r2;
            if (lane == 63)
                wsum[wave] = incl;
            __syncthreads();
            for (uint32_t w = 0; w < wave; w++)
                incl += wsum[w];
            if (incl - own < need && need <= incl) // exactly one bin holds the rank
            {
                s_digit = tid;
                s_below = incl - own;
            }
            __syncthreads();
            H |= s_digit << shift;
            need -= s_below;
            __syncthreads();
        }
    if (tid == 0)
        s_cnt = 0;
    sel[tid] = KEY_NONE;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += BLOCK) // below the kc-th value: fewer than kc keys
    {
        const uint64_t key = src[i];
        if (all || (uint32_t)(key >> 32) < H)
            sel[atomicAdd(&s_cnt, 1u)] = key;
    }
    __syncthreads();
    if (!all)
        for (uint32_t i = tid; i < n; i += BLOCK) // at the value: as many as there is room for
        {
            const uint64_t key = src[i];
            if ((uint32_t)(key >> 32) == H)
            {
                const uint32_t pos = atomicAdd(&s_cnt, 1u);
                if (pos < kc)
                    sel[pos] = key;
            }
        }
    __syncthreads();
    const uint64_t mine = sel[tid];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < 256; j++)
    {
        const uint64_t other = sel[j];
        rank += other < mine || (other == mine && j < tid) ? 1u : 0u;
    }
    if (rank < kc)
        out[(size_t)q * kc + rank] = mine;
    if (tid == 0) // an overflowed buffer dropped unknown keys: bound 0 = nothing can be certified
        bound[q] = qcnt[q] > cap ? 0 : (qthr[q] == 0xFFFFFFFFu ? KEY_NONE : (uint64_t)qthr[q] << 32);
}

/// (second chance: ivf_rerank_all_kernel below)
constexpr uint32_t RA_KMAX = 128, RA_CHUNK = 256;
struct RerankAllParams
{
    const uint64_t * partial; // [nq][cap] candidate keys (approximate word << 32 | row position)
    const uint32_t * qcnt;    // [nq] candidates appended (may exceed cap: overflow)
    const uint32_t * qthr;    // [nq] the cut (0xFFFFFFFF = none: every probed row is a candidate)
    uint32_t cap;
    const uint32_t * failq_in;
    const uint32_t * nfail_in;
    uint32_t * failq_out;
    uint32_t * nfail_out; // zeroed by the caller
    unsigned long long * stat_fail;
    const uint64_t * ek_in; // nullable [nq]: RerankParams::ek_out of the first stage
};

struct RerankParams
{
    const float4 * Y;      // rows, ld4 float4 each
    const uint32_t * ids;  // id of stored row r
    const float4 * Q;      // queries
    const float * qnorm;   // |q|^2 (approximate)
    const uint64_t * cand; // [nq][kc] approximate keys, low word = row position; the largest value LAST (ascending for early_exit to pay)
    const uint64_t * bound; // nullable [nq]: smallest key any earlier stage may have dropped (KEY_NONE = none dropped)
    uint32_t kc, k, ld4;
    int64_t * out_ids; // [nq][k]
    float * out_dis;
    int32_t * out_probes; // non-null: write [nq][k] int32 ids instead (the coarse quantiser's probe lists)
    int cosine;
    // error model of the approximate pass (each times the experiment knob MSVS_IVF_EPS_SCALE):
    double c_dot;   // |approximate <x,q> - true| <= c_dot * |x||q|
    const float * qrho; // nullable [nq] (fp16 shadow passes): ... <= (c_dot + qrho_scale * qrho[q]) * |x||q| -- the measured rounding
    double qrho_scale;  //   error of the query's own image (set_error_model_h16)
    double c_norm;  // |approximate |v|^2 - true| <= c_norm * |v|^2
    double c_canon; // |canonical result - true| <= c_canon * (|x|+|q|)^2 (L2), * |x||q| (IP)
    float xmax;     // max |x|^2 over the index rows
    uint32_t * failq; // queries whose certificate failed ...
    uint32_t * nfail; // ... and their count (zeroed by the caller)
    unsigned long long * stat_fail; // nullable: process-wide running total
    unsigned long long * stat_skip; // nullable (experiments): [0] += candidates beyond e_k +- eps, [1] += candidates
    int early_exit; // rounds after the first ceil(k / 16) skip candidates whose approximate value is beyond e_k +- eps
    int band; // probe lists (out_probes) only: candidates that are certainly inside / outside the exact top-k by their approximate
              // values alone are not evaluated (see ivf_rerank_kernel)
    uint64_t * ek_out; // nullable [nq]: a query WITHOUT a certificate leaves the k-th exact key of the candidates it evaluated here
                       // (KEY_NONE: fewer than k) -- an upper bound of its true k-th distance for the second chance    // Round 6: the SECOND CHANCE in the same launch (result passes, 256-thread blocks): a query whose certificate fails has its whole
    // candidate buffer re-ranked by its own block right away (rerank_all_query) instead of being queued for ivf_rerank_all_kernel --
    // one launch less, and the second chances (a sixth of the queries on SURVEY 8d's sigma-0.3 blobs) run beside the other queries'
    // first stage.  fuse_second != 0: `ra` is the second chance's view; what still fails goes to ra.failq_out / ra.nfail_out.
    int fuse_second = 0;
    RerankAllParams ra{};
    // Round 6 (probe lists): only the queries qmap[0 .. *qcount) -- those coarse_tail_kernel could not serve with a band (block b = entry b)
    const uint32_t * qmap = nullptr;
    const uint32_t * qcount = nullptr;
};

/// |approximate value - canonical value| <= eps for every row of the table and this query (sx, sq: upper bounds of |x|, |q|).
/// L2: a = |x|^2 + |q|^2 - 2<x,q> with approximate norms and product; IP: a = <x,q>.
template <int METRIC>
__device__ __forceinline__ double rerank_eps(const RerankParams & a, double sx, double sq, uint32_t q)
{
    const double c_dot = a.qrho ? a.c_dot + a.qrho_scale * (double)a.qrho[q] : a.c_dot;
    return (METRIC == M_L2 ? 2.0 * c_dot * sx * sq + a.c_norm * (sx * sx + sq * sq) + (a.c_canon + 4e-7) * (sx + sq) * (sx + sq)
                           : (c_dot + a.c_canon) * sx * sq)
        + 1e-30;
}

template <int METRIC>
__device__ __forceinline__ bool rerank_all_query(const RerankParams & a, const RerankAllParams & b, const uint32_t q, float4 * qs,
                                                 const bool qs_ready, unsigned char * lds, const uint64_t hint);

/// One block of 16 G threads per query: G groups of 16 lanes, a group per candidate row and round.  G = 16; 32 (all
/// candidates of a k <= 12 search in flight at once) measured SLOWER: 70 against 47 us per 4096 queries -- kept as a knob.
/// dynamic LDS: ld4*16 + 64*R*8 bytes.
template <int METRIC, int G, int R> // R: kc <= 64 R (256 candidates for 40 < k <= 128)
__global__ __launch_bounds__(16 * G) void ivf_rerank_kernel(const RerankParams a)
{
    float4 * qs = reinterpret_cast<float4 *>(msvs_smem);
    uint64_t * keys = reinterpret_cast<uint64_t *>(msvs_smem + (size_t)a.ld4 * 16);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = tid >> 4, g = tid & 15;
    if (a.qmap && blockIdx.x >= *a.qcount)
        return;
    const uint32_t q = a.qmap ? a.qmap[blockIdx.x] : blockIdx.x, ld4 = a.ld4, kc = a.kc;
    for (uint32_t c = tid; c < ld4; c += 16 * G)
        qs[c] = a.Q[(size_t)q * ld4 + c];
    for (uint32_t c = tid; c < 64 * R; c += 16 * G)
        keys[c] = KEY_NONE;
    __syncthreads();
    const uint32_t jfull = ld4 >> 4, jtail = ld4 & 15;
    // BAND (probe lists): only the SET of the k best rows leaves this kernel, and most of it is decided by the approximate
    // values alone.  With a_(k), a_(k+1) the k-th and (k+1)-th smallest approximate values of the candidates and
    // |a - exact| <= eps for every row: a candidate with a + 2 eps < a_(k+1) is beaten by at most the k - 1 other rows of
    // approximate rank <= k (everything else has exact >= a_(k+1) - eps > a + eps) -- certainly IN; one with a - 2 eps > a_(k)
    // is beaten by the k rows of approximate rank <= k -- certainly OUT, and so is every row that is not a candidate when
    // min(last, bound) - 2 eps > a_(k).  The rest, the band around the boundary, is evaluated canonically and fills the
    // remaining slots in exact order.  On the bench step 3-4 of 64 centroid rows per query instead of 64 (45 -> ~10 us).
    __shared__ uint8_t s_state[64 * R]; // 0: evaluate, 1: certainly in, 2: certainly out
    __shared__ uint64_t s_ak, s_ak1;
    __shared__ uint32_t s_nin;
    __shared__ int s_band;
    if (a.band && a.out_probes && kc > a.k && kc <= 16 * G)
    {
        if (tid < kc)
            keys[tid] = a.cand[(size_t)q * kc + tid];
        if (tid == 0)
        {
            s_ak = s_ak1 = KEY_NONE;
            s_nin = 0;
            s_band = 0;
        }
        __syncthreads();
        const uint64_t mine = tid < kc ? keys[tid] : KEY_NONE;
        uint32_t rank = 0; // of this candidate among all of them by approximate key
        if (mine != KEY_NONE)
        {
            for (uint32_t j = 0; j < kc; j++)
                rank += keys[j] < mine || (keys[j] == mine && j < tid) ? 1u : 0u;
            if (rank == a.k - 1)
                s_ak = mine;
            if (rank == a.k)
                s_ak1 = mine;
        }
        __syncthreads();
        const float qn = a.qnorm[q];
        const bool usable = s_ak != KEY_NONE && s_ak1 != KEY_NONE && qn < 1e30f && a.xmax < 1e30f;
        double eps2 = 0.0, ak = 0.0, ak1 = 0.0;
        bool band_ok = usable;
        if (usable)
        {
            eps2 = 2.0 * rerank_eps<METRIC>(a, sqrt((double)a.xmax * 1.001), sqrt((double)qn * 1.001), q);
            ak = (double)key_value<METRIC>(s_ak);
            ak1 = (double)key_value<METRIC>(s_ak1);
            uint64_t last = keys[kc - 1]; // the candidates' largest approximate value comes last; KEY_NONE: every row is a candidate
            if (a.bound && a.bound[q] < last)
                last = a.bound[q];
            if (last != KEY_NONE)
            {
                const double al = (double)key_value<METRIC>(last);
                band_ok = METRIC == M_L2 ? (al - eps2 > ak) : (al + eps2 < ak);
            }
        }
        if (band_ok && tid < kc)
        {
            uint8_t st = 2;
            if (mine != KEY_NONE)
            {
                const double aj = (double)key_value<METRIC>(mine);
                const bool in = METRIC == M_L2 ? (aj + eps2 < ak1) : (aj - eps2 > ak1);
                const bool out = METRIC == M_L2 ? (aj - eps2 > ak) : (aj + eps2 < ak);
                st = in ? 1 : (out ? 2 : 0);
                if (in)
                {
                    // "in" is monotone in the approximate value: the certainly-in rows are the ranks 0 .. n_in - 1, and each
                    // takes the slot of its rank -- the probe list's head is in approximate order, the same from run to run
                    // (slots taken with an atomic counter made the order a race: ADVICE round 3)
                    const uint32_t pos = (uint32_t)mine;
                    a.out_probes[(size_t)q * a.k + rank] = (int32_t)(a.ids ? a.ids[pos] : pos);
                    atomicAdd(&s_nin, 1u);
                }
            }
            s_state[tid] = st;
        }
        if (tid == 0)
            s_band = band_ok ? 1 : 0;
        __syncthreads();
        if (tid < kc)
            keys[tid] = KEY_NONE;
        __syncthreads();
    }
    else
    {
        if (tid == 0)
        {
            s_band = 0;
            s_nin = 0;
        }
        __syncthreads();
    }
    const bool band = s_band != 0;
    // Early exit: the candidates arrive in ascending approximate order, G per round.  Once the first ceil(k / G) rounds
    // have given an exact k-th distance e, a later candidate with approximate value a beyond e by more than the error
    // bound (a - eps > e: its exact distance is > e >= the final k-th) cannot enter the result: its row is not read
    // (valid whatever the order is; the order makes it effective).  List scan of the bench step: 57 % of the 32 candidates
    // per query are never read (of 69 % that are not results), 75 -> 50 us; the coarse quantiser's table lives in L2 and
    // gains nothing (measured), so the host leaves it off there and its candidates unsorted.
    __shared__ double s_e, s_eps;
    __shared__ int s_skip;
    __shared__ uint64_t s_ek;
    const uint32_t first = a.early_exit && kc % G == 0 ? (a.k + G - 1) / G * G : kc; // every group walks kc / G rounds: the barrier is uniform
    for (uint32_t c = grp; c < kc; c += G)
    {
        if (c - grp == first) // uniform over the block: the first rounds are complete
        {
            __syncthreads();
            // the exact k-th key among the first rounds' results: every key ranks itself (keys are distinct up to duplicates
            // of a row, which rank by their slot); a sorted-list insertion per key took 17 us per block at kc = 256
            for (uint32_t i = tid; i < first; i += 16 * G)
            {
                const uint64_t mine = keys[i];
                uint32_t rank = 0;
                for (uint32_t j = 0; j < first; j++)
                    rank += keys[j] < mine || (keys[j] == mine && j < i) ? 1u : 0u;
                if (rank == a.k - 1)
                    s_ek = mine;
            }
            __syncthreads();
            if (tid == 0)
            {
                const float qn = a.qnorm[q];
                const bool usable = s_ek != KEY_NONE && qn < 1e30f && a.xmax < 1e30f;
                s_skip = usable ? 1 : 0;
                if (usable)
                {
                    const double sx = sqrt((double)a.xmax * 1.001), sq = sqrt((double)qn * 1.001);
                    s_e = (double)key_value<METRIC>(s_ek);
                    s_eps = rerank_eps<METRIC>(a, sx, sq, q);
                }
            }
            __syncthreads();
        }
        const uint64_t ck = a.cand[(size_t)q * kc + c]; // uniform over the 16 lanes that own candidate c
        if (ck == KEY_NONE || (band && s_state[c] != 0))
            continue;
        if (c - grp >= first && s_skip)
        {
            const double aj = (double)key_value<METRIC>(ck);
            if (METRIC == M_L2 ? (aj - s_eps > s_e) : (aj + s_eps < s_e))
            {
                if (a.stat_skip && g == 0) // experiments: rows really skipped (slot 6 = result passes, 7 = coarse passes)
                    atomicAdd(a.stat_skip + (a.out_probes ? 3 : 4), 1ull);
                continue;
            }
        }
        const uint32_t pos = (uint32_t)ck;
        const float4 * yrow = a.Y + (size_t)pos * ld4 + g;
        const float4 * qrow = qs + g;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t j = 0;
        for (; j + 4 <= jfull; j += 4) // (8 pieces in flight measured slower here: 4096 blocks, the occupancy is the parallelism)
        {
            const float4 y0 = yrow[j * 16], y1 = yrow[(j + 1) * 16], y2 = yrow[(j + 2) * 16], y3 = yrow[(j + 3) * 16];
            canonical_update<METRIC>(acc, qrow[j * 16], y0);
            canonical_update<METRIC>(acc, qrow[(j + 1) * 16], y1);
            canonical_update<METRIC>(acc, qrow[(j + 2) * 16], y2);
            canonical_update<METRIC>(acc, qrow[(j + 3) * 16], y3);
        }
        for (; j < jfull; j++)
            canonical_update<METRIC>(acc, qrow[j * 16], yrow[j * 16]);
        if (g < jtail)
            canonical_update<METRIC>(acc, qrow[jfull * 16], yrow[jfull * 16]);
        float s = __fadd_rn(__fadd_rn(acc.x, acc.y), __fadd_rn(acc.z, acc.w));
        s = row16_tree_sum(s);
        if (g == 0)
            keys[c] = make_key<METRIC>(s, a.ids ? a.ids[pos] : pos);
    }
    __syncthreads();
    // exact top-k of the evaluated candidates: every key ranks itself among the 64 R slots (skipped / absent = KEY_NONE)
    for (uint32_t i = tid; i < 64 * R; i += 16 * G)
    {
        const uint64_t mine = keys[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < 64 * R; j++)
            rank += keys[j] < mine || (keys[j] == mine && j < i) ? 1u : 0u;
        if (band)
        {
            // the band's rows fill the slots the certainly-in rows left, in exact order
            if (mine != KEY_NONE && s_nin + rank < a.k)
                a.out_probes[(size_t)q * a.k + s_nin + rank] = (int32_t)(uint32_t)mine;
        }
        else if (rank < a.k)
        {
            const size_t o = (size_t)q * a.k + rank;
            if (a.out_probes)
                a.out_probes[o] = mine == KEY_NONE ? -1 : (int32_t)(uint32_t)mine;
            else
            {
                a.out_ids[o] = mine == KEY_NONE ? -1 : (int64_t)(uint32_t)mine;
                const float v = key_value<METRIC>(mine);
                a.out_dis[o] = a.cosine ? __fsub_rn(1.0f, v) : v;
            }
            if (rank == a.k - 1)
                s_ek = mine;
        }
    }
    __syncthreads();
    const bool fuse = a.fuse_second != 0 && G == 16 && !a.out_probes; // the second chance in this launch (uniform)
    if (band || (wave != 0 && !fuse)) // (a band that could be formed IS the certificate)
        return;
    __shared__ int s_failed;
    if (wave == 0)
    {
    // certificate (see the header comment): `last` = the smallest approximate key a non-candidate row can have; a
    // candidate list that is not full (and no truncated slice) holds every probed row
    uint64_t last = a.cand[(size_t)q * kc + kc - 1];
    if (a.bound && a.bound[q] < last)
        last = a.bound[q];
    bool ok = true;
    if (last != KEY_NONE)
    {
        const uint64_t ek = s_ek;
        const float qn = a.qnorm[q];
        if (ek == KEY_NONE || !(qn < 1e30f) || !(a.xmax < 1e30f))
            ok = false;
        else
        {
            const double al = (double)key_value<METRIC>(last), e = (double)key_value<METRIC>(ek);
            const double sx = sqrt((double)a.xmax * 1.001), sq = sqrt((double)qn * 1.001);
            const double eps = rerank_eps<METRIC>(a, sx, sq, q);
            ok = METRIC == M_L2 ? (al - eps > e) : (al + eps < e);
            if (a.stat_skip) // experiment: how many candidates an early exit could have skipped (approximate value beyond e_k +- eps)
            {
                const uint64_t ck = lane < kc ? a.cand[(size_t)q * kc + lane] : KEY_NONE;
                const double aj = (double)key_value<METRIC>(ck);
                const bool skippable = ck != KEY_NONE && (METRIC == M_L2 ? (aj - eps > e) : (aj + eps < e));
                const uint32_t ns = (uint32_t)__popcll(__ballot(skippable)), nc = (uint32_t)__popcll(__ballot(ck != KEY_NONE));
                if (lane == 0)
                {
                    atomicAdd(a.stat_skip, (unsigned long long)ns);
                    atomicAdd(a.stat_skip + 1, (unsigned long long)nc);
                }
            }
        }
    }
    if (!ok && lane == 0 && !fuse)
    {
        a.failq[atomicAdd(a.nfail, 1u)] = q;
        if (a.ek_out)
            a.ek_out[q] = s_ek;
        if (a.stat_fail)
            atomicAdd(a.stat_fail, 1ull);
    }
    if (fuse && lane == 0)
        s_failed = ok ? 0 : 1;
    }
    if (!fuse)
        return;
    __syncthreads();
    if (!s_failed)
        return;
    // the second chance, here and now: every row of the query's candidate buffer, certified against the cut (rerank_all_query); the
    // query is in LDS already, the first stage's k-th exact key is the hint
    const bool ok2 = rerank_all_query<METRIC>(a, a.ra, q, qs, true, msvs_smem + (size_t)ld4 * 16 + (size_t)64 * R * 8, s_ek);
    if (!ok2 && tid == 0)
    {
        a.ra.failq_out[atomicAdd(a.ra.nfail_out, 1u)] = q;
        if (a.ra.stat_fail)
            atomicAdd(a.ra.stat_fail, 1ull);
    }
}

/// SECOND CHANCE of a query whose certificate failed: the canonical distance of EVERY row in its candidate buffer (everything the
/// pass kept below the query's cut: a few hundred rows, not the kc best of them), exact top-k by ranking, certificate against
/// the cut itself -- the smallest approximate value a row outside the buffer can have.  The first certificate compares the
/// k-th exact distance with the kc-th approximate one (32 candidates for k <= 12): on data whose distances concentrate
/// (1024 blobs, sigma 0.3, in R^768: the 10th and the 32nd neighbour are 3 eps apart) a sixth of the queries fail it, and the
/// canonical scan of all their probed lists costs 20 x the step.  Here the margin is cut - e_k (rank ~250 against rank k);
/// only a query that fails this one too, or whose buffer overflowed, goes to the canonical scan (failq_out).
/// Not every row of the buffer is evaluated: the first stage found k rows, so their k-th exact distance E0 (ek_in) bounds the
/// true k-th from above, and a row whose approximate value lies beyond E0 by more than eps has an exact distance > E0 -- it
/// cannot enter the result and its 3 KB are not read (on the sigma-0.3 blobs at nprobe 2 most of the ~250 rows of a buffer).
/// One block of 256 threads per failed query (grid-stride over *nfail_in), 16 lanes per candidate row as in ivf_rerank_kernel.
/// dynamic LDS: ld4 * 16 + (2 * RA_KMAX + RA_CHUNK) * 8 + RA_CHUNK * 4 + 16 bytes.

/// The second chance of ONE query by a block of 256 threads (see above).  qs: the query in LDS (loaded here unless qs_ready); lds: (2 *
/// RA_KMAX + RA_CHUNK) * 8 + RA_CHUNK * 4 + 16 bytes of scratch; hint: the first stage's k-th exact key (KEY_NONE: none).  -> true: the
/// query has its certificate and its results are written.  Uniform over the block (barriers inside).
template <int METRIC>
__device__ __forceinline__ bool rerank_all_query(const RerankParams & a, const RerankAllParams & b, const uint32_t q, float4 * qs,
                                                 const bool qs_ready, unsigned char * lds, const uint64_t hint)
{
    uint64_t * best = reinterpret_cast<uint64_t *>(lds);          // [RA_KMAX] running exact top-k, ascending
    uint64_t * chunk = best + RA_KMAX;                            // [RA_CHUNK] this round's exact keys
    uint64_t * tmp = chunk + RA_CHUNK;                            // [RA_KMAX]
    uint32_t * sel = reinterpret_cast<uint32_t *>(tmp + RA_KMAX); // [RA_CHUNK] row positions to evaluate this round
    uint32_t * wcnt = sel + RA_CHUNK;                             // [4] of them per wavefront
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = tid >> 4, g = tid & 15;
    const uint32_t ld4 = a.ld4, k = a.k;
    const uint32_t jfull = ld4 >> 4, jtail = ld4 & 15;
    const uint32_t cnt = b.qcnt[q];
    bool ok = cnt <= b.cap; // an overflowed buffer dropped rows below the cut: nothing to certify
    __syncthreads();        // whoever used the LDS arrays before is done with them
    if (!ok)
        return false;
    // rows that cannot beat the first stage's k-th exact distance are skipped
    const float qn0 = a.qnorm[q];
    const bool have_hint = hint != KEY_NONE && qn0 < 1e30f && a.xmax < 1e30f;
    double e0 = 0.0, eps0 = 0.0;
    if (have_hint)
    {
        e0 = (double)key_value<METRIC>(hint);
        eps0 = rerank_eps<METRIC>(a, sqrt((double)a.xmax * 1.001), sqrt((double)qn0 * 1.001), q);
    }
    if (!qs_ready)
        for (uint32_t c = tid; c < ld4; c += 256)
            qs[c] = a.Q[(size_t)q * ld4 + c];
    for (uint32_t c = tid; c < RA_KMAX; c += 256)
        best[c] = tmp[c] = KEY_NONE;
    __syncthreads();
    for (uint32_t base = 0; base < cnt; base += RA_CHUNK)
    {
        // which of this chunk's candidates have to be evaluated: compacted, so that the 16 row groups share them evenly
        // (most are skipped by the hint; a group per candidate slot left most groups idle round after round, and one
        // failing query of a small batch costs the whole step its ~16 rounds)
        {
            bool take = base + tid < cnt;
            uint64_t pk = 0;
            if (take)
            {
                pk = b.partial[(size_t)q * b.cap + base + tid];
                if (have_hint)
                {
                    const double aj = (double)key_value<METRIC>(pk & 0xFFFFFFFF00000000ull);
                    take = METRIC == M_L2 ? !(aj - eps0 > e0) : !(aj + eps0 < e0);
                }
            }
            const uint64_t m = __ballot(take);
            if (lane == 0)
                wcnt[wave] = (uint32_t)__popcll(m);
            chunk[tid] = KEY_NONE;
            __syncthreads();
            uint32_t off = 0;
            for (uint32_t w2 = 0; w2 < wave; w2++)
                off += wcnt[w2];
            if (take)
                sel[off + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint32_t)pk;
            __syncthreads();
        }
        const uint32_t ntake = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        for (uint32_t c = grp; c < ntake; c += 16)
        {
            const uint32_t pos = sel[c];
            const float4 * yrow = a.Y + (size_t)pos * ld4 + g;
            const float4 * qrow = qs + g;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            uint32_t j = 0;
            for (; j + 8 <= jfull; j += 8) // 8 row pieces in flight per lane (the arithmetic stays in column order)
            {
                float4 y[8];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    y[u] = yrow[(j + u) * 16];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    canonical_update<METRIC>(acc, qrow[(j + u) * 16], y[u]);
            }
            for (; j + 4 <= jfull; j += 4)
            {
                const float4 y0 = yrow[j * 16], y1 = yrow[(j + 1) * 16], y2 = yrow[(j + 2) * 16], y3 = yrow[(j + 3) * 16];
                canonical_update<METRIC>(acc, qrow[j * 16], y0);
                canonical_update<METRIC>(acc, qrow[(j + 1) * 16], y1);
                canonical_update<METRIC>(acc, qrow[(j + 2) * 16], y2);
                canonical_update<METRIC>(acc, qrow[(j + 3) * 16], y3);
            }
            for (; j < jfull; j++)
                canonical_update<METRIC>(acc, qrow[j * 16], yrow[j * 16]);
            if (g < jtail)
                canonical_update<METRIC>(acc, qrow[jfull * 16], yrow[jfull * 16]);
            float s = __fadd_rn(__fadd_rn(acc.x, acc.y), __fadd_rn(acc.z, acc.w));
            s = row16_tree_sum(s);
            if (g == 0)
                chunk[c] = make_key<METRIC>(s, a.ids ? a.ids[pos] : pos);
        }
        __syncthreads();
        // the k best of (running k, this chunk): every key ranks itself among the RA_KMAX + RA_CHUNK slots
        for (uint32_t i = tid; i < RA_KMAX + RA_CHUNK; i += 256)
        {
            const uint64_t mine = best[i]; // best and chunk are contiguous
            if (mine == KEY_NONE)
                continue;
            uint32_t rank = 0;
            for (uint32_t j = 0; j < RA_KMAX + RA_CHUNK; j++)
                rank += best[j] < mine || (best[j] == mine && j < i) ? 1u : 0u;
            if (rank < RA_KMAX)
                tmp[rank] = mine;
        }
        __syncthreads();
        for (uint32_t c = tid; c < RA_KMAX; c += 256) // ranks nobody took stay KEY_NONE
        {
            best[c] = tmp[c];
            tmp[c] = KEY_NONE;
        }
        __syncthreads();
    }
    // certificate against the cut
    if (tid == 0)
    {
        const uint32_t cutw = b.qthr[q];
        bool good = true;
        if (cutw != 0xFFFFFFFFu)
        {
            const uint64_t ek = best[k - 1];
            const float qn = a.qnorm[q];
            if (ek == KEY_NONE || !(qn < 1e30f) || !(a.xmax < 1e30f))
                good = false;
            else
            {
                const double al = (double)key_value<METRIC>((uint64_t)cutw << 32), e = (double)key_value<METRIC>(ek);
                const double sx = sqrt((double)a.xmax * 1.001), sq = sqrt((double)qn * 1.001);
                const double eps = rerank_eps<METRIC>(a, sx, sq, q);
                good = METRIC == M_L2 ? (al - eps > e) : (al + eps < e);
            }
        }
        *reinterpret_cast<volatile uint32_t *>(tmp) = good ? 1u : 0u;
    }
    __syncthreads();
    ok = *reinterpret_cast<volatile uint32_t *>(tmp) != 0;
    if (ok)
        for (uint32_t r = tid; r < k; r += 256)
        {
            const uint64_t mine = best[r];
            const size_t o = (size_t)q * k + r;
            a.out_ids[o] = mine == KEY_NONE ? -1 : (int64_t)(uint32_t)mine;
            const float v = key_value<METRIC>(mine);
            a.out_dis[o] = a.cosine ? __fsub_rn(1.0f, v) : v;
        }
    return ok;
}

template <int METRIC>
__global__ __launch_bounds__(256) void ivf_rerank_all_kernel(const RerankParams a, const RerankAllParams b)
{
    float4 * qs = reinterpret_cast<float4 *>(msvs_smem);
    unsigned char * lds = msvs_smem + (size_t)a.ld4 * 16;
    const uint32_t nf = *b.nfail_in;
    for (uint32_t f = blockIdx.x; f < nf; f += gridDim.x)
    {
        const uint32_t q = b.failq_in[f];
        const bool ok = rerank_all_query<METRIC>(a, b, q, qs, false, lds, b.ek_in ? b.ek_in[q] : KEY_NONE);
        if (!ok && threadIdx.x == 0)
        {
            b.failq_out[atomicAdd(b.nfail_out, 1u)] = q;
            if (b.stat_fail)
                atomicAdd(b.stat_fail, 1ull);
        }
    }
}

/// ivf_scan_kernel over a device-side list of queries: grid (seg_max, nprobe, Z); block z handles the entries
/// slot_base + z, + Z, ... of a.qmap inside this round's window (usually *a.qcount is 0: every block exits at once); the
/// partial lists are indexed by the entry's position in the window, so the buffers hold slot_cap queries, not nq.
template <int METRIC, int R>
__global__ __launch_bounds__(BLOCK) void ivf_scan_subset_kernel(const ScanParams a)
{
    const uint32_t s = blockIdx.x, p = blockIdx.y;
    const uint32_t nf = *a.qcount < a.slot_base + a.slot_cap ? *a.qcount : a.slot_base + a.slot_cap;
    float4 * qs = reinterpret_cast<float4 *>(msvs_smem);
    uint64_t * lds_merge = reinterpret_cast<uint64_t *>(msvs_smem + (size_t)a.ld4 * 16);
    for (uint32_t f = a.slot_base + blockIdx.z; f < nf; f += gridDim.z)
    {
        const uint32_t q = a.qmap[f];
        const int32_t list = a.probes[(size_t)q * a.nprobe + p];
        int64_t lb = 0, le = 0;
        if (list >= 0)
        {
            lb = a.list_off[list] + (int64_t)s * a.rows_per_block;
            le = a.list_off[list + 1];
            if (le > lb + a.rows_per_block)
                le = lb + a.rows_per_block;
        }
        if (lb >= le)
            continue;
        uint32_t qidx[1] = {q};
        uint64_t * out[1] = {a.partial + (((size_t)(f - a.slot_base) * a.nprobe + p) * a.seg_max + s) * a.k};
        __syncthreads();
        stage_queries<1>(a, qidx, qs);
        scan_rows<METRIC, 1, R>(a, (uint32_t)lb, (uint32_t)le, qs, lds_merge, out);
    }
}

}
