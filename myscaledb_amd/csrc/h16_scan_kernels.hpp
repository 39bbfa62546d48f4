// h16_scan_kernels.hpp -- the list scan of batched IVF searches over the index's fp16 SHADOW.
//
// The candidate pass of mfma_scan_kernels.hpp reads the f32 rows and splits them into bf16 pairs on the fly: 4 B per
// element from HBM, ~100 VALU ops per thread-step for the split, 3 MFMAs per product, a distance tile through LDS and
// a radix select per (query, slice).  This file is the same idea with the work moved to where it is free:
//
//   * BUILD TIME: every IVF list is stored a second time as fp16 (x * s rounded to nearest, s = one power of two per
//     index chosen from the largest |element|), in the register layout of the B operand of v_mfma_f32_32x32x16_f16:
//     a shadow block = 32 rows, per 16-element reduction step 1 KiB = 64 lanes x 16 B, lane (r, h) holding elements
//     16 s + 8 h .. + 7 of row r.  A wavefront streams its 32 rows with one fully coalesced global_load_dwordx4 per
//     step straight into the MFMA operand registers: 2 B per element from HBM, no LDS round trip, no conversion.
//     Lists are padded to whole blocks with zero rows (never offered).  +50 % index memory: 288 GB of HBM is what
//     makes that the right trade (1M x 768: 3.07 GB f32 + 1.6 GB shadow).
//   * PER SEARCH: the queries are rounded the same way once (h16_prep_queries_kernel: own power-of-two scale per
//     query) into the image of the A operand.  A work item is (list, tile of 32 * NCB probing queries): the tile is
//     loaded ONCE into LDS ([chunk][query][8 x 16 B], XOR-swizzled so the ds_read_b128 operand reads are bank-conflict
//     free) and stays there while the 8 wavefronts of the workgroup walk the list's blocks against it -- staging 64
//     reduction elements of the queries per barrier instead (the first version) cost more than the rows and the MFMAs
//     together (profiles/r02_h16_notes.txt).  Row chunks arrive through a 4-slot register ring per wavefront, chained
//     across block boundaries, the prefetch issue pinned with sched_barrier.
//   * ONE product per (row, query, step) instead of three: fp16 keeps 11 significant bits, so the single MFMA has
//     error <= 2^-10 |x||q| where bf16 x 3 had 3.1 * 2^-16 -- larger, and still far below the spread of real
//     distances (tests/test_gpu_parity.py::test_mfma_accumulation_error_bound_on_hardware measures it); the certificate of ivf_rerank_kernel takes the bound as a
//     parameter, so the returned ids / distance bits stay those of the canonical scan in every case.
//   * NO selection inside the scan.  A first launch (SAMPLE) computes the approximate distance of block 0 of every
//     probed list (32 rows per list, ~3 % of the rows) and writes them out; h16_sample_thr_kernel turns the m-th best
//     sample distance of a query into its cut `qthr` (about m * rows / sample rows of everything lie below it) and
//     appends the sample rows below the cut.  The main launch scans the other blocks and appends every row whose
//     approximate key is below the cut -- a compare + ballot per accumulator register, an atomic append for the few
//     dozen survivors per query.  Every row that is not appended has key >= qthr, which is exactly the `bound` the
//     certificate already knows (cand_select_kernel / ivf_rerank_kernel are reused unchanged).
//
// MFMA orientation: queries are the A operand (M, accumulator registers), rows the B operand (N, lanes), so a lane
// owns ONE row and 16 queries per accumulator: the row's norm, id and filter bit are per-lane scalars and the
// per-query constants come from LDS as broadcast float4 reads.
#pragma once

#include "mfma_scan_kernels.hpp"

namespace msvs
{

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int H_ROWS = 32;  // rows per shadow block (one wavefront's B operand)

/// LDS traffic of ONE wavefront: its ds operations execute in order, a wait + a compiler barrier is all a write -> read-by-another-lane
/// hand-over needs (bm25p_kernels.hpp has the same helper).
__device__ __forceinline__ void bp_wave_lds_fence_h16()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
constexpr int H_CHUNK = 64; // reduction elements per LDS stage = 4 MFMA steps = 128 B per query

struct H16Params
{
    const uint4 * H;          // shadow blocks: [block][step][64 lanes] x 16 B
    const uint32_t * hoff;    // [nlist + 1] first shadow block of list l
    uint32_t nks;             // 16-element steps per stored row = 4 nch (rows are zero padded to whole chunks)
    uint32_t nch;             // 64-element chunks per row / query = ceil(d / 64)
    const uint4 * Qh;         // query image: [nq][nch][8 pieces] x 16 B (piece p of chunk c = elements 64 c + 8 p .. + 7)
    const float2 * qinfo;     // [nq] {m2, qn}: L2 a = fma(m2, acc, |x|^2) + qn with m2 = -2 / (s_x s_q); IP a = m2 * acc
    const float * xnorm;      // [n] |x|^2 (approximate)
    const int64_t * list_off; // [nlist + 1] row range of list l in the f32 storage
    const uint32_t * ids;     // id of stored row r
    const uint64_t * alive;   // nullable filter bitmap over ids
    uint32_t nbits;
    const uint32_t * pairs;   // (query, probe) pairs grouped by list (IvfPlanParams)
    const uint32_t * pair_off;
    const uint32_t * work_off;
    const uint32_t * seg_blocks; // nullable (main launch): [0] = blocks per row segment chosen by the plan kernel (IvfPlanParams::seg_out)
    uint32_t nlist, nprobe, xcd_order;
    // main launch
    const uint32_t * qthr;    // [nq] cut (ordered distance word; 0xFFFFFFFF = none)
    uint32_t * qcnt;          // [nq] append cursors
    uint64_t * partial;       // [nq][cand_cap] appended keys (ordered distance word << 32 | row position)
    uint32_t cand_cap;
    // sample launch
    uint32_t * sample_out;    // [nq * nprobe][32] ordered distance word of row r of block 0 (0xFFFFFFFF = no row)
    uint32_t * sched;         // [8] work-queue cursors of this launch (zeroed by the caller)
    uint32_t group_appends;   // n: tiles of <= n queries append their survivors with one atomic per (wavefront, query) (0: one per record)
    uint32_t lazy_flush;      // 1: survivors stay in the wavefront's LDS stage from block to block and leave when it is full / at the end of the item
    uint64_t * stamps;        // nullable (option h16_stamps): [grid][H_STAMP_ITEMS][4] {item popped, tile resident, rows done, l << 32 | nvalid << 8}
                              // in wall_clock64 ticks (100 MHz), [grid][0][0] = items of the workgroup
};
constexpr uint32_t H_STAMP_ITEMS = 64;

/// Largest |element| of a table as float bits (NaN compares largest); max_bits zeroed by the caller.
static __global__ void absmax_kernel(const float4 * x, size_t n4, uint32_t * max_bits)
{
    uint32_t m = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    {
        const float4 v = x[i];
        const uint32_t a = __float_as_uint(v.x) & 0x7fffffffu, b = __float_as_uint(v.y) & 0x7fffffffu;
        const uint32_t c = __float_as_uint(v.z) & 0x7fffffffu, e = __float_as_uint(v.w) & 0x7fffffffu;
        m = max(max(m, a), max(b, max(c, e)));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        m = max(m, (uint32_t)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m)
        atomicMax(max_bits, m);
}

/// Smallest value of an array of non-negative floats as float bits (min_bits = 0x7f800000 from the caller).
static __global__ void min_f32_kernel(const float * v, size_t n, uint32_t * min_bits)
{
    uint32_t m = 0x7f800000u;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        m = min(m, __float_as_uint(v[i]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        m = min(m, (uint32_t)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0)
        atomicMin(min_bits, m);
}

__device__ __forceinline__ uint32_t pack_h2(const float a, const float b)
{
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    half2v t;
    t[0] = (_Float16)a; // round to nearest even, subnormals kept
    t[1] = (_Float16)b;
    return __builtin_bit_cast(uint32_t, t);
}

/// What the rounding really did to a table (round 4): rho = max over rows of |x' - x| / |x|, x' = fp16(x scale) / scale, as float
/// bits (atomicMax; zeroed by the caller).  The dot product of two rounded vectors is off by <dx, q> + <x, dq> + <dx, dq>, at most
/// (rho_x + rho_q + rho_x rho_q) |x||q| by Cauchy-Schwarz -- with the MEASURED rho of the stored rows and of the query image
/// instead of the worst case 2^-11 per element each (a rounding error is uniform in its interval and relative to the element's
/// binade: ~0.43 x 2^-11 in the root mean square over a row), the certificate's eps is less than half, still a proof
/// (set_error_model_h16).  One wavefront per row; differences and sums in double.
static __global__ __launch_bounds__(256) void h16_rho_kernel(const float * rows, size_t n, uint32_t ld, float scale, float inv_scale,
                                                             uint32_t * rho_bits)
{
    const size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (r >= n)
        return;
    const float * src = rows + r * ld;
    double num = 0.0, den = 0.0;
    for (uint32_t e = lane * 4; e < ld; e += 256)
    {
        const float4 v = *reinterpret_cast<const float4 *>(src + e);
        const float c[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const float back = (float)(_Float16)(c[i] * scale) * inv_scale; // (powers of two: the products are exact)
            const double dlt = (double)c[i] - (double)back;
            num += dlt * dlt;
            den += (double)c[i] * (double)c[i];
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
    {
        num += __shfl_xor(num, o);
        den += __shfl_xor(den, o);
    }
    if (lane == 0 && den > 0.0)
    {
        const float rho = (float)(sqrt(num / den) * 1.000001) ; // (double arithmetic + one rounding to float: far inside the margin)
        // (a look before the atomic: one atomic per row on ONE address serialised the launch -- 11.4 ms per 1M rows, round 4; a stale
        // look costs a needless atomic, never a lost maximum)
        const uint32_t bits = __float_as_uint(rho > 0.f ? rho : 0.f);
        if (bits > *reinterpret_cast<volatile uint32_t *>(rho_bits))
            atomicMax(rho_bits, bits);
    }
}

/// f32 list-major rows -> shadow blocks; one thread per 16-byte piece.
static __global__ void h16_build_kernel(const float * vecs, uint32_t ld, const int64_t * list_off,
                                        const uint32_t * blk_list, const uint32_t * hoff, uint32_t nks, float scale,
                                        uint4 * H, size_t piece0, size_t npieces)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npieces)
        return;
    i += piece0;
    const uint32_t per_blk = nks * 64;
    const uint32_t hb = (uint32_t)(i / per_blk), rem = (uint32_t)(i - (size_t)hb * per_blk);
    const uint32_t s = rem >> 6, lane = rem & 63;
    const uint32_t l = blk_list[hb];
    const int64_t row = list_off[l] + (int64_t)(hb - hoff[l]) * H_ROWS + (lane & 31);
    const uint32_t k0 = 16 * s + 8 * (lane >> 5);
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (row < list_off[l + 1])
    {
        const float * src = vecs + (size_t)row * ld;
        if (k0 < ld)
            v0 = *reinterpret_cast<const float4 *>(src + k0);
        if (k0 + 4 < ld)
            v1 = *reinterpret_cast<const float4 *>(src + k0 + 4);
    }
    H[i] = make_uint4(pack_h2(v0.x * scale, v0.y * scale), pack_h2(v0.z * scale, v0.w * scale),
                      pack_h2(v1.x * scale, v1.y * scale), pack_h2(v1.z * scale, v1.w * scale));
}

/// Queries -> fp16 image + per-query constants; one wavefront per query.
/// qnorm[q] = |q|^2: taken as given, or (compute_norm) computed here with row_sqnorm16_kernel's arithmetic -- 16 lanes,
/// fma chains, the DPP row tree -- so one launch serves where there were three (norms, a copy of them, this).  It is
/// overwritten with +inf when the query cannot be represented (NaN / inf / a scale outside 2^+-100): the certificate then
/// fails and the canonical fallback serves the query.
/// The small set-up work of a search that would otherwise be launches of its own rides along (aux, all nullable): the
/// one-list plan of a table pass (every query "probes" list 0 = rows [0, row_end): single_list_plan_kernel) and the zeroing of
/// the search's counters.
struct H16PrepAux
{
    uint32_t * pairs = nullptr;   // [nq] = 0 .. nq - 1
    int32_t * probes0 = nullptr;  // [nq] = 0
    int64_t * list_off = nullptr; // [2]
    uint32_t * pair_off = nullptr, * work_off = nullptr; // [2] each
    uint32_t * nfail = nullptr;   // the pass's fail counter = 0
    uint32_t row_end = 0, rows_per_block = 1, tq = 1;
    uint32_t * zero[2] = {nullptr, nullptr}; // two regions of 32-bit words to clear ...
    uint32_t nzero[2] = {0, 0};              // ... and their lengths
};

static __global__ void h16_prep_queries_kernel(const float * Q, uint32_t nq, uint32_t ld, uint32_t nch,
                                               float inv_sx, int ip, uint4 * Qh, float2 * qinfo, float * qnorm, int compute_norm,
                                               const H16PrepAux aux, float * qrho)
{
    const uint32_t q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    {
        const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
        for (int z = 0; z < 2; z++)
            for (uint32_t i = gid; i < aux.nzero[z]; i += gsz)
                aux.zero[z][i] = 0;
        if (aux.pairs && gid == 0)
        {
            aux.list_off[0] = 0;
            aux.list_off[1] = aux.row_end;
            aux.pair_off[0] = 0;
            aux.pair_off[1] = nq;
            aux.work_off[0] = 0;
            aux.work_off[1] = ((nq + aux.tq - 1) / aux.tq) * ((aux.row_end + aux.rows_per_block - 1) / aux.rows_per_block);
            *aux.nfail = 0;
        }
    }
    if (q >= nq)
        return;
    if (aux.pairs && lane == 0)
    {
        aux.pairs[q] = q;
        aux.probes0[q] = 0;
    }
    const float * src = Q + (size_t)q * ld;
    uint32_t m = 0;
    for (uint32_t e = lane * 4; e < ld; e += 256)
    {
        const float4 v = *reinterpret_cast<const float4 *>(src + e);
        m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), max(__float_as_uint(v.y) & 0x7fffffffu,
                max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu)));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        m = max(m, (uint32_t)__shfl_xor((int)m, o));
    const float maxabs = __uint_as_float(m);
    bool bad = !(maxabs < 3.0e38f);
    int ex = 0;
    if (maxabs > 0.f && !bad)
        (void)frexpf(maxabs, &ex); // maxabs = f * 2^ex, f in [0.5, 1): maxabs * 2^(14 - ex) < 2^14
    int sh = 14 - ex;
    if (sh > 100 || sh < -100)
    {
        bad = true;
        sh = sh > 0 ? 100 : -100;
    }
    const float sq = ldexpf(1.f, sh), inv_sq = ldexpf(1.f, -sh);
    double rnum = 0.0, rden = 0.0; // |q' - q|^2, |q|^2 of this lane's elements: the image's measured rounding error (h16_rho_kernel)
    for (uint32_t p = lane; p < nch * 8; p += 64)
    {
        const uint32_t k0 = p * 8;
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (k0 < ld)
            v0 = *reinterpret_cast<const float4 *>(src + k0);
        if (k0 + 4 < ld)
            v1 = *reinterpret_cast<const float4 *>(src + k0 + 4);
        const uint4 img = make_uint4(pack_h2(v0.x * sq, v0.y * sq), pack_h2(v0.z * sq, v0.w * sq),
                                     pack_h2(v1.x * sq, v1.y * sq), pack_h2(v1.z * sq, v1.w * sq));
        Qh[(size_t)q * nch * 8 + p] = img;
        if (qrho)
        {
            typedef _Float16 half2v __attribute__((ext_vector_type(2)));
            const float c[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            const uint32_t w[4] = {img.x, img.y, img.z, img.w};
#pragma unroll
            for (int i = 0; i < 8; i++)
            {
                const half2v h = __builtin_bit_cast(half2v, w[i >> 1]);
                const double dlt = (double)c[i] - (double)((float)h[i & 1] * inv_sq);
                rnum += dlt * dlt;
                rden += (double)c[i] * (double)c[i];
            }
        }
    }
    if (qrho)
    {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1)
        {
            rnum += __shfl_xor(rnum, o);
            rden += __shfl_xor(rden, o);
        }
        if (lane == 0) // (a query with non-finite or out-of-range elements is marked `bad` below and never certified)
            qrho[q] = rden > 0.0 && !bad ? (float)(sqrt(rnum / rden) * 1.000001) : 0.f;
    }
    float qn = 0.f;
    if (compute_norm)
    {
        if (lane < 16)
            for (uint32_t c = lane; c < ld / 4; c += 16)
            {
                const float4 v = *reinterpret_cast<const float4 *>(src + 4 * c);
                qn = fmaf(v.x, v.x, qn);
                qn = fmaf(v.y, v.y, qn);
                qn = fmaf(v.z, v.z, qn);
                qn = fmaf(v.w, v.w, qn);
            }
        qn = row16_tree_sum(qn);
    }
    if (lane == 0)
    {
        if (!compute_norm)
            qn = qnorm[q];
        const float unscale = inv_sx * inv_sq; // powers of two: exact
        qinfo[q] = ip ? make_float2(unscale, 0.f) : make_float2(-2.f * unscale, qn);
        if (bad || compute_norm)
            qnorm[q] = bad ? __uint_as_float(0x7f800000u) : qn;
    }
}

// ------------------------------------------------------------------------------------------ the scan
//
// Work item = (list, tile of <= 32 NCB probing queries).  The workgroup (H_NW wavefronts, one per CU) first makes the
// WHOLE query tile resident in LDS -- all d elements of its 32 NCB queries, 128 B per query and 64-element chunk,
// XOR-swizzled -- and then its wavefronts stream the list's shadow blocks independently: block after block, no
// barrier, no shared stage to wait for.  Per block a wavefront keeps H_RING - 1 chunks (4 KiB each) of rows in flight in a
// register ring (plain global_load_dwordx4 that hipcc counts: the waits are vmcnt(N), never vmcnt(0)), reads the A
// operands of the step from the resident tile and issues NCB MFMAs per 1 KiB of rows.
//
// How this shape was reached (profiles/r02_h16_notes.txt): a k-chunked tile staged per 128-row slice (every wavefront
// loading rows AND queries) drains the row queue at every stage wait (vmcnt retires in order); a dedicated loader
// wavefront fixes that but moves 16 KiB of queries per 16 KiB of rows through one wavefront's LDS-DMA stream
// (~25 GB/s per CU) and a barrier per chunk: 0.6-0.8 ms of the step went to staging even with the row stream or the
// matrix-core work removed.  With the tile resident the query bytes are read once per (list, tile) instead of once
// per slice and the steady state has no barrier at all.
// The price: a tile is at most what LDS holds (NCB <= 160 KiB / (d * 64 B): 3 column blocks = 96 queries at d = 768), so
// a list probed by more queries is streamed once per tile; the tiles of a list are consecutive work items handed to
// CUs of one XCD at the same time (per-XCD work queues), so the re-reads meet in that XCD's L2.
//
// SAMPLE launch: work item = (list, tile) as well, rows = block 0 only; wavefront w < ncb multiplies it by column
// block w and writes the 32 x 32 ordered distance words to a.sample_out.

constexpr int H_NW = 8;      // wavefronts per workgroup of the main launch
constexpr int H_RING = 4;    // row chunks per wavefront in registers (H_RING - 1 in flight + the one being multiplied)
constexpr int H_STAGE = 64;  // survivor records a wavefront stages in LDS before one round of atomics
constexpr uint32_t H_NONE = 0xFFFFFFFFu;

/// LDS bytes of the scan kernel for a tile of 32 * ncb queries.
inline size_t h16_lds_bytes(uint32_t ncb, uint32_t nch)
{
    const size_t tq = 32 * (size_t)ncb;
    return tq * nch * 128 + 4 * tq * 4 + (size_t)H_NW * H_STAGE * 12 + 16;
}

/// A wavefront's share of a work item: shadow blocks blk0, blk0 + stride, ... < nblk of the list (32 rows each) against
/// NCBI column blocks of the resident tile.  The register ring runs THROUGH the block boundaries when nch is a
/// multiple of H_RING (the chunks requested past the end of a block are the first chunks of the wavefront's next
/// block), so in steady state every request is a useful one; otherwise, and after the last block, the requests past
/// the end re-load the last chunk (harmless, L2 hits).
/// (Round 4 measured a dynamic hand-out of the blocks -- one counter per item, idle workgroups joining items in progress -- against
/// this static striding: 3 % slower without joiners, no faster with them; profiles/r04_scan_notes.txt.)
/// NRB = 2 (exhaustive batches: h16_flat_kernel): a wavefront walks TWO consecutive blocks at a time, so every A fragment read
/// from LDS feeds two MFMAs -- with hundreds of queries per row the pass is bound by the LDS reads of the A operands (1 KiB per
/// MFMA against 128 B/clk per CU: exactly the matrix pipe's rate), not by the row stream, which then comes out of L2.
/// The cut of a query as the stream's epilogue tests it.  The pass keeps a row when the ordered word of its key is below the cut
/// word: word(make_key(v)) < cut.  L2: make_key admits v < FLT_MAX (never NaN) and f2ord is monotone on the floats, so that is
/// v < c with c = the float whose ordered word is the cut, capped at FLT_MAX -- as a FLOAT comparison, provided the two orders
/// agree on the values that occur: they differ only on (-0, +0), and v = fl(fma(..) + qn) with qn = |q|^2 >= +0 is never -0 (a
/// sum rounds to -0 only from two -0 addends).  Cut words below f2ord(-inf) (0: "nothing passes", the padding queries of a short
/// tile) map to negative NaN patterns: every comparison false, as before.  IP keeps the integer test on the word itself.
template <int METRIC>
__device__ __forceinline__ uint32_t h16_stream_cut(const uint32_t cut)
{
    if (METRIC != M_L2)
        return cut;
    return cut > f2ord(3.402823466e+38f) ? __float_as_uint(3.402823466e+38f) : __float_as_uint(ord2f(cut));
}

template <int METRIC, int NCBI, int RING = H_RING, int NRB = 1>
__device__ __forceinline__ void h16_stream(const H16Params & a, const unsigned char * qt /* tile */,
                                           const uint32_t chunk_stride, const float * m2_s, const float * qn_s,
                                           const uint32_t * thr_s, const uint32_t * qrow_s, uint32_t * stage, const uint32_t lane,
                                           const uint32_t nch, const uint32_t hb_list /* first shadow block of the list */,
                                           const uint32_t blk0, const uint32_t stride, const uint32_t nblk,
                                           const int64_t lbeg, const int64_t lend, const uint32_t tile_queries = 0xFFFFFFFFu)
{
    if (blk0 >= nblk)
        return;
    // (uniform: see flush.  Not in the two-row-block form -- exhaustive batches, tiles of 64+ queries: the extra code cost the kernel
    // its last free registers, 54 scratch instructions around the MFMA loops, 5.9 -> 6.6 ms per 4096-query pass)
    const bool grouped = NRB == 1 && a.group_appends != 0 && tile_queries <= a.group_appends;
    uint32_t blk = blk0;
    const uint32_t r32 = lane & 31, h = lane >> 5;
    // operand read offsets of this lane inside a 128-byte query row: piece (2 j + h) ^ swizzle
    const uint32_t sw = (r32 >> 1) & 7;
    uint32_t aoff[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
        aoff[j] = r32 * 128 + (((2 * j + h) ^ sw) << 4);
    const u32x4 * const hbase = reinterpret_cast<const u32x4 *>(a.H) + lane;
    const size_t blk_pieces = (size_t)nch * 256; // 4 steps x 64 lanes per chunk
    const uint32_t last = nch - 1;
    const bool chain = nch % RING == 0;

    u32x4 ring[NRB][RING][4];
    auto load_chunk = [&](u32x4 (&b)[4], const u32x4 * p) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            b[j] = p[j * 64];
    };
    // row block rb of the group that starts at block g: g + rb, or (past the end of the list) g again -- loaded, never offered
    auto blk_ptr = [&](const uint32_t g, const int rb) {
        return hbase + (size_t)(hb_list + (g + rb < nblk ? g + rb : g)) * blk_pieces;
    };
    const u32x4 * hp[NRB];
#pragma unroll
    for (int rb = 0; rb < NRB; rb++)
    {
        hp[rb] = blk_ptr(blk, rb);
#pragma unroll
        for (int u = 0; u < RING - 1; u++)
            load_chunk(ring[rb][u], hp[rb] + (size_t)((uint32_t)u < last ? (uint32_t)u : last) * 256);
    }
    __builtin_amdgcn_sched_barrier(0);

    uint32_t cnt = 0; // records staged, wave-uniform (0 again after every block)
    auto flush = [&]() {
        // Round 6: the records of a tile that holds FEW probing queries (most tiles of a small batch; many once the probe pruning has
        // thinned the pairs) are grouped by query first -- one round of ballots per distinct query -- and every group takes ONE atomic:
        // its leader adds the group's count (the leaders of all groups in one instruction), the members take consecutive slots.
        // A returning atomic per record on one address serialises in L2: ~250 of them per query of a 32-query batch, from every
        // wavefront that scans the query's one list (the per-item stamps showed 17 us of "rows" for 48 KB per wavefront; the list
        // scan of a 16-query batch 45 -> 33 us).  Tiles of many queries keep one atomic per record: their addresses differ.
        const bool have = lane < cnt;
        const uint32_t q = have ? stage[2 * H_STAGE + lane] : 0u;
        uint32_t pos = 0;
        if (grouped)
        {
            uint64_t rem = __ballot(have);
            uint32_t leader = lane, rank = 0, count = 0;
            while (rem)
            {
                const int lead = __builtin_ctzll(rem);
                const uint32_t ql = (uint32_t)__builtin_amdgcn_readlane((int)q, lead);
                const uint64_t m = __ballot(have && q == ql);
                if (have && q == ql)
                {
                    leader = (uint32_t)lead;
                    rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    count = (uint32_t)__popcll(m);
                }
                rem &= ~m;
            }
            uint32_t base = 0;
            if (have && leader == lane)
                base = atomicAdd(&a.qcnt[q], count);
            pos = (uint32_t)__shfl((int)base, (int)leader) + rank;
        }
        else if (have)
            pos = atomicAdd(&a.qcnt[q], 1u);
        if (have && pos < a.cand_cap)
            a.partial[(size_t)q * a.cand_cap + pos] = (uint64_t)stage[H_STAGE + lane] << 32 | stage[lane];
        cnt = 0;
    };
    half8 afp[2][NCBI]; // NRB >= 2: the A fragments of the step in flight and of the next one (see `step`)
    if (NRB >= 2)
    {
#pragma unroll
        for (int cb = 0; cb < NCBI; cb++)
            afp[0][cb] = *reinterpret_cast<const half8 *>(qt + cb * 4096 + aoff[0]);
    }
    for (; blk < nblk; blk += stride)
    {
        const uint32_t nxt = blk + stride;
        const bool more = nxt < nblk;
        const bool has_next = chain && more;
        const u32x4 * hp_next[NRB];
        int64_t row[NRB];
        bool ok[NRB];
        float xn[NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; rb++)
        {
            hp_next[rb] = has_next ? blk_ptr(nxt, rb) : hp[rb];
            row[rb] = lbeg + (int64_t)(blk + rb) * H_ROWS + r32;
            ok[rb] = blk + rb < nblk && row[rb] < lend;
            xn[rb] = 0.f;
            if (ok[rb])
            {
                if (METRIC == M_L2)
                    xn[rb] = a.xnorm[row[rb]];
                if (a.alive)
                {
                    const uint32_t id = a.ids[row[rb]];
                    ok[rb] = id < a.nbits && ((a.alive[id >> 6] >> (id & 63)) & 1);
                }
            }
        }
        f32x16 acc[NRB][NCBI];
#pragma unroll
        for (int rb = 0; rb < NRB; rb++)
#pragma unroll
            for (int cb = 0; cb < NCBI; cb++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    acc[rb][cb][r] = 0.f;
        auto step = [&](const int u, const uint32_t c) {
            const unsigned char * qb = qt + (size_t)c * chunk_stride;
            if (NRB >= 2)
            {
                // exhaustive batches (two row blocks per wavefront): the A fragments of the NEXT reduction step (of the next chunk after
                // the last step of this one; of chunk 0 after the last chunk -- the tile is the same for every block) are requested
                // BEFORE the MFMAs of this step issue: two register sets, the order pinned with sched_group_barrier (left to itself
                // hipcc reads each step's fragments just in time into one set and every step opens on the LDS latency)
                const unsigned char * qn_ = qt + (size_t)(c + 1 < nch ? c + 1 : 0) * chunk_stride;
#pragma unroll
                for (int j = 0; j < 4; j++)
                {
#pragma unroll
                    for (int cb = 0; cb < NCBI; cb++)
                        afp[(j + 1) & 1][cb] = *reinterpret_cast<const half8 *>((j < 3 ? qb : qn_) + cb * 4096 + aoff[(j + 1) & 3]);
#pragma unroll
                    for (int rb = 0; rb < NRB; rb++)
                    {
                        const half8 bf = __builtin_bit_cast(half8, ring[rb][u][j]);
#pragma unroll
                        for (int cb = 0; cb < NCBI; cb++)
                            acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afp[j & 1][cb], bf, acc[rb][cb], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, NCBI, 0);       // the next step's LDS reads ...
                    __builtin_amdgcn_sched_group_barrier(0x008, NRB * NCBI, 0); // ... then this step's MFMAs
                }
                return;
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                half8 af[NCBI];
#pragma unroll
                for (int cb = 0; cb < NCBI; cb++)
                    af[cb] = *reinterpret_cast<const half8 *>(qb + cb * 4096 + aoff[j]);
#pragma unroll
                for (int rb = 0; rb < NRB; rb++)
                {
                    const half8 bf = __builtin_bit_cast(half8, ring[rb][u][j]);
#pragma unroll
                    for (int cb = 0; cb < NCBI; cb++)
                        acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cb], bf, acc[rb][cb], 0, 0, 0);
                }
            }
        };
        for (uint32_t c0 = 0; c0 < nch; c0 += RING)
        {
#pragma unroll
            for (int u = 0; u < RING; u++)
            {
                const uint32_t c = c0 + u;
                if (c >= nch)
                    break;
                // the chunk RING - 1 ahead: of this block, of the wavefront's next block, or a re-load of the last one.
                // sched_barrier: without a fence hipcc sinks the prefetch loads down to their first use (register
                // pressure heuristic) and the ring degenerates into load -> vmcnt(0) -> use
                const uint32_t pc = c + RING - 1;
#pragma unroll
                for (int rb = 0; rb < NRB; rb++)
                {
                    const u32x4 * src = pc < nch ? hp[rb] + (size_t)pc * 256
                                                 : (has_next ? hp_next[rb] + (size_t)(pc - nch) * 256 : hp[rb] + (size_t)last * 256);
                    load_chunk(ring[rb][(u + RING - 1) % RING], src);
                }
                __builtin_amdgcn_sched_barrier(0);
                step(u, c);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!chain && more)
        {
            // no chaining: restart the ring on the next block
#pragma unroll
            for (int rb = 0; rb < NRB; rb++)
            {
                const u32x4 * const nx = blk_ptr(nxt, rb);
#pragma unroll
                for (int u = 0; u < RING - 1; u++)
                    load_chunk(ring[rb][u], nx + (size_t)((uint32_t)u < last ? (uint32_t)u : last) * 256);
                hp[rb] = nx;
            }
        }
        else
        {
#pragma unroll
            for (int rb = 0; rb < NRB; rb++)
                hp[rb] = hp_next[rb];
        }

        // ---- epilogue: accumulator register i of column block cb = query 32 cb + (i & 3) + 8 (i >> 2) + 4 h, row r32.
        // Survivors are compacted per wavefront into an LDS stage (ballot + mbcnt) and appended to the queries' candidate
        // buffers in ONE parallel round of global atomics per <= 64 records, at the end of every block (a returning atomic
        // per passing register serialises one L2 round trip each; carrying the stage from block to block and flushing only
        // when it is full -- fewer, fuller rounds -- measured SLOWER: 0.545 against 0.476 ms on the bench step, round 4).
        // Round 6: the TEST of a register is three instructions -- fma, add, compare against the query's cut as a FLOAT (thr_s holds
        // h16_stream_cut: see there why that is the same decision as `ordered word of make_key < cut word`), a row that is not
        // offered carries a NaN norm -- and the key is made only for the (rare) registers that pass.  The exhaustive batches spent
        // 5 VALU instructions per MFMA here (SQ_INSTS_VALU 1153 M against 192 M MFMAs per 4096-query pass: 12 per register), exactly
        // what the matrix pipe's shadow can hide and no more.
#pragma unroll
        for (int rb = 0; rb < NRB; rb++)
        {
            const float xq = METRIC == M_L2 ? (ok[rb] ? xn[rb] : __builtin_nanf("")) : 0.f;
            const uint64_t okb = METRIC == M_L2 ? ~0ull : __ballot(ok[rb]);
#pragma unroll
        for (int cb = 0; cb < NCBI; cb++)
        {
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++)
            {
                const uint32_t q0 = 32 * cb + 8 * g4 + 4 * h;
                const float4 m2 = *reinterpret_cast<const float4 *>(&m2_s[q0]);
                const float4 qn = *reinterpret_cast<const float4 *>(&qn_s[q0]);
                const uint4 cut = *reinterpret_cast<const uint4 *>(&thr_s[q0]);
                const float m2v[4] = {m2.x, m2.y, m2.z, m2.w}, qnv[4] = {qn.x, qn.y, qn.z, qn.w};
                const uint32_t cutv[4] = {cut.x, cut.y, cut.z, cut.w};
#pragma unroll
                for (int e = 0; e < 4; e++)
                {
                    bool pass;
                    float v;
                    if (METRIC == M_L2)
                    {
                        v = __fadd_rn(fmaf(m2v[e], acc[rb][cb][4 * g4 + e], xq), qnv[e]);
                        pass = v < __uint_as_float(cutv[e]);
                    }
                    else
                    {
                        v = __fmul_rn(m2v[e], acc[rb][cb][4 * g4 + e]);
                        pass = v > -3.402823466e+38f && ~f2ord(v) < cutv[e];
                    }
                    const uint64_t mask = __ballot(pass) & okb;
                    if (mask)
                    {
                        const uint32_t np = __popcll(mask);
                        if (cnt + np > (uint32_t)H_STAGE)
                            flush();
                        if (pass && ok[rb])
                        {
                            const uint64_t key = make_key<METRIC>(v, (uint32_t)row[rb]);
                            const uint32_t at = cnt
                                + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                            stage[at] = (uint32_t)key;
                            stage[H_STAGE + at] = (uint32_t)(key >> 32);
                            stage[2 * H_STAGE + at] = qrow_s[q0 + e];
                        }
                        cnt += np;
                    }
                }
            }
        }
        }
        if (!a.lazy_flush)
            flush();
    }
    if (a.lazy_flush && cnt) // (exhaustive batches: the stage is emptied when it is full and at the end of the item's share)
        flush();
}

/// Next work item of this workgroup: per-XCD queues (a.sched[x] = cursor of the x-th eighth of the items), own XCD
/// first, then the others' leftovers.  Placement (block b on XCD b % 8) is a speed assumption only.
__device__ __forceinline__ uint32_t h16_next_item(uint32_t * sched, const uint32_t total, const uint32_t xcd)
{
    const uint32_t per = (total + 7) / 8;
    for (uint32_t t = 0; t < 8; t++)
    {
        const uint32_t x = (xcd + t) & 7;
        if (x * per >= total)
            continue;
        const uint32_t i = atomicAdd(&sched[x], 1u);
        if (i < per && x * per + i < total)
            return x * per + i;
    }
    return H_NONE;
}

/// Persistent workgroups pulling work items (list, tile) from the per-XCD queues.  An item that holds fewer probing queries than
/// the tile can (the last tile of a list; most tiles once the probe pruning has thinned the pairs) loads and multiplies only the
/// column blocks it has queries for (round 4: the per-item stamps of option h16_stamps showed 61 % of the bench step's items
/// holding <= 32 queries; 0.505 -> 0.475 ms).
template <int METRIC, int NCB>
__global__ __launch_bounds__(64 * H_NW) void h16_scan_kernel(const H16Params a)
{
    constexpr uint32_t TQ = 32 * NCB;
    constexpr uint32_t NW = H_NW;
    const uint32_t nch = a.nch;
    unsigned char * const tile = msvs_smem; // [chunk][query][8 x 16 B]
    float * const m2_s = reinterpret_cast<float *>(tile + (size_t)TQ * nch * 128);
    float * const qn_s = m2_s + TQ;
    uint32_t * const thr_s = reinterpret_cast<uint32_t *>(qn_s + TQ);
    uint32_t * const qrow_s = thr_s + TQ;
    uint32_t * const stage_s = qrow_s + TQ; // [NW][3][H_STAGE]
    uint32_t * const item_s = stage_s + NW * 3 * H_STAGE;

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t total = a.work_off[a.nlist];
    uint64_t * const stamp = a.stamps ? a.stamps + (size_t)blockIdx.x * H_STAMP_ITEMS * 4 : nullptr;
    uint32_t n_items = 0;
    for (;;)
    {
        __syncthreads(); // the previous work item is done with the tile, the tables and item_s
        if (tid == 0)
        {
            if (stamp && n_items >= 1 && n_items < H_STAMP_ITEMS)
                stamp[n_items * 4 + 2] = wall_clock64();
            *item_s = h16_next_item(a.sched, total, blockIdx.x & 7);
        }
        __syncthreads();
        const uint32_t w = *item_s;
        if (w == H_NONE)
            break;
        n_items++;
        if (stamp && tid == 0 && n_items < H_STAMP_ITEMS)
            stamp[n_items * 4 + 0] = wall_clock64();
        uint32_t lo = 0, hi = a.nlist;
        while (hi - lo > 1)
        {
            const uint32_t mid = (lo + hi) >> 1;
            if (a.work_off[mid] <= w)
                lo = mid;
            else
                hi = mid;
        }
        const uint32_t l = lo;
        const int64_t lbeg = a.list_off[l], lend = a.list_off[l + 1];
        const uint32_t nblk = a.hoff[l + 1] - a.hoff[l];
        const uint32_t pe = a.pair_off[l + 1], p0 = a.pair_off[l];
        // item t of the list = (row segment, tile): the tiles of a segment are consecutive items (they meet in one XCD's L2)
        const uint32_t t_in = w - a.work_off[l], ntiles = (pe - p0 + TQ - 1) / TQ;
        const uint32_t seg = t_in / ntiles, tidx = t_in - seg * ntiles;
        uint32_t b_first = 1, b_end = nblk; // (block 0 is the sample launch's)
        if (a.seg_blocks)
        {
            const uint32_t nseg = plan_nseg(nblk - 1, a.seg_blocks[0]);
            const uint32_t per = (nblk - 1 + nseg - 1) / nseg;
            b_first = 1 + seg * per;
            b_end = b_first + per < nblk ? b_first + per : nblk;
        }
        const uint32_t pb = p0 + tidx * TQ;
        const uint32_t nvalid = pe - pb < TQ ? pe - pb : TQ;
        // column blocks this item needs, and the rows of its tile in LDS
        const uint32_t ncb_e = (nvalid + 31) >> 5;
        const uint32_t tq_e = 32 * ncb_e;

        if (tid < TQ)
        {
            const bool v = tid < nvalid;
            const uint32_t qp = a.pairs[v ? pb + tid : pe - 1];
            const uint32_t q = qp / a.nprobe;
            qrow_s[tid] = q;
            const float2 qi = a.qinfo[q];
            m2_s[tid] = qi.x;
            qn_s[tid] = qi.y;
            thr_s[tid] = h16_stream_cut<METRIC>(v ? a.qthr[q] : 0u); // padding queries of a short tile never pass
        }
        __syncthreads();
        // the tile: piece p = (chunk, query < tq_e, slot) holds piece slot ^ swizzle(query) of the query's chunk; LDS keeps the
        // full tile's strides (compile-time operand offsets in the stream), an item with fewer column blocks leaves rows unwritten
        {
            const uint32_t npieces = tq_e * nch * 8;
            for (uint32_t p0 = tid; p0 < npieces; p0 += 4 * 64 * NW)
            {
                uint4 v[4]; // 4 loads in flight per thread; pieces past the end re-load the last one
                uint32_t at[4];
#pragma unroll
                for (int u = 0; u < 4; u++)
                {
                    const uint32_t pp = p0 + u * 64 * NW;
                    const uint32_t p = pp < npieces ? pp : npieces - 1;
                    const uint32_t slot = p & 7, qi = (p >> 3) % tq_e, c = (p >> 3) / tq_e;
                    v[u] = a.Qh[((size_t)qrow_s[qi] * nch + c) * 8 + (slot ^ ((qi >> 1) & 7))];
                    at[u] = ((c * TQ + qi) * 8 + slot) * 16;
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (p0 + u * 64 * NW < npieces)
                        *reinterpret_cast<uint4 *>(tile + at[u]) = v[u];
            }
        }
        __syncthreads();
        if (stamp && tid == 0 && n_items < H_STAMP_ITEMS)
        {
            stamp[n_items * 4 + 1] = wall_clock64();
            stamp[n_items * 4 + 3] = (uint64_t)l << 32 | nvalid << 8;
        }
        uint32_t * const stage = stage_s + wave * 3 * H_STAGE;
#define MSVS_H16_STREAM(N)                                                                                                         \
    h16_stream<METRIC, N>(a, tile, TQ * 128, m2_s, qn_s, thr_s, qrow_s, stage, lane, nch, a.hoff[l], b_first + wave, NW, b_end, lbeg, lend, nvalid)
        if constexpr (NCB == 1)
            MSVS_H16_STREAM(1);
        else if (ncb_e == 1)
            MSVS_H16_STREAM(1);
        else if constexpr (NCB == 2)
            MSVS_H16_STREAM(2);
        else if (ncb_e == 2)
            MSVS_H16_STREAM(2);
        else if constexpr (NCB == 3)
            MSVS_H16_STREAM(3);
        else if (ncb_e == 3)
            MSVS_H16_STREAM(3);
        else
            MSVS_H16_STREAM(4);
#undef MSVS_H16_STREAM
    }
    if (stamp && tid == 0)
        stamp[0] = n_items;
}

/// The sample launch: block 0 (<= 32 rows) of every probed list against the queries probing it, one wavefront per
/// (list, column block of 32 queries), no LDS: both operands come straight from memory in MFMA register order -- the
/// rows as in the main launch (1 KiB per step, coalesced), the queries as a gather (lane (q, h) reads 16 bytes of
/// query q's image; the four steps of a chunk use up the 128-byte line).  The plan behind pair_off / work_off is built
/// with T = 32 over the row range [list_off, list_off + 32).  Writes the 32 x 32 ordered distance words of the item
/// to a.sample_out[pair][row].
/// NQB = 2 (not instantiated any more): an item is TWO column blocks (64 probing queries; the plan is built with T = 64), every
/// row fragment feeds two MFMAs, a quarter less operand traffic -- measured SLOWER (80 against 73 us per 4096-query step: 240
/// VGPRs, and half of the second blocks are empty).  A short second block repeats the item's last pair and stores nothing.
template <int METRIC, int NQB>
__global__ __launch_bounds__(BLOCK) void h16_sample_kernel(const H16Params a)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r32 = lane & 31, h = lane >> 5;
    const uint32_t nch = a.nch, total = a.work_off[a.nlist];
    for (uint32_t w = blockIdx.x * 4 + wave; w < total; w += gridDim.x * 4)
    {
        uint32_t lo = 0, hi = a.nlist;
        while (hi - lo > 1)
        {
            const uint32_t mid = (lo + hi) >> 1;
            if (a.work_off[mid] <= w)
                lo = mid;
            else
                hi = mid;
        }
        const uint32_t l = lo;
        const int64_t lbeg = a.list_off[l], lend = a.list_off[l + 1];
        const uint32_t pe = a.pair_off[l + 1];
        const uint32_t pb = a.pair_off[l] + (w - a.work_off[l]) * 32 * NQB;
        const uint32_t nvalid = pe - pb < 32u * NQB ? pe - pb : 32u * NQB;
        // this lane's queries (A operand row r32 of each column block); short tiles repeat their last pair
        uint32_t qp[NQB];
        float2 qi[NQB];
        const u32x4 * ap[NQB];
#pragma unroll
        for (int b = 0; b < NQB; b++)
        {
            const uint32_t slot = 32u * b + r32;
            qp[b] = a.pairs[pb + (slot < nvalid ? slot : nvalid - 1)];
            const uint32_t q = qp[b] / a.nprobe;
            qi[b] = a.qinfo[q];
            ap[b] = reinterpret_cast<const u32x4 *>(a.Qh) + (size_t)q * nch * 8 + h;
        }
        const u32x4 * const bp = reinterpret_cast<const u32x4 *>(a.H) + (size_t)a.hoff[l] * nch * 256 + lane;
        const int64_t row = lbeg + r32;
        bool ok = row < lend;
        float xn = 0.f;
        if (ok)
        {
            if (METRIC == M_L2)
                xn = a.xnorm[row];
            if (a.alive)
            {
                const uint32_t id = a.ids[row];
                ok = id < a.nbits && ((a.alive[id >> 6] >> (id & 63)) & 1);
            }
        }
        f32x16 acc[NQB];
#pragma unroll
        for (int b = 0; b < NQB; b++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                acc[b][r] = 0.f;
        u32x4 ar[2][NQB][4], br[2][4];
        auto load = [&](const int s, const uint32_t c) {
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
#pragma unroll
                for (int b = 0; b < NQB; b++)
                    ar[s][b][j] = ap[b][(size_t)c * 8 + 2 * j];
                br[s][j] = bp[(size_t)c * 256 + j * 64];
            }
        };
        auto mul = [&](const int s) {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int b = 0; b < NQB; b++)
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, ar[s][b][j]),
                                                                    __builtin_bit_cast(half8, br[s][j]), acc[b], 0, 0, 0);
        };
        load(0, 0);
        const uint32_t last = nch - 1;
        for (uint32_t c = 0; c < nch; c += 2)
        {
            load(1, c + 1 < last ? c + 1 : last);
            __builtin_amdgcn_sched_barrier(0);
            mul(0);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 >= nch)
                break;
            load(0, c + 2 < last ? c + 2 : last);
            __builtin_amdgcn_sched_barrier(0);
            mul(1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // accumulator register i = query (i & 3) + 8 (i >> 2) + 4 h of the column block, row r32: the query's constants live
        // in the lane that fed it as A operand
#pragma unroll
        for (int b = 0; b < NQB; b++)
#pragma unroll
            for (int i = 0; i < 16; i++)
            {
                const int qidx = (i & 3) + 8 * (i >> 2) + 4 * (int)h; // lanes qidx and qidx + 32 hold the same query
                const float m2 = __shfl(qi[b].x, qidx), qn = __shfl(qi[b].y, qidx);
                const uint32_t pair = (uint32_t)__shfl((int)qp[b], qidx);
                const float v = METRIC == M_L2 ? __fadd_rn(fmaf(m2, acc[b][i], xn), qn) : __fmul_rn(m2, acc[b][i]);
                const uint64_t key = ok ? make_key<METRIC>(v, (uint32_t)row) : KEY_NONE;
                if (32u * b + (uint32_t)qidx < nvalid)
                    a.sample_out[(size_t)pair * H_ROWS + r32] = (uint32_t)(key >> 32);
            }
    }
}

/// Probe words of a sharded search: words[q][p] = the coarse pass's approximate distance word of probe p's centroid (0xFFFFFFFF: none).
static __global__ void gather_probe_words_kernel(const int32_t * probes, const uint32_t * coarse_words, uint32_t npad, uint32_t nprobe,
                                                 size_t n_pairs, uint32_t * words)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs)
        return;
    const int32_t l = probes[i];
    words[i] = l >= 0 ? coarse_words[(i / nprobe) * npad + (uint32_t)l] : 0xFFFFFFFFu;
}

/// PROBE PRUNING (L2 and cosine: round 3; inner product, probes computed on another rank, k > 64: round 4).  The oracle scans all nprobe lists of a query; most of them cannot hold one of its k nearest rows,
/// and that can be PROVED before the main launch: the k-th smallest approximate distance among the query's sample rows (real,
/// probed, unfiltered rows) plus eps is an upper bound U of its k-th best canonical distance; a row x of list l has
/// ||q - x|| >= ||q - c_l|| - r_l (triangle inequality, r_l = the list's radius: ivf_build_kernels.hpp), ||q - c_l||^2 is known
/// from the coarse pass to within eps_c.  A pair (q, l) with (sqrt(a_c - 2 eps_c) - r_l)^2 > U + 2 eps_x holds only rows whose
/// canonical distance is strictly beyond the k-th best: dropping it from the plan of the main launch cannot change the result
/// (the sample rows it contributed stay candidates: they are real rows of probed lists).  On data with cluster structure most
/// pairs go -- a list is then probed by a third of the queries, ONE tile instead of two, and the launch reads every list once:
/// 548 -> ~300 us per 4096-query step on the bench index; on iid data nothing is dropped and the second plan costs ~20 us.
struct H16Prune
{
    const uint32_t * coarse_words; // [nq][npad]: the coarse pass's approximate distance word of every centroid; nullptr: ...
    uint32_t npad;
    const float * probe_dis;       // pre-pruning only: [nq][nprobe] CANONICAL distances of the probes' centroids (the canonical coarse scan of
                                   // a small batch: merge_kernel) instead of approximate words -- no query norms needed
    const uint32_t * probe_words;  // ... or [nq][nprobe]: the same words gathered per probe (a sharded search: the coarse pass of a
                                   // query ran on another rank and its words came with the probe lists); both nullptr: no pruning
    const float * radius; // [nlist]
    const float * cnorm;  // [nlist] |c_l|^2 (cosine form only)
    int ip;               // 0: L2 index; 1: cosine index (unit rows and queries, the scan ranks by inner product); 2: inner-product
                          // index: <q, x> = <q, c> + <q, x - c> <= <q, c> + |q| r_l (Cauchy-Schwarz)
    __host__ __device__ bool on() const { return coarse_words || probe_words || probe_dis; } // (the second stage: it needs the centroid distances)
    const float * qnorm;  // |q|^2 (+inf: unusable)
    float xmax, cmax;     // max |x|^2 over the rows / the centroids
    float xmin;           // min |x|^2 over the rows (cosine pre-pruning)
    const float * Q;      // the (normalised) queries, row stride ldq floats (cosine pre-pruning: |q|^2 is taken from them)
    uint32_t ldq;
    double c_dot, c_norm, c_canon; // the shadow passes' error model: rows ...
    double c_dot_c;                // ... and centroids (their table has its own measured rounding error: set_error_model_h16)
    const float * qrho;            // nullable [nq]: the query image's measured rounding error, added as qrho_scale{,_c} * qrho[q]
    double qrho_scale, qrho_scale_c;
    __device__ double cd_x(uint32_t q) const { return qrho ? c_dot + qrho_scale * (double)qrho[q] : c_dot; }
    __device__ double cd_c(uint32_t q) const { return qrho ? c_dot_c + qrho_scale_c * (double)qrho[q] : c_dot_c; }
    uint32_t k;
    float * upre;              // nullable [nq]: h16_preprune_kernel leaves its upper bound of the query's k-th best (real) distance,
                               // slack included, rounded up (+inf: none); the second stage takes the smaller of its own bound and this
    int32_t * out_probes;      // [nq][nprobe]: the probes that survive (-1: dropped or absent)
    unsigned long long * stat; // nullable: [0] += pairs dropped, [1] += pairs
    int stat_all = 1;          // the second stage: 0 = a pre-pruning (here or on the rank the probes came from) has counted the pairs
};

/// PRE-PRUNING (round 4; L2 indexes, no filter): before a single sample row is scored, the list radius alone rules most pairs
/// out on data with cluster structure.  Every row of probed list p lies within ||q - c_p|| + r_p of the query, so a list with at
/// least k rows puts an upper bound U_p = (||q - c_p|| + r_p)^2 on the query's k-th best distance; U = min over such probes.  A pair
/// whose nearest possible row is beyond it -- (||q - c_l|| - r_l)^2 > U -- cannot hold one of the k nearest rows: it is dropped from
/// the plan of the SAMPLE launch already (and with it from the cut, the second pruning and the main launch; the canonical fallback
/// still walks every probed list).  ||q - c||^2 comes from the coarse pass's approximate word, widened by its error bound both ways;
/// the slack on the right covers the canonical arithmetic of the rows' distances.  One wavefront per query, lane = probe.
/// On SURVEY 8d's sigma-0.3 blobs 97 % of the pairs go before the sample launch (0.080 -> ~0.01 ms).
static __global__ __launch_bounds__(BLOCK) void h16_preprune_kernel(const int32_t * probes, const H16Prune pr, const int64_t * list_off,
                                                                     uint32_t nq, uint32_t nprobe, int32_t * out_probes)
{
    const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nq)
        return;
    const int32_t l = lane < nprobe ? probes[(size_t)q * nprobe + lane] : -1;
    double ub = 1e300, lb = 0.0; // this probe's upper bound of the k-th distance; lower bound of its rows' distances
    double sq = 0.0;             // an upper bound of |q|
    bool usable = pr.xmax < 1e30f && pr.cmax < 1e30f;
    const double sc = sqrt((double)pr.cmax * 1.001);
    double spread = 0.0; // cosine: how far the row norms are from each other
    if (pr.ip == 1)
    {
        // COSINE index: rows and queries are normalised (to within rounding; a zero vector stays zero), the coarse stage ranks the
        // centroids by <q, c>.  ||q - c||^2 = |q|^2 + |c|^2 - 2 <q, c> puts the list's rows between (||q - c|| - r)^2 and
        // (||q - c|| + r)^2 away from the query as for L2; the scan ranks by <q, x> = (|q|^2 + |x|^2 - ||q - x||^2) / 2, so a row that
        // is farther than another by more than the spread of the row norms (+ the arithmetic's slack) has the smaller product.
        double qn = 0.0;
        for (uint32_t e = lane; e < pr.ldq; e += 64)
        {
            const double v = (double)pr.Q[(size_t)q * pr.ldq + e];
            qn += v * v;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1)
            qn += __shfl_xor(qn, o);
        usable = usable && qn < 1e30 && qn == qn;
        sq = usable ? sqrt(qn) * (1.0 + 1e-12) : 0.0;
        spread = (double)pr.xmax * (1.0 + pr.c_norm) - (double)pr.xmin * (1.0 - pr.c_norm);
        if (!(spread >= 0.0))
            usable = false;
        double ipc = 0.0, e_ip = 0.0;
        bool have = false;
        if (pr.probe_dis)
        {
            ipc = l >= 0 ? (double)pr.probe_dis[(size_t)q * nprobe + lane] : 0.0;
            e_ip = (pr.c_canon + 4e-7) * sc * sq + 1e-30;
            have = l >= 0 && ipc == ipc && fabs(ipc) < 1e30;
        }
        else
        {
            const uint32_t cw = l >= 0 ? (pr.probe_words ? pr.probe_words[(size_t)q * nprobe + lane] : pr.coarse_words[(size_t)q * pr.npad + (uint32_t)l])
                                       : 0xFFFFFFFFu;
            ipc = (double)ord2f(~cw);
            e_ip = 2.0 * ((pr.cd_c(q) + pr.c_canon) * sc * sq + 1e-30);
            have = l >= 0 && cw != 0xFFFFFFFFu && ipc == ipc && fabs(ipc) < 1e30;
        }
        if (usable && have)
        {
            const double cn = (double)pr.cnorm[l], r = (double)pr.radius[l];
            const double hi2 = qn + cn * (1.0 + pr.c_norm) - 2.0 * (ipc - e_ip), lo2 = qn + cn * (1.0 - pr.c_norm) - 2.0 * (ipc + e_ip);
            if ((uint64_t)(list_off[l + 1] - list_off[l]) >= pr.k && r == r && hi2 == hi2)
            {
                const double hi = sqrt(hi2 > 0.0 ? hi2 : 0.0) * (1.0 + 1e-7) + r;
                ub = hi * hi * (1.0 + 1e-7);
            }
            if (lo2 > 0.0)
            {
                const double lo = sqrt(lo2) * (1.0 - 1e-7);
                if (lo > r)
                    lb = (lo - r) * (lo - r) * (1.0 - 1e-7);
            }
        }
    }
    else if (pr.probe_dis)
    {
        // canonical centroid distances: their rounding is all there is to widen; |q| <= ||q - c|| + |c| (the nearest probe's)
        const double dc2 = l >= 0 ? (double)pr.probe_dis[(size_t)q * nprobe + lane] : 1e300;
        double mn = dc2 == dc2 ? dc2 : 1e300;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1)
        {
            const double other = __shfl_xor(mn, o);
            mn = other < mn ? other : mn;
        }
        usable = usable && mn < 1e30 && mn >= 0.0;
        sq = usable ? sqrt(mn * (1.0 + 4.0 * (pr.c_canon + 4e-7))) * (1.0 + 1e-6) + sc : 0.0; // (mn is a canonical f32 value: widened by its own error before the root)
        if (usable && l >= 0 && dc2 == dc2 && dc2 >= 0.0 && dc2 < 1e30)
        {
            const double eps_c = (pr.c_canon + 4e-7) * (sqrt(dc2) + 2.0 * sc) * (sqrt(dc2) + 2.0 * sc) + 1e-30, r = (double)pr.radius[l];
            if ((uint64_t)(list_off[l + 1] - list_off[l]) >= pr.k && r == r)
            {
                const double hi = sqrt(dc2 + eps_c) * (1.0 + 1e-7) + r;
                ub = hi * hi * (1.0 + 1e-7);
            }
            if (dc2 - eps_c > 0.0)
            {
                const double lo = sqrt(dc2 - eps_c) * (1.0 - 1e-7);
                if (lo > r)
                    lb = (lo - r) * (lo - r) * (1.0 - 1e-7);
            }
        }
    }
    else
    {
        const uint32_t cw = l >= 0 ? (pr.probe_words ? pr.probe_words[(size_t)q * nprobe + lane] : pr.coarse_words[(size_t)q * pr.npad + (uint32_t)l])
                                   : 0xFFFFFFFFu;
        const float qn = pr.qnorm[q];
        usable = usable && qn < 1e30f;
        sq = sqrt((double)qn * 1.001);
        if (usable && l >= 0 && cw != 0xFFFFFFFFu)
        {
            const double eps_c = 2.0 * pr.cd_c(q) * sc * sq + pr.c_norm * (sc * sc + sq * sq) + (pr.c_canon + 4e-7) * (sc + sq) * (sc + sq) + 1e-30;
            const double ac = (double)ord2f(cw), r = (double)pr.radius[l];
            const double hi2 = ac + 2.0 * eps_c, lo2 = ac - 2.0 * eps_c;
            if ((uint64_t)(list_off[l + 1] - list_off[l]) >= pr.k && r == r && hi2 == hi2)
            {
                const double hi = sqrt(hi2 > 0.0 ? hi2 : 0.0) * (1.0 + 1e-7) + r;
                ub = hi * hi * (1.0 + 1e-7);
            }
            if (lo2 > 0.0)
            {
                const double lo = sqrt(lo2) * (1.0 - 1e-7);
                if (lo > r)
                    lb = (lo - r) * (lo - r) * (1.0 - 1e-7);
            }
        }
    }
    double U = ub;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
    {
        const double other = __shfl_xor(U, o);
        U = other < U ? other : U;
    }
    bool keep = true;
    const double sx = sqrt((double)pr.xmax * 1.001);
    const double slack = 2.0 * (pr.c_canon + 4e-7) * (sx + sq) * (sx + sq) + spread * (1.0 + 1e-6) + 1e-30; // canonical vs real distance, both sides
    if (usable && l >= 0 && U < 1e299)
        keep = !(lb > U + slack);
    if (lane < nprobe)
        out_probes[(size_t)q * nprobe + lane] = keep ? l : -1;
    if (pr.upre && lane == 0)
    {
        float u = __uint_as_float(0x7f800000u);
        if (usable && U < 1e299)
        {
            const double v = U + slack;
            u = v < 3.0e38 ? (float)v : u;
            if ((double)u < v)
                u = nextafterf(u, __uint_as_float(0x7f800000u));
        }
        pr.upre[q] = u;
    }
    if (pr.stat)
    {
        // [0] pairs dropped, [1] pairs looked at: every probe of the query here; a second stage behind this one adds what IT drops only
        // (H16Prune::stat_all = 0)
        const uint64_t dropped = __ballot(l >= 0 && !keep), all = __ballot(l >= 0);
        if (lane == 0 && all)
        {
            atomicAdd(pr.stat, (unsigned long long)__popcll(dropped));
            atomicAdd(pr.stat + 1, (unsigned long long)__popcll(all));
        }
    }
}

template <int NW>
__device__ inline void h16_sample_thr_wave(const uint32_t * src, const int32_t * qprobes, const int64_t * list_off, uint32_t nprobe,
                                           uint32_t target, uint32_t * qthr, uint32_t * qcnt, uint64_t * dst, uint32_t cap,
                                           uint32_t lane, uint32_t * hist, uint32_t q, const H16Prune & pr)
{
    const uint32_t n = nprobe * H_ROWS;
    uint32_t word[NW];
    int64_t lbeg[NW]; // start of the list word u belongs to (lane-dependent: i / 32)
    uint64_t rows = 0;
    uint32_t have = 0;
#pragma unroll
    for (int u = 0; u < NW; u++)
    {
        const uint32_t i = u * WAVE + lane;
        const int32_t l = i < n ? qprobes[i / H_ROWS] : -1;
        lbeg[u] = l >= 0 ? list_off[l] : 0;
        // the list's length: loaded by the first lane of the pair's 32, which also tells the others whether the list has rows
        // at all -- an empty list has no sample item and its 32 words were never written
        uint32_t len = 0;
        if (l >= 0 && (i & (H_ROWS - 1)) == 0)
        {
            const uint64_t len64 = (uint64_t)(list_off[l + 1] - lbeg[u]);
            rows += len64;
            len = len64 ? 1u : 0u;
        }
        len = (uint32_t)__shfl((int)len, (int)(lane & ~(H_ROWS - 1)));
        word[u] = len ? src[i] : 0xFFFFFFFFu;
        have += word[u] != 0xFFFFFFFFu;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
    {
        rows += (uint64_t)__shfl_xor((int)(uint32_t)rows, o) | ((uint64_t)__shfl_xor((int)(uint32_t)(rows >> 32), o) << 32);
        have += (uint32_t)__shfl_xor((int)have, o);
    }
    uint32_t m = rows ? (uint32_t)(((uint64_t)target * have + rows - 1) / rows) : 4u;
    m = m < 4 ? 4 : (m > 64 ? 64 : m);
    // the cut doubles as the pruning's upper bound when it can: at least k sample rows must lie at or below it (one selection, not
    // two).  Not when the query has fewer than k live sample rows (a selective filter: the cut would be "none", every probed row a
    // candidate) and not beyond k = 64 (a hybrid search's top-100: the cut would triple the candidates) -- the bound then takes a
    // selection of its own.
    // Nor when the sample is a small share of the probed rows (long lists): the k-th of S sample rows leaves k R / S rows below
    // the cut, and beyond twice the target the appends cost the main launch more than the second selection (12.5M x 768, 8 probes
    // of ~6000 rows: 1900 candidates per query instead of 760 -- the scan of a 64-query batch took twice as long).
    if (pr.on() && m < pr.k && have >= pr.k && pr.k <= 64 && (uint64_t)pr.k * rows <= 2ull * target * have)
        m = pr.k;
    uint32_t cut = target == 0 ? 0xFFFFFFFFu : wave_kth_word<NW>(word, m, hist, lane);
    if (pr.on())
    {
        // the k-th smallest sample word or a larger one (0xFFFFFFFF: fewer than k sample rows -- no bound, nothing is dropped)
        const uint32_t uw = target != 0 && m >= pr.k ? cut : wave_kth_word<NW>(word, pr.k, hist, lane);
        const int32_t l = lane < nprobe ? qprobes[lane] : -1;
        // the probe's centroid distance (L2) / product (cosine, inner product): the coarse pass's approximate word, or -- a small
        // batch -- the canonical value of the canonical coarse scan (cd_c = 0 then: only the canonical rounding is left)
        double cval = 0.0;
        bool cok = false;
        if (pr.probe_dis)
        {
            cval = l >= 0 ? (double)pr.probe_dis[(size_t)q * nprobe + lane] : 0.0;
            cok = l >= 0 && cval == cval && fabs(cval) < 1e30;
        }
        else
        {
            const uint32_t cw = l >= 0 ? (pr.probe_words ? pr.probe_words[(size_t)q * nprobe + lane] : pr.coarse_words[(size_t)q * pr.npad + (uint32_t)l])
                                       : 0xFFFFFFFFu;
            cok = cw != 0xFFFFFFFFu;
            cval = (double)ord2f(pr.ip ? ~cw : cw);
        }
        const double cdc = pr.probe_dis ? 0.0 : pr.cd_c(q);
        bool keep = true;
        const float qn = pr.qnorm[q];
        if (uw != 0xFFFFFFFFu && l >= 0 && qn < 1e30f && pr.xmax < 1e30f && pr.cmax < 1e30f)
        {
            const double sq = sqrt((double)qn * 1.001), sx = sqrt((double)pr.xmax * 1.001), sc = sqrt((double)pr.cmax * 1.001);
            if (!cok)
                ; // no coarse value for this probe: kept
            else if (pr.ip == 0)
            {
                const double eps_x = 2.0 * pr.cd_x(q) * sx * sq + pr.c_norm * (sx * sx + sq * sq) + (pr.c_canon + 4e-7) * (sx + sq) * (sx + sq) + 1e-30;
                const double eps_c = 2.0 * cdc * sc * sq + pr.c_norm * (sc * sc + sq * sq) + (pr.c_canon + 4e-7) * (sc + sq) * (sc + sq) + 1e-30;
                const double ak = (double)ord2f(uw), ac = cval;
                const double inner = ac - 2.0 * eps_c;
                if (inner > 0.0)
                {
                    const double dc = sqrt(inner) * (1.0 - 1e-7), r = (double)pr.radius[l];
                    // the k-th best canonical distance is at most ak + 2 eps_x (k sample rows) and at most the pre-pruning's bound
                    double kth = ak + 2.0 * eps_x;
                    if (pr.upre && (double)pr.upre[q] < kth)
                        kth = (double)pr.upre[q];
                    if (dc > r)
                        keep = !((dc - r) * (dc - r) * (1.0 - 1e-7) > kth);
                }
            }
            else if (pr.ip == 2)
            {
                // inner-product index: the words order inner products (larger is better).  >= k sample rows have an approximate
                // value >= ipk, hence a canonical one >= ipk - eps_x: the k-th best canonical value of the query is at least that.
                // A row x of list l has <q, x> = <q, c> + <q, x - c> <= <q, c> + |q| r_l, the coarse pass knows <q, c> to within
                // eps_c (twice: approximate -> canonical -> real), a canonical value exceeds the real one by <= c_canon |x||q| <= eps_x
                const double eps_x = (pr.cd_x(q) + pr.c_canon) * sx * sq + 1e-30, eps_c = (cdc + pr.c_canon + 4e-7) * sc * sq + 1e-30;
                const double ipk = (double)ord2f(~uw), ipc = cval;
                const double ub = ipc + 2.0 * eps_c + sq * (double)pr.radius[l] * (1.0 + 1e-6);
                keep = !(ub < ipk - 2.0 * eps_x);
            }
            else
            {
                // cosine index: the words order inner products (larger is better).  ||q - c||^2 = |q|^2 + |c|^2 - 2 <q, c> from the
                // coarse pass's <q, c> and the (f32, fma-accumulated: relative error c_norm) norms; a row x of the list has
                // ||q - x|| >= ||q - c|| - r_l, i.e. <q, x> <= (|q|^2 + |x|^2 - (||q - c|| - r_l)^2) / 2
                const double eps_x = (pr.cd_x(q) + pr.c_canon) * sx * sq + 1e-30, eps_c = (cdc + pr.c_canon + 4e-7) * sc * sq + 1e-30;
                const double ipk = (double)ord2f(~uw), ipc = cval;
                const double cn = (double)pr.cnorm[l];
                const double d2 = (double)qn * (1.0 - pr.c_norm) + cn * (1.0 - pr.c_norm) - 2.0 * (ipc + eps_c);
                if (d2 > 0.0)
                {
                    const double dc = sqrt(d2) * (1.0 - 1e-7), r = (double)pr.radius[l];
                    if (dc > r)
                    {
                        const double ub = 0.5 * (((double)qn + (double)pr.xmax) * (1.0 + pr.c_norm) - (dc - r) * (dc - r) * (1.0 - 1e-7));
                        keep = !(ub < ipk - 2.0 * eps_x);
                    }
                }
            }
        }
        if (lane < nprobe)
            pr.out_probes[(size_t)q * nprobe + lane] = keep ? l : -1;
        if (pr.stat)
        {
            const uint64_t dropped = __ballot(l >= 0 && !keep), all = __ballot(l >= 0);
            if (lane == 0)
            {
                atomicAdd(pr.stat, (unsigned long long)__popcll(dropped));
                if (pr.stat_all)
                    atomicAdd(pr.stat + 1, (unsigned long long)__popcll(all));
            }
        }
    }
    uint32_t count = 0;
#pragma unroll
    for (int u = 0; u < NW; u++)
    {
        const bool take = word[u] < cut;
        const uint64_t mask = __ballot(take);
        if (take)
        {
            const uint32_t pos = count + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (pos < cap)
                dst[pos] = (uint64_t)word[u] << 32 | (uint32_t)(lbeg[u] + (lane & (H_ROWS - 1)));
        }
        count += (uint32_t)__popcll(mask);
    }
    if (lane == 0)
    {
        *qthr = cut;
        *qcnt = count;
    }
}

static __global__ __launch_bounds__(BLOCK) void h16_sample_thr_wave_kernel(const uint32_t * sample, const int32_t * probes,
                                                                            const int64_t * list_off, uint32_t nq, uint32_t nprobe,
                                                                            uint32_t target, uint32_t * qthr, uint32_t * qcnt,
                                                                            uint64_t * partial, uint32_t cap, int radix,
                                                                            const H16Prune pr)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[BLOCK / WAVE][256];
    const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nq)
        return;
    uint32_t * hist = radix ? s_hist[threadIdx.x >> 6] : nullptr;
    const uint32_t * src = sample + (size_t)q * nprobe * H_ROWS;
    const int32_t * qp = probes + (size_t)q * nprobe;
    uint64_t * dst = partial + (size_t)q * cap;
    if (nprobe <= 8)
        h16_sample_thr_wave<4>(src, qp, list_off, nprobe, target, qthr + q, qcnt + q, dst, cap, lane, hist, q, pr);
    else if (nprobe <= 16)
        h16_sample_thr_wave<8>(src, qp, list_off, nprobe, target, qthr + q, qcnt + q, dst, cap, lane, hist, q, pr);
    else if (nprobe <= 32)
        h16_sample_thr_wave<16>(src, qp, list_off, nprobe, target, qthr + q, qcnt + q, dst, cap, lane, hist, q, pr);
    else
        h16_sample_thr_wave<32>(src, qp, list_off, nprobe, target, qthr + q, qcnt + q, dst, cap, lane, hist, q, pr);
}

static __global__ __launch_bounds__(BLOCK) void h16_sample_thr_kernel(const uint32_t * sample, const int32_t * probes,
                                                                       const int64_t * list_off, uint32_t nq,
                                                                       uint32_t nprobe, uint32_t target, uint32_t * qthr,
                                                                       uint32_t * qcnt, uint64_t * partial, uint32_t cap)
{
    const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nq)
        return;
    const uint32_t * src = sample + (size_t)q * nprobe * H_ROWS;
    const uint32_t n = nprobe * H_ROWS;
    // R and S
    uint64_t rows = 0;
    for (uint32_t p = lane; p < nprobe; p += 64)
    {
        const int32_t l = probes[(size_t)q * nprobe + p];
        if (l >= 0)
            rows += (uint64_t)(list_off[l + 1] - list_off[l]);
    }
    // a pair whose list is empty has no sample item: its words were never written
    auto word_of = [&](uint32_t i) -> uint32_t {
        const int32_t l = probes[(size_t)q * nprobe + i / H_ROWS];
        return l >= 0 && list_off[l + 1] > list_off[l] ? src[i] : 0xFFFFFFFFu;
    };
    uint32_t have = 0;
    for (uint32_t i = lane; i < n; i += 64)
        have += word_of(i) != 0xFFFFFFFFu;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
    {
        rows += (uint64_t)__shfl_xor((int)(uint32_t)rows, o) | ((uint64_t)__shfl_xor((int)(uint32_t)(rows >> 32), o) << 32);
        have += (uint32_t)__shfl_xor((int)have, o);
    }
    uint32_t m = rows ? (uint32_t)(((uint64_t)target * have + rows - 1) / rows) : 4u;
    m = m < 4 ? 4 : (m > 64 ? 64 : m);
    WaveTopK<1> top;
    top.init();
    for (uint32_t base = 0; base < n; base += 4 * WAVE)
    {
        uint64_t key[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            const uint32_t i = base + u * WAVE + lane;
            const uint32_t word = i < n ? word_of(i) : 0xFFFFFFFFu;
            key[u] = word == 0xFFFFFFFFu ? KEY_NONE : ((uint64_t)word << 32 | i);
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            top.offer(key[u], m, lane);
    }
    // target 0 (a test knob): no cut at all, every probed row becomes a candidate
    const uint32_t cut = top.thr == KEY_NONE || target == 0 ? 0xFFFFFFFFu : (uint32_t)(top.thr >> 32);
    uint32_t count = 0;
    for (uint32_t base = 0; base < n; base += WAVE)
    {
        const uint32_t i = base + lane;
        const int32_t l = i < n ? probes[(size_t)q * nprobe + i / H_ROWS] : -1;
        const uint32_t word = i < n ? word_of(i) : 0xFFFFFFFFu;
        const bool take = word < cut; // 0xFFFFFFFF (no row) is never below a cut
        const uint64_t mask = __ballot(take);
        if (take)
        {
            const uint32_t pos = count + __popcll(mask & ((1ull << lane) - 1));
            if (pos < cap)
                partial[(size_t)q * cap + pos] = (uint64_t)word << 32 | (uint32_t)(list_off[l] + (i & (H_ROWS - 1)));
        }
        count += __popcll(mask);
    }
    if (lane == 0)
    {
        qthr[q] = cut;
        qcnt[q] = count;
    }
}

// ------------------------------------------------------------------------------------------ FLAT tables: exhaustive batches
//
// A FLAT index (BASELINE configs[0]; the operating point of data without cluster structure; tryBruteForceSearch's resident
// form) is one list of n rows.  A batch scans its fp16 shadow with the list scan's machinery -- resident query tile, register
// ring, cut + append epilogue -- over work items (segment of H_FLAT_SEGB blocks, tile of 32 NCB queries), the tiles of a segment
// consecutive so that they meet in one XCD's L2: the table leaves HBM once per step however many tiles there are.  No plan,
// no pair lists: item -> (segment, tile) is a division.  The cut comes from a SAMPLE of <= 64 blocks spread evenly over the table
// (coarse_h16_kernel with a block stride: every approximate distance of the batch to the sample rows) -- the m-th smallest
// sample word of a query, m chosen so that ~`target` rows of the whole table lie below it (flat_cut_kernel); the sample rows are
// scanned again by the main launch like all the others.  Candidates -> cand_select -> canonical re-rank -> certificate ->
// canonical fallback as everywhere (table_candidate_pass).
//
// With hundreds of queries per row the pass is bound by the matrix pipe / the LDS reads that feed it, not by HBM: NRB = 2 row
// blocks per wavefront halve the LDS traffic per MFMA (h16_stream) -- for every tile of two column blocks or more.  A single
// 32-query tile streams the table once: HBM-bound, NRB = 1 with the deep ring.

constexpr uint32_t H_FLAT_SEGB = 64;      // blocks per segment: 2048 rows (3 MB of shadow at d = 768)
constexpr uint32_t H_FLAT_SAMPLE_BLK = 64; // sample blocks: 2048 rows, one register-resident selection per query

struct H16FlatParams
{
    uint32_t nq, nblk, n_rows; // queries of the batch; blocks and rows of the table
    uint32_t nseg, ntiles, segb; // segments of segb blocks each; tiles
    uint32_t rot;                // tile t starts rot * t rounds into its segment (0: every tile at the segment's first block)
};

/// One wavefront per query: the m-th smallest of its n_pad <= 2048 sample words becomes the cut; no candidate is appended here
/// (the main launch scans the sample rows again).  Fewer than m sample rows (a very selective filter): no cut.
static __global__ __launch_bounds__(BLOCK) void flat_cut_kernel(const uint32_t * sample, uint32_t nq, uint32_t n_pad, uint32_t m,
                                                                 uint32_t * qthr, uint32_t * qcnt)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[BLOCK / WAVE][256];
    const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nq)
        return;
    uint32_t word[32];
#pragma unroll
    for (int u = 0; u < 32; u++)
    {
        const uint32_t i = u * WAVE + lane;
        word[u] = i < n_pad ? sample[(size_t)q * n_pad + i] : 0xFFFFFFFFu;
    }
    const uint32_t cut = wave_kth_word<32>(word, m, s_hist[threadIdx.x >> 6], lane);
    if (lane == 0)
    {
        qthr[q] = cut;
        qcnt[q] = 0;
    }
}

template <int METRIC, int NCB, int NRB, int RING>
__global__ __launch_bounds__(64 * H_NW) void h16_flat_kernel(const H16Params a, const H16FlatParams f)
{
    constexpr uint32_t TQ = 32 * NCB;
    constexpr uint32_t NW = H_NW;
    const uint32_t nch = a.nch;
    unsigned char * const tile = msvs_smem; // [chunk][query][8 x 16 B]
    float * const m2_s = reinterpret_cast<float *>(tile + (size_t)TQ * nch * 128);
    float * const qn_s = m2_s + TQ;
    uint32_t * const thr_s = reinterpret_cast<uint32_t *>(qn_s + TQ);
    uint32_t * const qrow_s = thr_s + TQ;
    uint32_t * const stage_s = qrow_s + TQ; // [NW][3][H_STAGE]
    uint32_t * const item_s = stage_s + NW * 3 * H_STAGE;

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t total = f.nseg * f.ntiles;
    uint32_t resident = H_NONE; // the tile in LDS (a batch of one tile loads it once per workgroup, whatever the number of segments)
    for (;;)
    {
        __syncthreads(); // the previous work item is done with the tile, the tables and item_s
        if (tid == 0)
            *item_s = h16_next_item(a.sched, total, blockIdx.x & 7);
        __syncthreads();
        const uint32_t w = *item_s;
        if (w == H_NONE)
            break;
        const uint32_t seg = w / f.ntiles, tidx = w - seg * f.ntiles; // the tiles of a segment are consecutive items
        const uint32_t q0 = tidx * TQ;
        const uint32_t nvalid = f.nq - q0 < TQ ? f.nq - q0 : TQ;
        const uint32_t ncb_e = (nvalid + 31) >> 5;
        const uint32_t tq_e = 32 * ncb_e;
        if (tidx != resident)
        {
        resident = tidx;
        if (tid < TQ)
        {
            const bool v = tid < nvalid;
            const uint32_t q = v ? q0 + tid : f.nq - 1;
            qrow_s[tid] = q;
            const float2 qi = a.qinfo[q];
            m2_s[tid] = qi.x;
            qn_s[tid] = qi.y;
            thr_s[tid] = h16_stream_cut<METRIC>(v ? a.qthr[q] : 0u); // padding queries of a short tile never pass
        }
        __syncthreads();
        {
            const uint32_t npieces = tq_e * nch * 8;
            for (uint32_t p0 = tid; p0 < npieces; p0 += 4 * 64 * NW)
            {
                uint4 v[4];
                uint32_t at[4];
#pragma unroll
                for (int u = 0; u < 4; u++)
                {
                    const uint32_t pp = p0 + u * 64 * NW;
                    const uint32_t p = pp < npieces ? pp : npieces - 1;
                    const uint32_t slot = p & 7, qi = (p >> 3) % tq_e, c = (p >> 3) / tq_e;
                    v[u] = a.Qh[((size_t)qrow_s[qi] * nch + c) * 8 + (slot ^ ((qi >> 1) & 7))];
                    at[u] = ((c * TQ + qi) * 8 + slot) * 16;
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (p0 + u * 64 * NW < npieces)
                        *reinterpret_cast<uint4 *>(tile + at[u]) = v[u];
            }
        }
        __syncthreads();
        }
        uint32_t * const stage = stage_s + wave * 3 * H_STAGE;
        const uint32_t b0 = seg * f.segb, b1 = b0 + f.segb < f.nblk ? b0 + f.segb : f.nblk;
        // ROTATION (round 6, option flat_rot, OFF): the tiles of a segment run at the same time on the CUs of one XCD and ask for the
        // same lines at the same moment; starting tile t `rot * t` rounds (of NW * NRB blocks) into the segment, wrapping around,
        // was meant to turn the requests queued behind one fill into hits.  MEASURED SLOWER (6.23 -> 7.14 ms per 4096-query pass
        // over 1M x 768): walking the segment in step is what keeps its working set inside the XCD's L2 -- the knob stays for the
        // record (profiles/r06_flat_notes.txt).
        const uint32_t round_blk = NW * NRB, rounds = (b1 - b0 + round_blk - 1) / round_blk;
        const uint32_t r0 = f.rot ? (tidx * f.rot) % rounds : 0;
        const uint32_t bs = b0 + r0 * round_blk;
#define MSVS_H16_FLAT_STREAM(N)                                                                                                    \
    do                                                                                                                             \
    {                                                                                                                              \
        h16_stream<METRIC, N, RING, NRB>(a, tile, TQ * 128, m2_s, qn_s, thr_s, qrow_s, stage, lane, nch, 0, bs + wave * NRB,       \
                                         NW * NRB, b1, 0, (int64_t)f.n_rows, nvalid);                                              \
        if (r0)                                                                                                                    \
            h16_stream<METRIC, N, RING, NRB>(a, tile, TQ * 128, m2_s, qn_s, thr_s, qrow_s, stage, lane, nch, 0, b0 + wave * NRB,   \
                                             NW * NRB, bs, 0, (int64_t)f.n_rows, nvalid);                                          \
    } while (0)
        if constexpr (NCB == 1)
            MSVS_H16_FLAT_STREAM(1);
        else if (ncb_e == 1)
            MSVS_H16_FLAT_STREAM(1);
        else if constexpr (NCB == 2)
            MSVS_H16_FLAT_STREAM(2);
        else if (ncb_e == 2)
            MSVS_H16_FLAT_STREAM(2);
        else
            MSVS_H16_FLAT_STREAM(3);
#undef MSVS_H16_FLAT_STREAM
    }
}

// ------------------------------------------------------------------------------------------ coarse quantiser of a batch
//
// The centroid table through the same shadow: its ceil(nlist / 32) blocks are presented to h16_sample_kernel as G "lists"
// of one block each, every query "probing" all of them, so the kernel writes ALL approximate centroid distances of the
// batch (nq x nlist words) -- 6 GFLOP for 4096 x 1024 x 768, a few microseconds of MFMA, against 83 us for the split-bf16
// table pass with its per-slice selection.  coarse_select_kernel then keeps the kc best per query for the canonical re-rank.

/// The centroid table against the whole batch, two 32-centroid blocks x two 32-query blocks per wavefront (4 accumulators:
/// every operand fragment loaded feeds two MFMAs, half the L2 traffic of one-tile items; h16_sample_kernel on the trivial
/// plan took 44 us for 4096 x 1024 x 768, its operands being 96 KB per 32 x 32 tile).  No plan, no LDS: block g of the
/// shadow = centroids [32 g, 32 g + 32), query block b = queries [32 b, 32 b + 32).  Writes EVERY word of
/// sample_out[q][n_pad = 32 G] (0xFFFFFFFF for the padding rows of the last block): no memset.
/// blk_stride > 1 (the sample of a FLAT table: h16_flat_kernel's cut): block g of this launch is block g * blk_stride of the shadow,
/// its rows g * blk_stride * 32 .. of n_total; ids / alive (nullable): rows the filter drops get the word 0xFFFFFFFF.
template <int METRIC>
__global__ __launch_bounds__(BLOCK, 1) void coarse_h16_kernel(const uint4 * H, uint32_t nch, const uint4 * Qh, const float2 * qinfo,
                                                               const float * xnorm, uint32_t n_rows, uint32_t nq,
                                                               uint32_t * sample_out, uint32_t blk_stride = 1, uint32_t n_total = 0xFFFFFFFFu,
                                                               const uint32_t * ids = nullptr, const uint64_t * alive = nullptr,
                                                               uint32_t nbits = 0)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r32 = lane & 31, h = lane >> 5;
    const uint32_t G = (n_rows + H_ROWS - 1) / H_ROWS, n_pad = G * H_ROWS, QB = (nq + 31) / 32;
    const uint32_t G2 = (G + 1) / 2, QB2 = (QB + 1) / 2, total = G2 * QB2;
    for (uint32_t w = blockIdx.x * 4 + wave; w < total; w += gridDim.x * 4)
    {
        // consecutive wavefronts: the same two query blocks against successive centroid blocks
        const uint32_t g2 = w % G2, q2 = w / G2;
        const uint32_t gb[2] = {2 * g2, 2 * g2 + 1 < G ? 2 * g2 + 1 : 2 * g2};
        const uint32_t qb[2] = {2 * q2, 2 * q2 + 1 < QB ? 2 * q2 + 1 : 2 * q2};
        const u32x4 * ap[2];
        const u32x4 * bp[2];
        float2 qi[2];
#pragma unroll
        for (int t = 0; t < 2; t++)
        {
            const uint32_t q = qb[t] * 32 + r32 < nq ? qb[t] * 32 + r32 : nq - 1;
            qi[t] = qinfo[q];
            ap[t] = reinterpret_cast<const u32x4 *>(Qh) + (size_t)q * nch * 8 + h;
            bp[t] = reinterpret_cast<const u32x4 *>(H) + (size_t)gb[t] * blk_stride * nch * 256 + lane;
        }
        f32x16 acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int s = 0; s < 2; s++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    acc[t][s][r] = 0.f;
        u32x4 ar[2][2][4], br[2][2][4]; // [buffer][query block | centroid block][k step]
        auto load = [&](const int b, const uint32_t c) {
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                {
                    ar[b][t][j] = ap[t][(size_t)c * 8 + 2 * j];
                    br[b][t][j] = bp[t][(size_t)c * 256 + j * 64];
                }
        };
        auto mul = [&](const int b) {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int s = 0; s < 2; s++)
                        acc[t][s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, ar[b][t][j]),
                                                                           __builtin_bit_cast(half8, br[b][s][j]), acc[t][s], 0, 0, 0);
        };
        load(0, 0);
        const uint32_t last = nch - 1;
        for (uint32_t c = 0; c < nch; c += 2)
        {
            load(1, c + 1 < last ? c + 1 : last);
            __builtin_amdgcn_sched_barrier(0);
            mul(0);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 >= nch)
                break;
            load(0, c + 2 < last ? c + 2 : last);
            __builtin_amdgcn_sched_barrier(0);
            mul(1);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < 2; s++)
        {
            if (s == 1 && gb[1] == gb[0])
                break;
            const uint32_t row = gb[s] * H_ROWS + r32;                    // in this launch's numbering (the output slot)
            const uint32_t srow = gb[s] * blk_stride * H_ROWS + r32;      // in the table
            bool ok = row < n_rows && srow < n_total;
            const float xn = ok && METRIC == M_L2 ? xnorm[srow] : 0.f;
            if (ok && alive)
            {
                const uint32_t id = ids ? ids[srow] : srow;
                ok = id < nbits && ((alive[id >> 6] >> (id & 63)) & 1);
            }
#pragma unroll
            for (int t = 0; t < 2; t++)
            {
                if (t == 1 && qb[1] == qb[0])
                    break;
#pragma unroll
                for (int i = 0; i < 16; i++)
                {
                    // accumulator register i = query (i & 3) + 8 (i >> 2) + 4 h of the block, row r32 (as in h16_sample_kernel)
                    const int qidx = (i & 3) + 8 * (i >> 2) + 4 * (int)h;
                    const float m2 = __shfl(qi[t].x, qidx), qn = __shfl(qi[t].y, qidx);
                    const float v = METRIC == M_L2 ? __fadd_rn(fmaf(m2, acc[t][s][i], xn), qn) : __fmul_rn(m2, acc[t][s][i]);
                    const uint64_t key = ok ? make_key<METRIC>(v, srow) : KEY_NONE;
                    const uint32_t q = qb[t] * 32 + (uint32_t)qidx;
                    if (q < nq)
                        sample_out[(size_t)q * n_pad + row] = (uint32_t)(key >> 32);
                }
            }
        }
    }
}

/// The same table pass as a workgroup-tiled product (round 6): 128 queries x 128 centroids per workgroup of four wavefronts (2 x 2, each
/// 64 x 64 = four accumulators), BOTH operands staged once per 64-element chunk in LDS (double-buffered, one barrier per chunk).
/// coarse_h16_kernel's wavefront tile of 64 x 64 fetches 192 KB of operands for 6.3 MFLOP -- 196 MB out of L2 per 4096 x 1024 x 768
/// pass, and the pass runs at what L2 delivers (31.6 us; a 32 x 64 tile with three wavefronts per SIMD, 295 MB, ran 42 us: the
/// traffic, not the latency, is the limit).  A 128 x 128 tile moves 384 KB for 25 MFLOP: 98 MB.  Query chunks arrive with coalesced
/// loads (thread = (query, piece)) and take the list scan's XOR swizzle in LDS; the shadow's blocks are already in operand order and
/// are copied as they are.  Same words, same output layout as coarse_h16_kernel (every word of sample_out[q][32 G] written).
// (queries per workgroup: 64 NT, centroids per workgroup: 64 NS -- NT x NS blocks of 32 x 32 per wavefront)
template <int METRIC, int NT, int NS>
__global__ __launch_bounds__(256, 2) void coarse_gemm_kernel(const uint4 * H, uint32_t nch, const uint4 * Qh, const float2 * qinfo,
                                                             const float * xnorm, uint32_t n_rows, uint32_t nq, uint32_t * sample_out)
{
    constexpr uint32_t CG_TQ = 64 * NT, CG_TC = 64 * NS;
    __shared__ __attribute__((aligned(16))) uint4 a_s[2][CG_TQ * 8];      // [buffer][query][8 pieces, swizzled]
    __shared__ __attribute__((aligned(16))) uint4 b_s[2][(CG_TC / 32) * 256]; // [buffer][block][step][lane]
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r32 = lane & 31, h = lane >> 5;
    const uint32_t G = (n_rows + H_ROWS - 1) / H_ROWS, n_pad = G * H_ROWS;
    const uint32_t q_base = blockIdx.x * CG_TQ, g_base = blockIdx.y * (CG_TC / 32);
    const uint32_t wq = wave >> 1, wc = wave & 1; // this wavefront: query blocks NT wq ..; centroid blocks NS wc .. of the tile
    // load side: A -- thread = (query tid / 8 + 32 i, piece tid % 8); B -- thread = u32x4 tid of block i's chunk
    const uint32_t lq = tid >> 3, lp = tid & 7;
    const u32x4 * asrc[2 * NT];
    const u32x4 * bsrc[2 * NS];
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        if (i < 2 * NT)
        {
            const uint32_t q = q_base + lq + 32 * i < nq ? q_base + lq + 32 * i : nq - 1;
            asrc[i] = reinterpret_cast<const u32x4 *>(Qh) + (size_t)q * nch * 8 + lp;
        }
        if (i < 2 * NS)
        {
            const uint32_t g = g_base + i < G ? g_base + i : G - 1;
            bsrc[i] = reinterpret_cast<const u32x4 *>(H) + (size_t)g * nch * 256 + tid;
        }
    }
    u32x4 ar[2][2 * NT], br[2][2 * NS]; // two chunks in flight: chunk c + 2 is requested before chunk c is multiplied
    auto fetch = [&](const int r, const uint32_t c) {
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            if (i < 2 * NT)
                ar[r][i] = asrc[i][(size_t)c * 8];
            if (i < 2 * NS)
                br[r][i] = bsrc[i][(size_t)c * 256];
        }
    };
    auto put = [&](const int b, const int r) {
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            if (i < 2 * NT)
            {
                const uint32_t qq = lq + 32 * i;
                reinterpret_cast<u32x4 *>(a_s[b])[qq * 8 + (lp ^ ((qq >> 1) & 7))] = ar[r][i];
            }
            if (i < 2 * NS)
                reinterpret_cast<u32x4 *>(b_s[b])[i * 256 + tid] = br[r][i];
        }
    };
    f32x16 acc[NT][NS];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int s2 = 0; s2 < NS; s2++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                acc[t][s2][r] = 0.f;
    const uint32_t sw = (r32 >> 1) & 7;
    auto mul = [&](const int b) {
        const u32x4 * const al = reinterpret_cast<const u32x4 *>(a_s[b]);
        const u32x4 * const bl = reinterpret_cast<const u32x4 *>(b_s[b]);
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            half8 af[NT], bf[NS];
#pragma unroll
            for (int t = 0; t < 2; t++)
            {
                if (t < NT)
                    af[t] = __builtin_bit_cast(half8, al[(32 * NT * wq + 32 * t + r32) * 8 + ((2 * j + h) ^ sw)]);
                if (t < NS)
                    bf[t] = __builtin_bit_cast(half8, bl[(NS * wc + t) * 256 + j * 64 + lane]);
            }
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int s2 = 0; s2 < NS; s2++)
                    acc[t][s2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t], bf[s2], acc[t][s2], 0, 0, 0);
        }
    };
    const uint32_t last = nch - 1;
    fetch(0, 0);
    fetch(1, 1 < last ? 1 : last);
    put(0, 0);
    __syncthreads();
    // chunk c lives in LDS buffer c & 1; register set c & 1 holds chunk c + 1 at the top of iteration c (set (c + 1) & 1 ... c + 2)
    for (uint32_t c = 0; c < nch; c += 2)
    {
        fetch(0, c + 2 < last ? c + 2 : last);
        __builtin_amdgcn_sched_barrier(0);
        mul(0);
        if (c + 1 < nch)
            put(1, 1); // chunk c + 1 (the other buffer: its last readers passed the barrier that ended chunk c - 1)
        __syncthreads();
        if (c + 1 >= nch)
            break;
        fetch(1, c + 3 < last ? c + 3 : last);
        __builtin_amdgcn_sched_barrier(0);
        mul(1);
        if (c + 2 < nch)
            put(0, 0); // chunk c + 2
        __syncthreads();
    }
    float2 qi[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
    {
        const uint32_t q = q_base + 32 * NT * wq + 32 * t + r32;
        qi[t] = qinfo[q < nq ? q : nq - 1];
    }
#pragma unroll
    for (int s2 = 0; s2 < NS; s2++)
    {
        const uint32_t g = g_base + NS * wc + s2;
        if (g >= G)
            break;
        const uint32_t row = g * H_ROWS + r32;
        const bool ok = row < n_rows;
        const float xn = ok && METRIC == M_L2 ? xnorm[row] : 0.f;
#pragma unroll
        for (int t = 0; t < NT; t++)
        {
#pragma unroll
            for (int i = 0; i < 16; i++)
            {
                // accumulator register i = query (i & 3) + 8 (i >> 2) + 4 h of the block, row r32 (as in coarse_h16_kernel)
                const int qidx = (i & 3) + 8 * (i >> 2) + 4 * (int)h;
                const float m2 = __shfl(qi[t].x, qidx), qn = __shfl(qi[t].y, qidx);
                const float v = METRIC == M_L2 ? __fadd_rn(fmaf(m2, acc[t][s2][i], xn), qn) : __fmul_rn(m2, acc[t][s2][i]);
                const uint64_t key = ok ? make_key<METRIC>(v, row) : KEY_NONE;
                const uint32_t q = q_base + 32 * NT * wq + 32 * t + (uint32_t)qidx;
                if (q < nq)
                    sample_out[(size_t)q * n_pad + row] = (uint32_t)(key >> 32);
            }
        }
    }
}

/// The sample of a FLAT table for ONE query tile (<= 32 queries: the few-query path), with the cut: a workgroup per sample block,
/// its four wavefronts take every fourth 64-element chunk of the reduction (coarse_h16_kernel walks all of them in one wavefront: 12
/// dependent steps, 15 us for a launch of 32 wavefronts), the partial tiles meet in LDS; the last workgroup to finish selects every
/// query's cut (flat_cut_kernel's selection: one launch and its gap less).  The words differ from the scan's in the last bits (another
/// summation order): the cut is a threshold, whatever it is the scan reports it as the bound of what it dropped.
template <int METRIC>
__global__ __launch_bounds__(BLOCK) void flat_sample_few_kernel(const uint4 * H, uint32_t nch, const uint4 * Qh, const float2 * qinfo,
                                                                 const float * xnorm, uint32_t gs, uint32_t nq, uint32_t * sample_out,
                                                                 uint32_t blk_stride, uint32_t n_total, const uint32_t * ids,
                                                                 const uint64_t * alive, uint32_t nbits, uint32_t mth, uint32_t * qthr,
                                                                 uint32_t * qcnt, uint32_t * ticket)
{
    __shared__ float part_s[BLOCK / WAVE][16][WAVE];
    __shared__ __attribute__((aligned(16))) uint32_t hist_s[BLOCK / WAVE][256];
    __shared__ uint32_t last_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r32 = lane & 31, h = lane >> 5;
    const uint32_t g = blockIdx.x, n_pad = gs * H_ROWS;
    const uint32_t ql = r32 < nq ? r32 : nq - 1;
    const float2 qi = qinfo[ql];
    const u32x4 * const ap = reinterpret_cast<const u32x4 *>(Qh) + (size_t)ql * nch * 8 + h;
    const u32x4 * const bp = reinterpret_cast<const u32x4 *>(H) + (size_t)g * blk_stride * nch * 256 + lane;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++)
        acc[r] = 0.f;
    u32x4 ar[2][4], br[2][4];
    auto load = [&](const int b, const uint32_t c) {
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            ar[b][j] = ap[(size_t)c * 8 + 2 * j];
            br[b][j] = bp[(size_t)c * 256 + j * 64];
        }
    };
    auto mul = [&](const int b) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, ar[b][j]), __builtin_bit_cast(half8, br[b][j]), acc, 0, 0, 0);
    };
    if (wave < nch)
    {
        load(0, wave);
        for (uint32_t c = wave; c < nch; c += 2 * (BLOCK / WAVE))
        {
            const uint32_t c1 = c + BLOCK / WAVE, c2 = c + 2 * (BLOCK / WAVE);
            if (c1 < nch)
                load(1, c1);
            __builtin_amdgcn_sched_barrier(0);
            mul(0);
            __builtin_amdgcn_sched_barrier(0);
            if (c1 >= nch)
                break;
            if (c2 < nch)
                load(0, c2);
            __builtin_amdgcn_sched_barrier(0);
            mul(1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; r++)
        part_s[wave][r][lane] = acc[r];
    __syncthreads();
    if (wave == 0)
    {
        const uint32_t row = g * H_ROWS + r32;               // the output slot
        const uint32_t srow = g * blk_stride * H_ROWS + r32; // in the table
        bool ok = srow < n_total;
        const float xn = ok && METRIC == M_L2 ? xnorm[srow] : 0.f;
        if (ok && alive)
        {
            const uint32_t id = ids ? ids[srow] : srow;
            ok = id < nbits && ((alive[id >> 6] >> (id & 63)) & 1);
        }
#pragma unroll
        for (int i = 0; i < 16; i++)
        {
            // accumulator register i = query (i & 3) + 8 (i >> 2) + 4 h of the tile, row r32 (as in coarse_h16_kernel)
            const int qidx = (i & 3) + 8 * (i >> 2) + 4 * (int)h;
            const float a4 = __fadd_rn(__fadd_rn(part_s[0][i][lane], part_s[1][i][lane]), __fadd_rn(part_s[2][i][lane], part_s[3][i][lane]));
            const float m2 = __shfl(qi.x, qidx), qn = __shfl(qi.y, qidx);
            const float v = METRIC == M_L2 ? __fadd_rn(fmaf(m2, a4, xn), qn) : __fmul_rn(m2, a4);
            const uint64_t key = ok ? make_key<METRIC>(v, srow) : KEY_NONE;
            if ((uint32_t)qidx < nq)
                sample_out[(size_t)qidx * n_pad + row] = (uint32_t)(key >> 32);
        }
    }
    // the last workgroup selects the cuts
    __threadfence();
    __syncthreads();
    if (tid == 0)
        last_s = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!last_s)
        return;
    __threadfence();
    for (uint32_t q = wave; q < nq; q += BLOCK / WAVE)
    {
        uint32_t word[32];
#pragma unroll
        for (int u = 0; u < 32; u++)
        {
            const uint32_t i = u * WAVE + lane;
            word[u] = i < n_pad ? __builtin_nontemporal_load(&sample_out[(size_t)q * n_pad + i]) : 0xFFFFFFFFu;
        }
        const uint32_t cut = wave_kth_word<32>(word, mth, hist_s[wave], lane);
        if (lane == 0)
        {
            qthr[q] = cut;
            qcnt[q] = 0;
        }
    }
    if (tid == 0)
        *ticket = 0;
}

/// The trivial plan: pairs of list g = (query i, g) for every i; pair index = i * G + g.
static __global__ void coarse_plan_kernel(uint32_t nq, uint32_t G, uint32_t * pairs, uint32_t * pair_off, uint32_t * work_off)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)nq * G)
    {
        const uint32_t g = (uint32_t)(i / nq), q = (uint32_t)(i - (size_t)g * nq);
        pairs[i] = q * G + g;
    }
    if (i <= G)
    {
        pair_off[i] = (uint32_t)i * nq;
        work_off[i] = (uint32_t)i * ((nq + 31) / 32);
    }
}

/// One wavefront per query: the kc (<= 64) smallest of its n_pad sample words -> cand[q][kc] keys
/// (word << 32 | centroid), the largest value last; bound[q] = KEY_NONE (the certificate then uses that last
/// candidate as the cut).  Up to 2048 centroids the words sit in registers and wave_select_words picks them
/// (52 -> ~8 us for 4096 queries x 1024 centroids against inserting into a sorted wave list); beyond, the insertion loop.
template <int NW>
__device__ inline void coarse_select_wave(const uint32_t * src, uint32_t n_pad, uint32_t kc, uint64_t * out, uint32_t lane, uint32_t * hist,
                                          uint64_t * stage)
{
    uint32_t hi[NW], lo[NW];
#pragma unroll
    for (int u = 0; u < NW; u++)
    {
        const uint32_t i = u * WAVE + lane;
        hi[u] = i < n_pad ? src[i] : 0xFFFFFFFFu;
        lo[u] = hi[u] == 0xFFFFFFFFu ? 0xFFFFFFFFu : i;
    }
    wave_select_words<NW>(hi, lo, kc, out, lane, hist, stage, false);
}

static __global__ __launch_bounds__(BLOCK) void coarse_select_kernel(const uint32_t * sample, uint32_t nq, uint32_t n_pad, uint32_t kc,
                                                                     uint64_t * cand, uint64_t * bound, int wave_select)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[BLOCK / WAVE][256];
    const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nq)
        return;
    __shared__ uint64_t s_stage[BLOCK / WAVE][WAVE];
    uint32_t * hist = wave_select == 3 ? nullptr : s_hist[threadIdx.x >> 6]; // 3: the bitwise search (experiments)
    uint64_t * stage = s_stage[threadIdx.x >> 6];
    const uint32_t * src = sample + (size_t)q * n_pad;
    uint64_t * dst = cand + (size_t)q * kc;
    if (lane == 0)
        bound[q] = KEY_NONE;
    if (wave_select && n_pad <= 32 * WAVE)
    {
        if (n_pad <= 4 * WAVE)
            coarse_select_wave<4>(src, n_pad, kc, dst, lane, hist, stage);
        else if (n_pad <= 8 * WAVE)
            coarse_select_wave<8>(src, n_pad, kc, dst, lane, hist, stage);
        else if (n_pad <= 16 * WAVE)
            coarse_select_wave<16>(src, n_pad, kc, dst, lane, hist, stage);
        else
            coarse_select_wave<32>(src, n_pad, kc, dst, lane, hist, stage);
        return;
    }
    WaveTopK<1> top;
    top.init();
    for (uint32_t base = 0; base < n_pad; base += 4 * WAVE)
    {
        uint64_t key[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            const uint32_t i = base + u * WAVE + lane;
            const uint32_t word = i < n_pad ? src[i] : 0xFFFFFFFFu;
            key[u] = word == 0xFFFFFFFFu ? KEY_NONE : ((uint64_t)word << 32 | i);
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            top.offer(key[u], kc, lane);
    }
    top.store(dst, kc, lane);
}

/// coarse_tail_kernel, a query whose band cannot be formed, served by its own wavefront (`slow_inline`): the canonical distance of EVERY
/// centroid -- 16 lanes per row, four rows per step, the arithmetic and order of scan_rows -- replaces the approximate word of that
/// centroid in the query's row of `words` (an exact word is inside every error bound its later readers assume), and the selection
/// that picked the candidates picks the k probes out of them: smallest (distance, id) first, the oracle's order of keys (ties inside
/// wave_select_words go by ascending slot = id).  ~0.25 ms for that one wavefront over 1024 centroids of 768 dimensions -- the host
/// takes this form only while the index has not queued a query for a while (the stamp in pinned memory: coarse_tail_kernel); it saves
/// the three launches of the queue's chain on every search that has none.
template <int METRIC>
__device__ inline void coarse_exact_rows(const RerankParams & a, const uint32_t q, uint32_t * words /* [n_pad] of this query */,
                                         const uint32_t n_rows, const uint32_t lane)
{
    const uint32_t grp = lane >> 4, g = lane & 15, ld4 = a.ld4;
    const uint32_t jfull = ld4 >> 4, jtail = ld4 & 15;
    const float4 * const qrow = a.Q + (size_t)q * ld4 + g;
#pragma unroll 1
    for (uint32_t r0 = 0; r0 < n_rows; r0 += 4)
    {
        const uint32_t row = r0 + grp;
        const bool rv = row < n_rows;
        const float4 * yrow = a.Y + (size_t)(rv ? row : n_rows - 1) * ld4 + g;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t j = 0;
        for (; j + 6 <= jfull; j += 6)
        {
            float4 y[6];
#pragma unroll
            for (int v = 0; v < 6; v++)
                y[v] = yrow[(j + v) * 16];
#pragma unroll
            for (int v = 0; v < 6; v++)
                canonical_update<METRIC>(acc, qrow[(j + v) * 16], y[v]);
        }
        for (; j < jfull; j++)
            canonical_update<METRIC>(acc, qrow[j * 16], yrow[j * 16]);
        if (g < jtail)
            canonical_update<METRIC>(acc, qrow[jfull * 16], yrow[jfull * 16]);
        float sum = __fadd_rn(__fadd_rn(acc.x, acc.y), __fadd_rn(acc.z, acc.w));
        sum = row16_tree_sum(sum);
        if (g == 0 && rv)
            words[row] = (uint32_t)(make_key<METRIC>(sum, row) >> 32);
    }
    // the words go back in through the same loads that read their approximate versions a moment ago: written back, the CU's vector
    // cache dropped
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

/// Round 6 -- the coarse quantiser's selection AND band re-rank in one launch, one WAVEFRONT per query, no workgroup barrier:
/// coarse_select_kernel (15 us per 4096 queries) + ivf_rerank_kernel in band mode (33 us: a 256-thread block per query, eight
/// barriers around three or four canonical rows) were two launches whose cost is the latency chain of a block, not work.
///   1. the kc <= 64 smallest of the query's approximate centroid words (coarse_select_wave, wave-private LDS), one per lane;
///   2. ivf_rerank_kernel's band rule on them: with a_(k), a_(k+1) the k-th / (k+1)-th smallest approximate values and eps the error
///      bound, a candidate with a + 2 eps < a_(k+1) is certainly IN (it takes the slot of its approximate rank), one with
///      a - 2 eps > a_(k) certainly OUT, and so is every centroid that is not a candidate when the largest candidate value
///      - 2 eps > a_(k); the band in between is evaluated canonically -- 16 lanes per row, four rows at a time, the arithmetic and
///      order of scan_rows -- and fills the remaining slots in exact order;
///   3. a query whose band cannot be formed (fewer than k + 1 candidates, non-finite norms, the margin test fails) leaves its
///      candidates in cand / bound and its number on slowq: ivf_rerank_kernel serves that list (RerankParams::qmap) as before.
/// Same probes as the two-launch form, slot by slot (test_small_batches... / the coarse parity tests compare both).
template <int METRIC>
__global__ __launch_bounds__(BLOCK) void coarse_tail_kernel(const uint32_t * sample, uint32_t nq, uint32_t n_pad, const RerankParams a,
                                                            uint64_t * cand, uint64_t * bound, uint32_t * slowq, uint32_t * nslow,
                                                            uint32_t n_rows, uint32_t * slow_stamp, uint32_t seq, int slow_inline)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[BLOCK / WAVE][256];
    __shared__ uint64_t s_stage[BLOCK / WAVE][WAVE];
    __shared__ uint64_t s_cand[BLOCK / WAVE][WAVE];
    __shared__ uint64_t s_keys[BLOCK / WAVE][WAVE];
    __shared__ uint32_t s_band[BLOCK / WAVE][WAVE];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t q = blockIdx.x * (BLOCK / WAVE) + wave;
    if (q >= nq)
        return;
    const uint32_t kc = a.kc, k = a.k, ld4 = a.ld4;
    uint64_t * const cd = s_cand[wave];
    cd[lane] = KEY_NONE;
    bp_wave_lds_fence_h16();
    const uint32_t * src = sample + (size_t)q * n_pad;
    if (n_pad <= 4 * WAVE)
        coarse_select_wave<4>(src, n_pad, kc, cd, lane, s_hist[wave], s_stage[wave]);
    else if (n_pad <= 8 * WAVE)
        coarse_select_wave<8>(src, n_pad, kc, cd, lane, s_hist[wave], s_stage[wave]);
    else if (n_pad <= 16 * WAVE)
        coarse_select_wave<16>(src, n_pad, kc, cd, lane, s_hist[wave], s_stage[wave]);
    else
        coarse_select_wave<32>(src, n_pad, kc, cd, lane, s_hist[wave], s_stage[wave]);
    bp_wave_lds_fence_h16();
    const uint64_t mine = lane < kc ? cd[lane] : KEY_NONE;
    uint32_t rank = 0; // of this candidate among all of them by approximate key (ties by slot)
    for (uint32_t j = 0; j < kc; j++)
    {
        const uint64_t kj = cd[j];
        rank += kj < mine || (kj == mine && j < lane) ? 1u : 0u;
    }
    const uint64_t m_k = __ballot(mine != KEY_NONE && rank == k - 1), m_k1 = __ballot(mine != KEY_NONE && rank == k);
    const uint64_t key_k = m_k ? readlane64(mine, __builtin_ctzll(m_k)) : KEY_NONE;
    const uint64_t key_k1 = m_k1 ? readlane64(mine, __builtin_ctzll(m_k1)) : KEY_NONE;
    const float qn = a.qnorm[q];
    bool band_ok = key_k != KEY_NONE && key_k1 != KEY_NONE && qn < 1e30f && a.xmax < 1e30f && kc > k;
    double eps2 = 0.0, ak = 0.0, ak1 = 0.0;
    if (band_ok)
    {
        eps2 = 2.0 * rerank_eps<METRIC>(a, sqrt((double)a.xmax * 1.001), sqrt((double)qn * 1.001), q);
        ak = (double)key_value<METRIC>(key_k);
        ak1 = (double)key_value<METRIC>(key_k1);
        const uint64_t last = cd[kc - 1]; // the candidates' largest approximate value comes last; KEY_NONE: every centroid is a candidate
        if (last != KEY_NONE)
        {
            const double al = (double)key_value<METRIC>(last);
            band_ok = METRIC == M_L2 ? (al - eps2 > ak) : (al + eps2 < ak);
        }
    }
    if (!band_ok) // (uniform over the wavefront)
    {
        // the host reads the stamp without synchronisation before its NEXT searches of this index: which form of the queue they take
        if (slow_stamp && lane == 0)
        {
            __hip_atomic_store(slow_stamp, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(slow_stamp + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (slow_inline)
        {
            if (lane == 0 && a.stat_fail)
                atomicAdd(a.stat_fail, 1ull);
            uint32_t * const words = const_cast<uint32_t *>(src);
            coarse_exact_rows<METRIC>(a, q, words, n_rows, lane);
            cd[lane] = KEY_NONE;
            bp_wave_lds_fence_h16();
            if (n_pad <= 4 * WAVE)
                coarse_select_wave<4>(src, n_pad, k, cd, lane, s_hist[wave], s_stage[wave]);
            else if (n_pad <= 8 * WAVE)
                coarse_select_wave<8>(src, n_pad, k, cd, lane, s_hist[wave], s_stage[wave]);
            else if (n_pad <= 16 * WAVE)
                coarse_select_wave<16>(src, n_pad, k, cd, lane, s_hist[wave], s_stage[wave]);
            else
                coarse_select_wave<32>(src, n_pad, k, cd, lane, s_hist[wave], s_stage[wave]);
            bp_wave_lds_fence_h16();
            if (lane < k)
            {
                const uint64_t key = cd[lane];
                a.out_probes[(size_t)q * k + lane] = key == KEY_NONE ? -1 : (int32_t)(a.ids ? a.ids[(uint32_t)key] : (uint32_t)key);
            }
            return;
        }
        if (lane < kc)
            cand[(size_t)q * kc + lane] = mine;
        if (lane == 0)
        {
            bound[q] = KEY_NONE;
            slowq[atomicAdd(nslow, 1u)] = q;
        }
        return;
    }
    bool in = false, mid = false;
    if (mine != KEY_NONE)
    {
        const double aj = (double)key_value<METRIC>(mine);
        in = METRIC == M_L2 ? (aj + eps2 < ak1) : (aj - eps2 > ak1);
        const bool out = METRIC == M_L2 ? (aj - eps2 > ak) : (aj + eps2 < ak);
        mid = !in && !out;
        if (in) // "in" is monotone in the approximate value: the certainly-in rows are the ranks 0 .. n_in - 1
            a.out_probes[(size_t)q * k + rank] = (int32_t)(a.ids ? a.ids[(uint32_t)mine] : (uint32_t)mine);
    }
    const uint32_t n_in = (uint32_t)__popcll(__ballot(in));
    const uint64_t mb = __ballot(mid);
    const uint32_t nb = (uint32_t)__popcll(mb);
    if (mid)
        s_band[wave][__builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, 0u))] = (uint32_t)mine;
    s_keys[wave][lane] = KEY_NONE;
    bp_wave_lds_fence_h16();
    // the band's rows, canonically: 16 lanes per row (scan_rows' arithmetic and order), four rows per round
    const uint32_t grp = lane >> 4, g = lane & 15;
    const uint32_t jfull = ld4 >> 4, jtail = ld4 & 15;
    const float4 * const qrow = a.Q + (size_t)q * ld4 + g;
    for (uint32_t c0 = 0; c0 < nb; c0 += 4)
    {
        const uint32_t c = c0 + grp;
        if (c < nb)
        {
            const uint32_t pos = s_band[wave][c];
            const float4 * yrow = a.Y + (size_t)pos * ld4 + g;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            uint32_t j = 0;
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
            float sum = __fadd_rn(__fadd_rn(acc.x, acc.y), __fadd_rn(acc.z, acc.w));
            sum = row16_tree_sum(sum);
            if (g == 0)
                s_keys[wave][c] = make_key<METRIC>(sum, a.ids ? a.ids[pos] : pos);
        }
    }
    bp_wave_lds_fence_h16();
    // the band's rows fill the slots the certainly-in rows left, in exact order (ties by their slot among the candidates)
    const uint64_t kb = lane < nb ? s_keys[wave][lane] : KEY_NONE;
    if (kb != KEY_NONE)
    {
        uint32_t rb = 0;
        for (uint32_t j = 0; j < nb; j++)
        {
            const uint64_t kj = s_keys[wave][j];
            rb += kj < kb || (kj == kb && j < lane) ? 1u : 0u;
        }
        if (n_in + rb < k)
            a.out_probes[(size_t)q * k + n_in + rb] = (int32_t)(uint32_t)kb;
    }
}

}

