// h16r_scan_kernels.hpp -- the list scan over the fp16 shadow with the QUERIES in registers and the ROWS through LDS.
//
// h16_scan_kernel (h16_scan_kernels.hpp) keeps a tile of 32 NCB queries resident in LDS and lets every wavefront stream its
// own shadow blocks straight into MFMA operand registers.  LDS bounds the tile: 96 queries at d = 768, 32 from d = 1088 on,
// so a list probed by n queries is streamed ceil(n / tile) times -- at 4096 queries per step (128 probing queries per list
// on the bench index) every list is read twice, at d = 1536 once per 32 queries, and the time of the launch is
// proportional to the number of (list, tile) passes, not to the bytes of the shadow (profiles/r03_h16r_notes.txt).
//
// This kernel turns the roles around.  The register file of a CU is 512 KiB, three times its LDS:
//
//   * QUERIES IN REGISTERS.  A wavefront keeps the fp16 image of 32 queries -- the A operand of every reduction step of
//     its share of the row -- in 192 VGPRs (12 chunks of 64 elements x 4 steps x 16 bytes per lane).  The 8 wavefronts of
//     the workgroup (2 per SIMD, 256 VGPRs each) form NQB = 8 / KS query blocks x KS parts of the reduction dimension:
//     KS = 1 up to d = 768 (256 queries per pass over a list), 2 up to d = 1536 (128), 4 up to d = 3072 (64).  The parts
//     of a query block add their accumulators through LDS once per 32-row block (4 KiB per wavefront per 32 rows).
//   * ROWS THROUGH LDS, ONCE PER CU.  The shadow of the list arrives by LDS-DMA (global_load_lds_dwordx4: no staging
//     registers, a 1 KiB step lands exactly as it is stored = in B-operand order, so the consumers' ds_read_b128 are
//     lane-linear and conflict free) into a ring of 96 KiB; every wavefront issues 1/8 of each stage and reads all of it.
//     One workgroup barrier per phase (2 chunks = 8 MFMAs per wavefront); the DMAs are counted by hand (asm s_waitcnt
//     vmcnt(N), never 0 in steady state), D = ring - 1 stages in flight (88 / 80 / 64 KiB per CU).
//   * NO VECTOR-MEMORY INSTRUCTION IN THE STEADY STATE except the DMAs: the row norms come through the ring as well, the
//     survivors of the cut are staged per wavefront in LDS and appended (returning atomics) when the stage is full or the
//     work item ends -- a (query, list) pair belongs to ONE wavefront, so nothing is shared.
//
// Work item = (list, tile of <= 32 NQB probing queries), the plan of h16_scan_kernel with T = 32 NQB; same cut, same
// candidate records, same certificate downstream (the approximate value of a row is computed by the same MFMA chain in
// the same order when KS = 1; for KS > 1 the partial sums are added in part order -- the error model bounds any order).
#pragma once

#include <type_traits>

#include "h16_scan_kernels.hpp"

namespace msvs
{

constexpr int HR_CPP = 12;                  // chunks (64 elements) of a query block a wavefront multiplies: its share of the row
constexpr int HR_CR = 11;                   // ... of which in registers (176 VGPRs per block); the 12th is an A operand read from LDS
constexpr uint32_t HR_RING = 80 * 1024;     // bytes of the row ring
constexpr uint32_t HR_TAIL = 32 * 1024;     // 12th chunks: [8 query blocks x parts][32 queries][128 B], XOR-swizzled
constexpr uint32_t HR_SCAP = 384;           // survivor records staged per query block
constexpr uint32_t HR_XN = 16;              // row-norm slots (32-row blocks whose norms may be in flight or in use)

/// LDS bytes of h16r_scan_kernel<*, KS, *>: ring, row norms, tile tables, partial sums of the parts, survivor stages.
inline size_t h16r_lds_bytes(uint32_t ks)
{
    const size_t nqb = 8 / ks, tq = 32 * nqb;
    return HR_RING + HR_TAIL + HR_XN * 256 + 5 * tq * 4 + (ks > 1 ? (8 - nqb) * 4096 : 0) + nqb * 3 * HR_SCAP * 4 + 16;
}

/// One LDS-DMA of 16 bytes per lane: lane i's 16 bytes land at lds_dst + 16 i (lds_dst wave-uniform, in an SGPR).
__device__ __forceinline__ void hr_glds16(const void * gsrc, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

/// 4 bytes per lane (256 bytes per wavefront).
__device__ __forceinline__ void hr_glds4(const void * gsrc, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

template <int N>
__device__ __forceinline__ void hr_wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void hr_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

/// Byte offset of an LDS pointer inside the workgroup's allocation (what M0 / ds instructions address).
__device__ __forceinline__ uint32_t hr_lds_addr(const void * p)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}

/// KS parts of the reduction dimension, QPW query blocks (of 32) per wavefront, 8 / QPW wavefronts:
/// QPW = 1: 8 wavefronts (2 per SIMD, 256 registers each); QPW = 2: 4 wavefronts (1 per SIMD, 512 registers each: the
/// query images spill over into the accumulation registers, every B fragment feeds two MFMAs).
template <int METRIC, int KS, int QPW>
__global__ __launch_bounds__(512 / QPW) void h16r_scan_kernel(const H16Params a)
{
    constexpr uint32_t NWV = 8 / QPW;          // wavefronts
    constexpr uint32_t NQB = 8 / KS, TQ = 32 * NQB;
    constexpr uint32_t NEW = NWV / KS;         // wavefronts with an epilogue (part 0 of their query blocks)
    constexpr uint32_t SB = KS * 8192;         // bytes of a stage: 2 chunks per part of the reduction dimension
    constexpr uint32_t NS = HR_RING / SB;      // ring slots: 10 / 5 / 2
    constexpr uint32_t D = NS - 1;             // stages in flight ahead of the one being multiplied
    constexpr uint32_t PPW = KS * QPW;         // 1 KiB pieces of a stage a wavefront issues
    constexpr int WAITN = (int)((D - 1) * PPW);
    constexpr uint32_t SCAP = HR_SCAP * QPW;   // survivor records an epilogue wavefront stages

    unsigned char * const ring = msvs_smem;
    unsigned char * const tail_s = ring + HR_RING;                            // [NWV][QPW][32][128]
    float * const xn_s = reinterpret_cast<float *>(tail_s + HR_TAIL);         // [HR_XN][64]
    float * const m2_s = xn_s + HR_XN * 64; // five tables of TQ words, consecutive: the epilogue addresses them off one base
    float * const qn_s = m2_s + TQ;
    float * const cutf_s = qn_s + TQ;
    uint32_t * const thr_s = reinterpret_cast<uint32_t *>(cutf_s + TQ);
    uint32_t * const qrow_s = thr_s + TQ;
    float4 * const red_s = reinterpret_cast<float4 *>(qrow_s + TQ); // [8 - NQB][4][64] (KS > 1)
    uint32_t * const stage_s = reinterpret_cast<uint32_t *>(red_s + (KS > 1 ? (8 - NQB) * 256 : 0)); // [NEW][3][SCAP]
    uint32_t * const item_s = stage_s + NEW * 3 * SCAP;

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t qg = wave / KS, kp = wave % KS; // query blocks QPW qg .. + QPW - 1, part kp
    const uint32_t r32 = lane & 31, h = lane >> 5;
    const uint32_t nch = a.nch;
    const uint32_t cpp = (nch + KS - 1) / KS;      // chunks per part (<= HR_CPP, checked by the host)
    const uint32_t npp = (cpp + 1) / 2;            // phases per 32-row block
    const uint32_t c_first = kp * cpp;             // this wavefront's chunks: [c_first, c_first + cpp_w)
    const uint32_t cpp_w = c_first >= nch ? 0u : (nch - c_first < cpp ? nch - c_first : cpp);
    const uint32_t ring_a = hr_lds_addr(ring), xn_a = hr_lds_addr(xn_s);
    const uint32_t total = a.work_off[a.nlist];
    const u32x4 * const hbase = reinterpret_cast<const u32x4 *>(a.H) + lane;
    uint32_t * const stage = stage_s + qg * 3 * SCAP;

    for (;;)
    {
        __syncthreads(); // the previous work item is done with the tables, the ring and item_s
        if (tid == 0)
            *item_s = h16_next_item(a.sched, total, blockIdx.x & 7);
        __syncthreads();
        const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)*item_s);
        if (w == H_NONE)
            break;
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
        const uint32_t lbeg32 = (uint32_t)a.list_off[l], lend32 = (uint32_t)a.list_off[l + 1];
        const uint32_t tidx = w - a.work_off[l];
        const uint32_t pe = a.pair_off[l + 1];
        const uint32_t pb = a.pair_off[l] + tidx * TQ;
        const uint32_t nvalid = pe - pb < TQ ? pe - pb : TQ;
        const uint32_t hb_list = a.hoff[l];
        const uint32_t nblk = a.hoff[l + 1] - hb_list;
        const uint32_t P = nblk > 1 ? (nblk - 1) * npp : 0; // phases of the item: blocks 1 .. nblk - 1 (block 0 = the sample)
        const bool active = 32 * QPW * qg < nvalid && cpp_w > 0;

        // ---- issue of stage s = (block pf_blk, chunk pair pf_pi) into ring slot pf_slot: this wavefront's PPW pieces
        uint32_t pf_blk = 1, pf_pi = 0, pf_slot = 0, pf_s = 0;
        auto issue = [&]() {
            const size_t blk_base = (size_t)(hb_list + pf_blk) * nch * 256; // in 16-byte pieces
#pragma unroll
            for (uint32_t i = 0; i < PPW; i++)
            {
                const uint32_t pc = wave + NWV * i;                      // piece of the stage: [part][chunk of the pair][step]
                const uint32_t part = pc >> 3, cc = (pc >> 2) & 1, j = pc & 3;
                const uint32_t cl = 2 * pf_pi + cc, c = part * cpp + cl;
                const bool valid = cl < cpp && c < nch;
                const u32x4 * src = hbase + blk_base + (size_t)(valid ? c : 0u) * 256 + j * 64;
                hr_glds16(src, ring_a + pf_slot * SB + pc * 1024);
            }
            if (METRIC == M_L2 && pf_pi == 0 && kp == 0 && !(a.dbg & 4))
            {
                uint32_t row = lbeg32 + pf_blk * H_ROWS + lane;
                row = row < lend32 ? row : lend32 - 1;
                hr_glds4(a.xnorm + row, xn_a + (pf_blk % HR_XN) * 256);
            }
            pf_s++;
            pf_slot = pf_slot + 1 == NS ? 0 : pf_slot + 1;
            if (++pf_pi == npp)
            {
                pf_pi = 0;
                pf_blk++;
            }
        };
        const uint32_t dd = (a.dbg >> 8) & 15;                  // experiment: stages in flight (2, 4 or 6 instead of D)
        const uint32_t Dr = dd ? dd : D;
        const bool nodma = (a.dbg & 128) != 0;                  // experiment: no DMA at all
        const uint32_t npro = nodma ? 0u : (P < Dr ? P : Dr);
        for (uint32_t s = 0; s < npro; s++)
            issue();

        // ---- tables of the tile + this wavefront's query images
        for (uint32_t t = tid; t < TQ; t += 64 * NWV)
        {
            const bool v = t < nvalid;
            const uint32_t qp = a.pairs[v ? pb + t : pe - 1];
            const uint32_t q = qp / a.nprobe;
            qrow_s[t] = q;
            const float2 qi = a.qinfo[q];
            m2_s[t] = qi.x;
            qn_s[t] = qi.y;
            const uint32_t cut = v ? a.qthr[q] : 0u; // padding queries of a short tile never pass
            thr_s[t] = cut;
            // the cut as a float for a one-instruction pre-test that never misses a survivor: L2 passes need v <= cutf, inner
            // product passes v >= cutf (words outside the real range: NaN = nothing can pass, +-inf = anything may)
            const uint32_t cw = METRIC == M_L2 ? cut : ~cut;
            float cf = ord2f(cw);
            if (METRIC == M_L2)
                cf = cut <= 0x007FFFFFu ? __uint_as_float(0x7fc00000u) : (cut > 0xFF800000u ? __uint_as_float(0x7f800000u) : cf);
            else
                cf = cw >= 0xFF800000u ? __uint_as_float(0x7fc00000u) : (cw < 0x007FFFFFu ? __uint_as_float(0xff800000u) : cf);
            cutf_s[t] = cf;
        }
        u32x4 aq[QPW][HR_CR][4];
        unsigned char * const tail_w = tail_s + wave * (QPW * 4096);
#pragma unroll
        for (int t = 0; t < QPW; t++)
        {
            // every wavefront loads a full set -- chunks past its share are whatever follows in the image buffer (the host pads
            // it by HR_CPP chunks) and are never multiplied; a block without queries repeats the tile's last pair: one base
            // address per block, immediate offsets, no branches around the loads
            const uint32_t slot = 32 * (QPW * qg + t) + r32;
            const uint32_t qp = a.pairs[slot < nvalid ? pb + slot : pb + nvalid - 1];
            const u32x4 * qsrc = reinterpret_cast<const u32x4 *>(a.Qh) + ((size_t)(qp / a.nprobe) * nch + (cpp_w ? c_first : 0u)) * 8;
#pragma unroll
            for (int c = 0; c < HR_CR; c++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    aq[t][c][j] = qsrc[c * 8 + 2 * j + h];
            // the 12th chunk of the block's queries -> LDS (wave-private: no barrier), piece p of query r at r * 128 + ((p ^ swizzle(r)) << 4)
            if (cpp_w == HR_CPP)
            {
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    const uint32_t piece = 4 * h + i;
                    *reinterpret_cast<u32x4 *>(tail_w + t * 4096 + r32 * 128 + ((piece ^ ((r32 >> 1) & 7)) << 4)) = qsrc[HR_CR * 8 + piece];
                }
            }
        }
        // the images have landed HERE, as far as the compiler is concerned: without a use it waits for them where they are first
        // multiplied -- a counted vmcnt inside the block loop, in the middle of the DMA stream
#pragma unroll
        for (int t = 0; t < QPW; t++)
#pragma unroll
            for (int c = 0; c < HR_CR; c++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    asm volatile("" : "+v"(aq[t][c][j]));
        __syncthreads(); // tables visible
        hr_wait_vm<0>(); // whatever the compiler left in flight (spill stores included): the counted waits below assume only DMAs

        uint32_t cnt = 0; // survivor records staged by this wavefront, wave-uniform
        auto flush = [&]() {
            for (uint32_t i0 = 0; i0 < cnt; i0 += 64)
            {
                const uint32_t i = i0 + lane;
                if (i < cnt)
                {
                    const uint32_t q = stage[2 * SCAP + i];
                    const uint32_t pos = atomicAdd(&a.qcnt[q], 1u);
                    if (pos < a.cand_cap)
                        a.partial[(size_t)q * a.cand_cap + pos] = (uint64_t)stage[SCAP + i] << 32 | stage[i];
                }
            }
            cnt = 0;
            hr_wait_vm<0>(); // the stores too: the counted waits assume nothing but DMAs in flight
        };

        uint32_t s = 0, slot = 0;
        for (uint32_t blk = 1; blk < nblk; blk++)
        {
            f32x16 acc[QPW];
#pragma unroll
            for (int t = 0; t < QPW; t++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    acc[t][r] = 0.f;
#pragma unroll
            for (int pi = 0; pi < HR_CPP / 2; pi++)
            {
                if ((uint32_t)pi < npp)
                {
                    // stage s has landed (this wavefront's pieces: counted wait; everybody's: the barrier), and everybody is done
                    // with stage s - 1, whose slot the next DMA overwrites
                    if (s + Dr > P)
                        hr_wait_vm<0>();
                    else if (dd == 0)
                        hr_wait_vm<WAITN>();
                    else if (dd == 2)
                        hr_wait_vm<1 * PPW>();
                    else if (dd == 4)
                        hr_wait_vm<3 * PPW>();
                    else
                        hr_wait_vm<5 * PPW>();
                    if (a.dbg & 64)
                        ; // experiment: no barrier (races)
                    else if (a.dbg & 8)
                        asm volatile("s_barrier" ::: "memory");
                    else
                        hr_barrier();
                    if (pf_s < P && !nodma)
                        issue();
                    if (active && !(a.dbg & 1))
                    {
                        // the B operand of step t + 1 is requested before the MFMAs of step t are issued (one fragment in flight
                        // beside the one being multiplied; pinned: the scheduler otherwise sinks the read to its use)
                        const unsigned char * sb = ring + slot * SB + kp * 8192 + lane * 16;
                        uint32_t hl = h;
                        asm volatile("" : "+v"(hl)); // opaque: keeps the swizzled offsets of the LDS-resident chunk out of the loop invariants
                        auto run = [&](auto nsteps) {
                            constexpr int NST = decltype(nsteps)::value;
                            u32x4 b = *reinterpret_cast<const u32x4 *>(sb);
#pragma unroll
                            for (int t = 0; t < NST; t++)
                            {
                                u32x4 bn = b;
                                if (t + 1 < NST)
                                    bn = *reinterpret_cast<const u32x4 *>(sb + (t + 1) * 1024);
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int u = 0; u < QPW; u++)
                                {
                                    u32x4 af;
                                    if (2 * pi + (t >> 2) < HR_CR)
                                        af = aq[u][2 * pi + (t >> 2) < HR_CR ? 2 * pi + (t >> 2) : 0][t & 3];
                                    else // piece 2 j + h of the query's 12th chunk (offset recomputed here: four hoisted address registers are four too many)
                                        af = *reinterpret_cast<const u32x4 *>(tail_w + u * 4096 + r32 * 128 + (((2 * (t & 3) + hl) ^ ((r32 >> 1) & 7)) << 4));
                                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, af), __builtin_bit_cast(half8, b),
                                                                                    acc[u], 0, 0, 0);
                                }
                                __builtin_amdgcn_sched_barrier(0);
                                b = bn;
                            }
                        };
                        if (a.dbg & 16) // experiment: MFMAs only (B operand = whatever a register holds)
                        {
#pragma unroll
                            for (int t = 0; t < 8; t++)
                                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, aq[0][2 * pi + (t >> 2) < HR_CR ? 2 * pi + (t >> 2) : 0][t & 3]),
                                                                                __builtin_bit_cast(half8, aq[0][0][0]), acc[0], 0, 0, 0);
                        }
                        else if (a.dbg & 32) // experiment: B reads only
                        {
#pragma unroll
                            for (int t = 0; t < 8; t++)
                            {
                                u32x4 b = *reinterpret_cast<const u32x4 *>(sb + t * 1024);
                                asm volatile("" ::"v"(b));
                            }
                        }
                        else if ((uint32_t)(2 * pi + 1) < cpp_w)
                            run(std::integral_constant<int, 8>{});
                        else if ((uint32_t)(2 * pi) < cpp_w)
                            run(std::integral_constant<int, 4>{});
                    }
                    s++;
                    slot = slot + 1 == NS ? 0 : slot + 1;
                }
            }
            // ---- parts of the reduction dimension -> part 0 of the query blocks
            if (KS > 1)
            {
                if (kp != 0)
                {
#pragma unroll
                    for (int t = 0; t < QPW; t++)
                    {
                        float4 * dst = red_s + (((qg * (KS - 1) + kp - 1) * QPW + t) * 256) + lane;
#pragma unroll
                        for (int i4 = 0; i4 < 4; i4++)
                            dst[i4 * 64] = make_float4(acc[t][4 * i4], acc[t][4 * i4 + 1], acc[t][4 * i4 + 2], acc[t][4 * i4 + 3]);
                    }
                }
                hr_barrier();
                if (kp == 0)
                {
#pragma unroll
                    for (int p = 0; p < KS - 1; p++)
#pragma unroll
                        for (int t = 0; t < QPW; t++)
                        {
                            const float4 * src = red_s + (((qg * (KS - 1) + p) * QPW + t) * 256) + lane;
#pragma unroll
                            for (int i4 = 0; i4 < 4; i4++)
                            {
                                const float4 v = src[i4 * 64];
                                acc[t][4 * i4] += v.x;
                                acc[t][4 * i4 + 1] += v.y;
                                acc[t][4 * i4 + 2] += v.z;
                                acc[t][4 * i4 + 3] += v.w;
                            }
                        }
                }
            }
            // ---- epilogue: accumulator register i of block t = query 32 (QPW qg + t) + (i & 3) + 8 (i >> 2) + 4 h, row r32.
            // Row positions are 32-bit (the host admits the pass only for n < 2^32 rows).  Per register: the approximate value
            // and ONE float compare with the query's cut; only a group of four registers with a possible survivor builds keys.
            if (kp == 0 && 32 * QPW * qg < nvalid && !(a.dbg & 2))
            {
                const uint32_t row = lbeg32 + blk * H_ROWS + r32;
                const uint64_t okmask = __ballot(row < lend32);
                float xn = 0.f;
                if (METRIC == M_L2)
                    xn = xn_s[(blk % HR_XN) * 64 + r32];
                const float4 * const tb = reinterpret_cast<const float4 *>(m2_s) + 8 * QPW * qg + h; // float4 index of query q0 = q0 / 4
#pragma unroll
                for (int t = 0; t < QPW; t++)
#pragma unroll
                    for (int g4 = 0; g4 < 4; g4++)
                    {
                        const float4 m2 = tb[8 * t + 2 * g4];
                        const float4 qn = tb[TQ / 4 + 8 * t + 2 * g4];
                        const float4 cf = tb[TQ / 2 + 8 * t + 2 * g4];
                        const float m2v[4] = {m2.x, m2.y, m2.z, m2.w}, qnv[4] = {qn.x, qn.y, qn.z, qn.w};
                        const float cfv[4] = {cf.x, cf.y, cf.z, cf.w};
                        auto value_of = [&](const int e) {
                            return METRIC == M_L2 ? __fadd_rn(fmaf(m2v[e], acc[t][4 * g4 + e], xn), qnv[e])
                                                  : __fmul_rn(m2v[e], acc[t][4 * g4 + e]);
                        };
                        uint64_t maybe = 0;
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            maybe |= __ballot(METRIC == M_L2 ? value_of(e) <= cfv[e] : value_of(e) >= cfv[e]);
                        if (maybe & okmask)
                        {
                            const uint4 cut = reinterpret_cast<const uint4 *>(tb)[3 * TQ / 4 + 8 * t + 2 * g4];
                            const uint32_t cutv[4] = {cut.x, cut.y, cut.z, cut.w};
                            uint64_t mask[4];
                            uint32_t np = 0;
#pragma unroll
                            for (int e = 0; e < 4; e++)
                            {
                                const uint32_t word = (uint32_t)(make_key<METRIC>(value_of(e), row) >> 32);
                                mask[e] = __ballot(word < cutv[e]) & okmask; // 0xFFFFFFFF (NaN, +-FLT_MAX) is never below a cut
                                np += (uint32_t)__popcll(mask[e]);
                            }
                            if (cnt + np > SCAP) // <= 256 records per group: they fit an empty stage
                                flush();
#pragma unroll
                            for (int e = 0; e < 4; e++)
                            {
                                if ((mask[e] >> lane) & 1)
                                {
                                    const uint32_t at = cnt
                                        + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask[e] >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)mask[e], 0u));
                                    stage[at] = row;
                                    stage[SCAP + at] = (uint32_t)(make_key<METRIC>(value_of(e), row) >> 32);
                                    stage[2 * SCAP + at] = reinterpret_cast<const uint32_t *>(tb)[4 * TQ + 32 * t + 8 * g4 + e];
                                }
                                cnt += (uint32_t)__popcll(mask[e]);
                            }
                        }
                    }
            }
        }
        if (cnt)
            flush();
    }
}

}
