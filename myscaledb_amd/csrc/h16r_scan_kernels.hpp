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
//   * QUERIES IN REGISTERS.  A wavefront keeps the fp16 image of 32 queries -- the A operand of the reduction steps of
//     its part of the row -- in 144 VGPRs (9 chunks of 64 elements x 4 steps x 16 bytes per lane); up to three more
//     chunks sit in a wave-private, XOR-swizzled LDS image and are read as A operands one step ahead.  The 8 wavefronts of
//     the workgroup (2 per SIMD, 256 VGPRs each) are NQB = 8 / KS query blocks x KS parts of the reduction dimension:
//     KS = 1 up to d = 768 (256 queries per pass over a list), 2 up to d = 1536 (128).  The parts of a query block add
//     their accumulators through LDS once per 32-row block (4 KiB per wavefront per 32 rows).
//   * ROWS THROUGH LDS, ONCE PER CU.  A stage = two chunks of every part of a 32-row block (8 / 16 KiB) = 1 / 2 KiB per
//     wavefront, loaded with plain global_load_dwordx4 into a ring of DR register sets (DR stages = 48 KiB per CU in flight,
//     the waits counted by the compiler), written to one of TWO LDS slots a phase before it is multiplied, read by every
//     wavefront as B operands (lane-linear ds_read_b128, conflict free, three fragments in flight).  One barrier per phase.
//     (LDS-DMA -- global_load_lds, no staging registers, a deep LDS ring -- was the first version: its landing writes
//     occupy the LDS for ~100 cycles per KiB and every consumer read queues behind them; DMA stream and B-operand reads
//     ADD UP instead of overlapping: profiles/r03_h16r_notes.txt.)
//   * A (query, list) pair belongs to ONE wavefront: the survivors of the cut are staged in wave-private LDS and appended
//     (returning atomics) when the stage is full or the work item ends.  Per accumulator register the epilogue costs the
//     approximate value and one float compare; keys are built only for registers with a possible survivor.
//
// Work item = (list, tile of <= 32 NQB probing queries), the plan of h16_scan_kernel with T = 32 NQB; same cut, same
// candidate records, same certificate downstream (the approximate value of a row is computed by the same MFMA chain in
// the same order when KS = 1; for KS > 1 the partial sums are added in part order -- the error model bounds any order).
#pragma once

#include <type_traits>

#include "h16_scan_kernels.hpp"

namespace msvs
{

constexpr int HR_CPP = 12;                  // chunks (64 elements) of a query block a wavefront multiplies at most: its part of the row
constexpr int HR_CR = 9;                    // ... of which in registers (144 VGPRs); the others are A operands read from LDS
constexpr uint32_t HR_SCAP = 192;           // survivor records staged per epilogue wavefront

/// LDS bytes of h16r_scan_kernel<*, KS, CPPT>: two row slots, LDS-resident chunks, tile tables, partial sums of the parts, survivor stages.
inline size_t h16r_lds_bytes(uint32_t ks, uint32_t cpp)
{
    const size_t nqb = 8 / ks, tq = 32 * nqb, ct = cpp > HR_CR ? cpp - HR_CR : 0;
    return 2 * ks * 8192 + ct * 8 * 4096 + 5 * tq * 4 + (ks > 1 ? (8 - nqb) * 4096 : 0) + nqb * 3 * HR_SCAP * 4 + 16;
}

/// Chunks per part the kernel is instantiated for (the reduction dimension must be exactly KS x one of these chunks long).
inline bool h16r_cpp_supported(uint32_t cpp) { return cpp == 6 || cpp == 8 || cpp == 12; }

/// The launch (h16r.hip): persistent workgroups, one per CU.  metric M_L2 / M_IP, ks in {1, 2}, a.nch = ks * (supported chunks per part).
void h16r_dispatch(int metric, uint32_t ks, const H16Params & a, uint32_t grid, hipStream_t stream);

/// A 16-byte-per-lane load the compiler does not count: its completion is ours to wait for (hr_wait_vm naming the register),
/// so the load ring keeps its stages in flight across the block loop's back edge (hipcc drains its own loop-carried loads
/// with vmcnt(0) once per block).
__device__ __forceinline__ void hr_load16(u32x4 & dst, const u32x4 * src)
{
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
}

__device__ __forceinline__ void hr_load4(float & dst, const float * src)
{
    asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
}

/// At most N vector-memory operations of this wavefront still outstanding.  No register operand on purpose: a tied operand
/// lets the compiler copy the register BEFORE the wait (seen: v_mov of a ring register ahead of its s_waitcnt); the consumers
/// of a waited-for register are LDS stores / the epilogue's arithmetic behind a scheduling fence, and the loads are issued
/// and waited for in straight-line code with no branch around them, so no phi ever moves a register with a load in flight.
template <int N>
__device__ __forceinline__ void hr_wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void hr_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

/// KS parts of the reduction dimension of CPPT chunks each (nch = KS CPPT exactly), 8 wavefronts = (8 / KS) query blocks x KS parts.
/// Everything that recurs per phase is compile-time or a running register (the first version spent ~1000 cycles per 8 KiB
/// on its control path).
template <int METRIC, int KS, int CPPT>
__global__ __launch_bounds__(512) void h16r_scan_kernel(const H16Params a)
{
    constexpr uint32_t NQB = 8 / KS, TQ = 32 * NQB;
    constexpr int CR = CPPT < HR_CR ? CPPT : HR_CR; // chunks of the queries in registers
    constexpr int CT = CPPT - CR;                   // ... in LDS
    constexpr int NPP = CPPT / 2;                   // phases per 32-row block: a phase = 2 chunks per part
    constexpr int DR = KS == 1 ? NPP : NPP / 2;     // register sets of the load ring = stages in flight
    constexpr uint32_t SB = KS * 8192;              // bytes of a stage
    static_assert(CPPT % 2 == 0 && CPPT <= HR_CPP && NPP % DR == 0 && DR * KS <= 6 && (KS == 1 || KS == 2), "unsupported shape");

    unsigned char * const ring = msvs_smem;                             // [2][SB]
    unsigned char * const tail_s = ring + 2 * SB;                       // [8][CT][32][128]
    float * const m2_s = reinterpret_cast<float *>(tail_s + CT * 8 * 4096); // five tables of TQ words, consecutive
    float * const qn_s = m2_s + TQ;
    float * const cutf_s = qn_s + TQ;
    uint32_t * const thr_s = reinterpret_cast<uint32_t *>(cutf_s + TQ);
    uint32_t * const qrow_s = thr_s + TQ;
    float4 * const red_s = reinterpret_cast<float4 *>(qrow_s + TQ); // [8 - NQB][4][64] (KS > 1)
    uint32_t * const stage_s = reinterpret_cast<uint32_t *>(red_s + (KS > 1 ? (8 - NQB) * 256 : 0)); // [NQB][3][HR_SCAP]
    uint32_t * const item_s = stage_s + NQB * 3 * HR_SCAP;

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t qb = wave % NQB, kp = wave / NQB; // query block, part
    const uint32_t r32 = lane & 31, h = lane >> 5;
    const uint32_t nch = a.nch;                      // = KS * CPPT (checked by the host)
    const uint32_t total = a.work_off[a.nlist];
    uint32_t * const stage = stage_s + qb * 3 * HR_SCAP;
    unsigned char * const tail_w = tail_s + wave * (CT * 4096);
    // this wavefront's piece of every stage: chunk (wave >> 2) of the pair, step (wave & 3), of each part
    const u32x4 * const hbase = reinterpret_cast<const u32x4 *>(a.H) + (wave >> 2) * 256 + (wave & 3) * 64 + lane;
    unsigned char * const wdst = ring + wave * 1024 + lane * 16; // where it lands in a slot (+ 8192 per part)

    for (;;)
    {
        __syncthreads(); // the previous work item is done with the tables, the slots and item_s
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
        const bool active = 32 * qb < nvalid;

        // ---- the row stream: stage after stage of blocks 1 .. nblk - 1; src = this wavefront's piece of the next stage to load
        const u32x4 * src = hbase + (size_t)(hb_list + 1) * nch * 256;
        u32x4 st[DR][KS];
        // (unconditional: past the end of the list it reads the next list's blocks -- the shadow is padded by the host -- into
        // registers nobody uses; a branch around an uncounted load would let the compiler move its register)
        auto load = [&](u32x4 (&dst)[KS], const bool last_of_block) {
#pragma unroll
            for (int i = 0; i < KS; i++)
                hr_load16(dst[i], src + i * (CPPT * 256));
            src += 512; // the next pair of chunks
            if (last_of_block)
                src += (KS - 1) * CPPT * 256; // ... of the next block: skip the other parts' chunks
        };
#pragma unroll
        for (int k = 0; k < DR; k++)
        {
            load(st[k], k % NPP == NPP - 1);
        }
        __builtin_amdgcn_sched_barrier(0);

        // ---- tables of the tile + this wavefront's query images
        for (uint32_t t = tid; t < TQ; t += 512)
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
        u32x4 aq[CR][4];
        {
            // a block without queries repeats the tile's last pair (its wavefront multiplies nothing): no branches around the loads
            const uint32_t slot = 32 * qb + r32;
            const uint32_t qp = a.pairs[slot < nvalid ? pb + slot : pb + nvalid - 1];
            const u32x4 * qsrc = reinterpret_cast<const u32x4 *>(a.Qh) + ((size_t)(qp / a.nprobe) * nch + kp * CPPT) * 8;
#pragma unroll
            for (int c = 0; c < CR; c++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    aq[c][j] = qsrc[c * 8 + 2 * j + h];
            // the chunks beyond the registers -> LDS (wave-private: no barrier), piece p of query r at r * 128 + ((p ^ swizzle(r)) << 4)
#pragma unroll
            for (int c = 0; c < CT; c++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    const uint32_t piece = 4 * h + i;
                    *reinterpret_cast<u32x4 *>(tail_w + c * 4096 + r32 * 128 + ((piece ^ ((r32 >> 1) & 7)) << 4)) = qsrc[(CR + c) * 8 + piece];
                }
        }
        // the images have landed HERE, as far as the compiler is concerned: without a use it waits for them where they are first
        // multiplied -- inside the block loop, in the middle of the row stream
#pragma unroll
        for (int c = 0; c < CR; c++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                asm volatile("" : "+v"(aq[c][j]));
        // row norms: requested at the head of a block, used in its epilogue (six younger stage loads later: the wait is counted)
        auto load_xn = [&](float & xn, const uint32_t blk) {
            uint32_t row = lbeg32 + blk * H_ROWS + r32;
            row = row < lend32 ? row : lend32 - 1;
            hr_load4(xn, a.xnorm + row);
        };
        // stage 0 -> slot 0
        hr_wait_vm<0>();
#pragma unroll
        for (int i = 0; i < KS; i++)
            *reinterpret_cast<u32x4 *>(wdst + i * 8192) = st[0][i];
        load(st[0], DR % NPP == NPP - 1);
        __syncthreads(); // tables visible

        uint32_t cnt = 0; // survivor records staged by this wavefront, wave-uniform
        auto flush = [&]() {
#pragma unroll 1
            for (uint32_t i0 = 0; i0 < cnt; i0 += 64)
            {
                const uint32_t i = i0 + lane;
                if (i < cnt)
                {
                    const uint32_t q = stage[2 * HR_SCAP + i];
                    const uint32_t pos = atomicAdd(&a.qcnt[q], 1u);
                    if (pos < a.cand_cap)
                        a.partial[(size_t)q * a.cand_cap + pos] = (uint64_t)stage[HR_SCAP + i] << 32 | stage[i];
                }
            }
            cnt = 0;
        };

        uint32_t c_off = 0; // LDS offset of the slot being multiplied
        const uint32_t sbv0 = lane * 16 + kp * 8192;
#pragma unroll 1
        for (uint32_t blk = 1; blk < nblk; blk++)
        {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++)
                acc[r] = 0.f;
            float xn = 0.f;
            if (METRIC == M_L2)
                load_xn(xn, blk); // every wavefront, part 0 or not: no branch around an uncounted load
#pragma unroll
            for (int pi = 0; pi < NPP; pi++)
            {
                // everybody is done multiplying stage s - 1 (its slot is free) and stage s is complete in its slot
                hr_barrier();
                // stage s + 1: registers -> the other slot; its register set then takes the load of stage s + 1 + DR
                // (its load is DR stages old: the DR - 1 younger stages stay in flight)
                hr_wait_vm<(DR - 1) * KS>();
#pragma unroll
                for (int i = 0; i < KS; i++)
                    *reinterpret_cast<u32x4 *>(wdst + (c_off ^ SB) + i * 8192) = st[(pi + 1) % DR][i];
                load(st[(pi + 1) % DR], (pi + 1 + DR) % NPP == NPP - 1);
                __builtin_amdgcn_sched_barrier(0);
                if (active && !(a.dbg & 1))
                {
                    // B operands two steps ahead of the MFMA that uses them, LDS-resident A operands one step ahead (LDS latency
                    // ~170 cycles with eight wavefronts reading, an MFMA 32); the issue order pinned
                    const unsigned char * sb = ring + (sbv0 + c_off);
                    uint32_t hl = h;
                    if (CT > 0)
                        asm volatile("" : "+v"(hl)); // opaque: keeps the swizzled offsets of the LDS-resident chunks out of the loop invariants
                    auto a_lds = [&](const int t) { // piece 2 j + h of LDS-resident chunk 2 pi + (t >> 2) - CR
                        return *reinterpret_cast<const u32x4 *>(tail_w + (2 * pi + (t >> 2) - CR) * 4096 + r32 * 128
                                                                + (((2 * (t & 3) + hl) ^ ((r32 >> 1) & 7)) << 4));
                    };
                    u32x4 b[3], al[2];
                    b[0] = *reinterpret_cast<const u32x4 *>(sb);
                    b[1] = *reinterpret_cast<const u32x4 *>(sb + 1024);
                    if (2 * pi >= CR)
                        al[0] = a_lds(0);
#pragma unroll
                    for (int t = 0; t < 8; t++)
                    {
                        if (t + 2 < 8)
                            b[(t + 2) % 3] = *reinterpret_cast<const u32x4 *>(sb + (t + 2) * 1024);
                        if (t + 1 < 8 && 2 * pi + ((t + 1) >> 2) >= CR)
                            al[(t + 1) & 1] = a_lds(t + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        const u32x4 af = 2 * pi + (t >> 2) < CR ? aq[2 * pi + (t >> 2) < CR ? 2 * pi + (t >> 2) : 0][t & 3] : al[t & 1];
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, af), __builtin_bit_cast(half8, b[t % 3]), acc, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                c_off ^= SB;
            }
            if (METRIC == M_L2)
                hr_wait_vm<NPP * KS>(); // the norms; the NPP stages requested after them stay in flight
            // ---- parts of the reduction dimension -> part 0 of the query block
            if (KS > 1)
            {
                if (kp != 0)
                {
                    float4 * dst = red_s + ((qb * (KS - 1) + kp - 1) * 256) + lane;
#pragma unroll
                    for (int i4 = 0; i4 < 4; i4++)
                        dst[i4 * 64] = make_float4(acc[4 * i4], acc[4 * i4 + 1], acc[4 * i4 + 2], acc[4 * i4 + 3]);
                }
                hr_barrier();
                if (kp == 0)
                {
#pragma unroll
                    for (int p = 0; p < KS - 1; p++)
                    {
                        const float4 * rsrc = red_s + ((qb * (KS - 1) + p) * 256) + lane;
#pragma unroll
                        for (int i4 = 0; i4 < 4; i4++)
                        {
                            const float4 v = rsrc[i4 * 64];
                            acc[4 * i4] += v.x;
                            acc[4 * i4 + 1] += v.y;
                            acc[4 * i4 + 2] += v.z;
                            acc[4 * i4 + 3] += v.w;
                        }
                    }
                }
            }
            // ---- epilogue: accumulator register i = query 32 qb + (i & 3) + 8 (i >> 2) + 4 h, row r32 of the block.
            // Row positions are 32-bit (the host admits the pass only for n < 2^32 rows).  Per register: the approximate value
            // and ONE float compare with the query's cut; only a register with a possible survivor builds keys.
            if (kp == 0 && active && !(a.dbg & 2))
            {
                const uint32_t row = lbeg32 + blk * H_ROWS + r32;
                const uint64_t okmask = __ballot(row < lend32);
                const float4 * const tb = reinterpret_cast<const float4 *>(m2_s) + 8 * qb + h; // float4 index of query q0 = q0 / 4
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++)
                {
                    const float4 m2 = tb[2 * g4];
                    const float4 qn = tb[TQ / 4 + 2 * g4];
                    const float4 cf = tb[TQ / 2 + 2 * g4];
                    const float m2v[4] = {m2.x, m2.y, m2.z, m2.w}, qnv[4] = {qn.x, qn.y, qn.z, qn.w};
                    const float cfv[4] = {cf.x, cf.y, cf.z, cf.w};
#pragma unroll
                    for (int e = 0; e < 4; e++)
                    {
                        const float v = METRIC == M_L2 ? __fadd_rn(fmaf(m2v[e], acc[4 * g4 + e], xn), qnv[e]) : __fmul_rn(m2v[e], acc[4 * g4 + e]);
                        if (__ballot(METRIC == M_L2 ? v <= cfv[e] : v >= cfv[e]) & okmask)
                        {
                            const uint32_t cut = reinterpret_cast<const uint32_t *>(tb)[3 * TQ + 8 * g4 + e];
                            const uint32_t word = (uint32_t)(make_key<METRIC>(v, row) >> 32);
                            const uint64_t mask = __ballot(word < cut) & okmask; // 0xFFFFFFFF (NaN, +-FLT_MAX) is never below a cut
                            if (mask)
                            {
                                const uint32_t np = (uint32_t)__popcll(mask);
                                if (cnt + np > HR_SCAP)
                                    flush();
                                if ((mask >> lane) & 1)
                                {
                                    const uint32_t at = cnt
                                        + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                                    stage[at] = row;
                                    stage[HR_SCAP + at] = word;
                                    stage[2 * HR_SCAP + at] = reinterpret_cast<const uint32_t *>(tb)[4 * TQ + 8 * g4 + e];
                                }
                                cnt += np;
                            }
                        }
                    }
                }
            }
        }
        if (cnt)
            flush();
        hr_wait_vm<0>(); // the loads past the end of the item: their registers are about to be reused
    }
}

}
