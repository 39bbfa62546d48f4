// filter_kernels.hpp -- the producer and the consumer side of a search filter.
//
// Producer.  The reference evaluates the PREWHERE expression with a CPU pipeline per query and part, pulls the passing
// `_part_offset`s out of it and sets one bit per row (performPrefilter / getFilterFromPipeline,
// src/VectorIndex/Storages/MergeTreeSelectWithHybridSearchProcessor.cpp:905-1112).  Here:
//   filter_from_offsets_kernel   : the same scatter of row offsets into a bitmap, on the device;
//   filter_predicate_kernel<T>   : `column OP constant` (=, !=, <, <=, >, >=, BETWEEN) over a numeric column -> bitmap words
//                                  directly (one wavefront ballot = one 64-bit word), so a simple PREWHERE never leaves the GPU;
//   filter_combine_kernel        : AND / OR / AND NOT of two bitmaps; filter_count_kernel: population count.
//
// Consumer.  A scan that tests the bit of every row still READS every row: with 1 % of the rows passing, 99 % of the
// bytes are wasted.  For selective filters the search runs over a COMPACTED VIEW of the index instead:
//   compact_count_kernel / compact_scan_kernel / compact_fill_kernel : positions of the passing rows in storage (= list-major)
//       order -> rowmap[], and the exclusive rank of every row -> rank[]; compact_offsets_kernel: view offsets of the lists,
//       sel_off[l] = rank[list_off[l]].
// The canonical scan then walks view rows (ScanParams::rowmap) -- the same rows the bit test would have let through, the
// same arithmetic, the same total order: results are identical, only the bytes differ.  Cost of the view: 4 B id + 1 bit
// per stored row per search (0.13 % of a 768-d row).
#pragma once

#include "scan_kernels.hpp"

namespace msvs
{

enum
{
    FOP_EQ = 0,
    FOP_NE = 1,
    FOP_LT = 2,
    FOP_LE = 3,
    FOP_GT = 4,
    FOP_GE = 5,
    FOP_BETWEEN = 6 // lo <= x <= hi
};

static __global__ void filter_from_offsets_kernel(const uint64_t * offsets, size_t n, size_t nbits, unsigned long long * bits)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint64_t r = offsets[i];
    if (r < nbits)
        atomicOr(bits + (r >> 6), 1ull << (r & 63));
}

/// One wavefront per 64 rows: the ballot of the predicate IS the bitmap word.  NaN compares false (!= true), like the host.
template <typename T>
static __global__ void filter_predicate_kernel(const T * col, size_t n, int op, T lo, T hi, uint64_t * bits)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool pass = false;
    if (i < n)
    {
        const T x = col[i];
        switch (op)
        {
            case FOP_EQ: pass = x == lo; break;
            case FOP_NE: pass = x != lo; break;
            case FOP_LT: pass = x < lo; break;
            case FOP_LE: pass = x <= lo; break;
            case FOP_GT: pass = x > lo; break;
            case FOP_GE: pass = x >= lo; break;
            default: pass = x >= lo && x <= hi; break;
        }
    }
    const uint64_t word = __ballot(pass);
    if ((threadIdx.x & 63) == 0 && i < n)
        bits[i >> 6] = word;
}

/// mode 0: a &= b, 1: a |= b, 2: a &= ~b.  Words of b past nb count as zero.
static __global__ void filter_combine_kernel(uint64_t * a, size_t na, const uint64_t * b, size_t nb, int mode)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= na)
        return;
    const uint64_t w = i < nb ? b[i] : 0ull;
    a[i] = mode == 0 ? (a[i] & w) : (mode == 1 ? (a[i] | w) : (a[i] & ~w));
}

static __global__ void filter_count_kernel(const uint64_t * bits, size_t words, size_t nbits, unsigned long long * out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t c = 0;
    if (i < words)
    {
        uint64_t w = bits[i];
        if ((i + 1) * 64 > nbits) // bits past nbits do not count
            w &= nbits > i * 64 ? (~0ull >> (64 - (nbits - i * 64))) : 0ull;
        c = (uint32_t)__popcll(w);
    }
    for (int o = 32; o > 0; o >>= 1)
        c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c)
        atomicAdd(out, (unsigned long long)c);
}

// ------------------------------------------------------------------------------------------ compacted view

constexpr uint32_t COMPACT_CHUNK = 2048; // rows per block of the three passes (256 threads x 8 rows)

struct CompactParams
{
    const uint32_t * ids;   // nullable: id of stored row r (else r)
    const uint64_t * alive; // the effective filter, indexed by id
    uint32_t nbits;
    uint32_t n;             // stored rows
    uint32_t * chunk_cnt;   // [chunks + 1]: passing rows per chunk, then (after the scan) the exclusive prefix; [chunks] = total
    uint32_t * rank;        // [n + 1]: passing rows before stored row r; rank[n] = total
    uint32_t * rowmap;      // [total] stored row of view row v
    uint32_t rowmap_cap;    // its capacity (writes beyond it are dropped: never past the buffer, whatever the caller promised)
    const int64_t * list_off; // [nlist + 1] (FLAT: nullptr)
    uint32_t nlist;
    int64_t * sel_off;      // [nlist + 1] view offsets of the lists
};

__device__ __forceinline__ bool compact_pass(const CompactParams & p, uint32_t r)
{
    if (r >= p.n)
        return false;
    const uint32_t id = p.ids ? p.ids[r] : r;
    return id < p.nbits && ((p.alive[id >> 6] >> (id & 63)) & 1);
}

static __global__ __launch_bounds__(BLOCK) void compact_count_kernel(const CompactParams p)
{
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0)
        s_cnt = 0;
    __syncthreads();
    uint32_t c = 0;
    for (uint32_t j = 0; j < COMPACT_CHUNK / BLOCK; j++)
        c += compact_pass(p, blockIdx.x * COMPACT_CHUNK + j * BLOCK + threadIdx.x) ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1)
        c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0)
        atomicAdd(&s_cnt, c);
    __syncthreads();
    if (threadIdx.x == 0)
        p.chunk_cnt[blockIdx.x] = s_cnt;
}

/// One block: exclusive prefix over the chunk counts (in place), total in chunk_cnt[chunks] and rank[n].
static __global__ __launch_bounds__(1024) void compact_scan_kernel(const CompactParams p, uint32_t chunks)
{
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0)
        s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < chunks; base += 1024)
    {
        const uint32_t i = base + tid;
        const uint32_t v = i < chunks ? p.chunk_cnt[i] : 0u;
        uint32_t incl = v;
        for (int o = 1; o < 64; o <<= 1)
        {
            const uint32_t t = __shfl_up(incl, o);
            if ((int)lane >= o)
                incl += t;
        }
        if (lane == 63)
            s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = s_carry;
        for (uint32_t w = 0; w < wave; w++)
            before += s_wave[w];
        if (i < chunks)
            p.chunk_cnt[i] = before + incl - v;
        __syncthreads();
        if (tid == 1023)
            s_carry = before + incl;
        __syncthreads();
    }
    if (tid == 0)
    {
        p.chunk_cnt[chunks] = s_carry;
        p.rank[p.n] = s_carry;
    }
}

static __global__ __launch_bounds__(BLOCK) void compact_fill_kernel(const CompactParams p)
{
    __shared__ uint32_t s_wave[BLOCK / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t running = p.chunk_cnt[blockIdx.x];
    for (uint32_t j = 0; j < COMPACT_CHUNK / BLOCK; j++)
    {
        const uint32_t r = blockIdx.x * COMPACT_CHUNK + j * BLOCK + tid;
        const bool pass = compact_pass(p, r);
        const uint64_t m = __ballot(pass);
        const uint32_t in_wave = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (lane == 0)
            s_wave[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = running;
        for (uint32_t w = 0; w < wave; w++)
            before += s_wave[w];
        uint32_t all = 0;
        for (uint32_t w = 0; w < BLOCK / 64; w++)
            all += s_wave[w];
        if (r < p.n)
            p.rank[r] = before + in_wave;
        if (pass && before + in_wave < p.rowmap_cap)
            p.rowmap[before + in_wave] = r;
        running += all;
        __syncthreads();
    }
}

static __global__ void compact_offsets_kernel(const CompactParams p)
{
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l > p.nlist)
        return;
    const int64_t at = p.list_off[l];
    p.sel_off[l] = p.rank[at < (int64_t)p.n ? at : p.n];
}

}
