// bin_kernels.hpp -- brute-force scan of BINARY vectors (FixedString(N) columns: N bytes = 8 N bits per row).
//
// Replaces faiss::hammings_knn_mc / jaccard_knn behind tryBruteForceSearch<BinaryVector>
// (src/VectorIndex/Common/BruteForceSearch.h:94-110).  Integer popcounts, so "bit-exact" is literal:
//   Hamming : popcount(x xor y), returned as a float (the reference's distance column is Float32: 4, 8, 12 ...);
//   Jaccard : (|x or y| - |x and y|) / |x or y| as one f32 division of the exact counts, 1 for two all-zero vectors.
// HBM-bound byte work: rows are padded to whole 16-byte words (zero bits change no count) and read
// with 16-byte loads, G lanes per row (G = 1 .. 16, the row's 16-byte words dealt round-robin), so a wavefront
// instruction reads 64 / G rows x 16 G contiguous bytes each; the query sits in LDS; per-row counts are folded over the
// G lanes with DPP-free xor shuffles, and the top-k machinery is the float scan's (WaveTopK, rank merge, merge_kernel)
// on keys (ordered distance word << 32 | row).
#pragma once

#include "scan_kernels.hpp"

namespace msvs
{

enum : int
{
    B_HAMMING = 0,
    B_JACCARD = 1
};

struct BinParams
{
    const uint4 * Y;        // rows, ld16 uint4 per row (zero padded)
    const uint4 * Q;        // queries, same stride
    const uint64_t * alive; // nullable filter bitmap over the rows' labels
    const uint32_t * labels; // nullable: label of row r (else r); results carry labels, ties break by label
    uint32_t nbits;
    uint32_t ld16;
    uint32_t n_rows, rows_per_block, n_blocks, k, nq;
    uint64_t * partial; // [nq][n_blocks][k]
};

/// grid (n_blocks, nq); dynamic LDS: ld16 * 16 + 5 * k * 8 bytes.
template <int METRIC, int G, int R>
__global__ __launch_bounds__(BLOCK) void bin_scan_kernel(const BinParams a)
{
    uint4 * qs = reinterpret_cast<uint4 *>(msvs_smem);
    uint64_t * lds_merge = reinterpret_cast<uint64_t *>(msvs_smem + (size_t)a.ld16 * 16);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = blockIdx.y, k = a.k;
    for (uint32_t c = tid; c < a.ld16; c += BLOCK)
        qs[c] = a.Q[(size_t)q * a.ld16 + c];
    __syncthreads();
    constexpr uint32_t RPW = 64 / G; // rows per wavefront step
    const uint32_t grp = lane / G, g = lane % G;
    const uint32_t row_begin = blockIdx.x * a.rows_per_block;
    uint32_t row_end = row_begin + a.rows_per_block;
    if (row_end > a.n_rows)
        row_end = a.n_rows;
    WaveTopK<R> top;
    top.init();
    for (uint32_t base = row_begin + wave * RPW; base < row_end; base += 4 * RPW)
    {
        const uint32_t r = base + grp;
        const bool rv = r < row_end;
        const uint4 * yrow = a.Y + (size_t)(rv ? r : row_end - 1) * a.ld16;
        uint32_t c0 = 0, c1 = 0; // Hamming: c0 = xor count; Jaccard: c0 = and count, c1 = or count
        for (uint32_t c = g; c < a.ld16; c += G)
        {
            const uint4 y = yrow[c], x = qs[c];
            if (METRIC == B_HAMMING)
                c0 += __popc(x.x ^ y.x) + __popc(x.y ^ y.y) + __popc(x.z ^ y.z) + __popc(x.w ^ y.w);
            else
            {
                c0 += __popc(x.x & y.x) + __popc(x.y & y.y) + __popc(x.z & y.z) + __popc(x.w & y.w);
                c1 += __popc(x.x | y.x) + __popc(x.y | y.y) + __popc(x.z | y.z) + __popc(x.w | y.w);
            }
        }
#pragma unroll
        for (int o = G / 2; o >= 1; o >>= 1)
        {
            c0 += (uint32_t)__shfl_xor((int)c0, o);
            if (METRIC == B_JACCARD)
                c1 += (uint32_t)__shfl_xor((int)c1, o);
        }
        bool ok = rv && g == 0;
        const uint32_t id = ok && a.labels ? a.labels[r] : r;
        if (ok && a.alive)
            ok = id < a.nbits && ((a.alive[id >> 6] >> (id & 63)) & 1);
        const float v = METRIC == B_HAMMING ? (float)c0 : (c1 == 0 ? 1.0f : __fdiv_rn((float)(c1 - c0), (float)c1));
        top.offer(ok ? make_key<M_L2>(v, id) : KEY_NONE, k, lane);
    }
    top.store(lds_merge + wave * k, k, lane);
    __syncthreads();
    uint64_t * merged = lds_merge + 4 * k;
    block_rank_merge(lds_merge, k, merged, k, tid);
    uint64_t * out = a.partial + ((size_t)q * a.n_blocks + blockIdx.x) * k;
    for (uint32_t e = tid; e < k; e += BLOCK)
        out[e] = merged[e];
}

}
