// scan_kernels.hpp -- CDNA4 (gfx950) device code of the candidate scan: batched L2 / IP distances of a
// query tile against row-major f32 rows streamed from HBM, wavefront top-k, block / grid merges.
//
// Numerics contract (must stay bit-identical to oracle/msvs_oracle.c "W64 tree"):
//   a row is owned by a 16-lane DPP row; lane g owns float4 columns g, g+16, g+32, ... and keeps one f32
//   accumulator per float4 component, i.e. accumulator index (k mod 64) for element k; products and sums are
//   separately rounded (__fmul_rn/__fadd_rn, never fma); the 64 partial sums are folded by the adjacent
//   pairwise tree: (c0+c1)+(c2+c3) inside the lane, then lane^1, lane^2, quad^1, half^1 through DPP.
// Top-k contract: 64-bit keys (orderable(distance) << 32 | id) ascending == canonical best-first order with
//   ascending-id tie break; key ~0 is the "no result" sentinel (id -1).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// __fmul_rn/__fadd_rn/__fsub_rn are plain operators in this toolchain: keep the compiler from fusing them.
#pragma clang fp contract(off)

namespace msvs
{

constexpr int WAVE = 64;
constexpr int BLOCK = 256; // 4 waves
constexpr uint64_t KEY_NONE = ~0ull;

enum : int
{
    M_L2 = 0,
    M_IP = 1
};

// ------------------------------------------------------------------------------------------ keys

__device__ __forceinline__ uint32_t f2ord(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float ord2f(uint32_t o)
{
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}

template <int METRIC>
__device__ __forceinline__ uint64_t make_key(float v, uint32_t id)
{
    // admitted only if strictly better than the heap's neutral value (NaN never): Faiss CMax/CMin semantics
    bool ok = METRIC == M_IP ? (v > -3.402823466e+38f) : (v < 3.402823466e+38f);
    uint32_t hi = METRIC == M_IP ? ~f2ord(v) : f2ord(v);
    return ok ? ((uint64_t)hi << 32 | id) : KEY_NONE;
}

template <int METRIC>
__device__ __forceinline__ float key_value(uint64_t key)
{
    if (key == KEY_NONE)
        return METRIC == M_IP ? -3.402823466e+38f : 3.402823466e+38f;
    uint32_t hi = (uint32_t)(key >> 32);
    return ord2f(METRIC == M_IP ? ~hi : hi);
}

// ------------------------------------------------------------------------------------------ cross-lane helpers

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int lane)
{
    uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return (uint64_t)hi << 32 | lo;
}

/// Lane i gets lane i-1's value (lane 0: unspecified).  DPP wave_shr:1 -- a register move, against the ds_bpermute
/// round trip through the LDS crossbar that __shfl_up costs; this sits on the serial path of every top-k insertion.
__device__ __forceinline__ uint64_t shfl_up64(uint64_t v)
{
    uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x138, 0xF, 0xF, false);
    uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x138, 0xF, 0xF, false);
    return (uint64_t)hi << 32 | lo;
}

template <int CTRL>
__device__ __forceinline__ float dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

/// Sum of the 16 lanes of a DPP row in adjacent-pairwise-tree order; every lane of the row gets the result.
__device__ __forceinline__ float row16_tree_sum(float s)
{
    s = __fadd_rn(s, dpp<0xB1>(s));  // quad_perm [1,0,3,2]  : lane ^ 1
    s = __fadd_rn(s, dpp<0x4E>(s));  // quad_perm [2,3,0,1]  : lane ^ 2
    s = __fadd_rn(s, dpp<0x141>(s)); // row_half_mirror      : other quad of the half (quads are uniform by now)
    s = __fadd_rn(s, dpp<0x140>(s)); // row_mirror           : other half of the row
    return s;
}

// ------------------------------------------------------------------------------------------ wavefront top-k

/// Sorted (ascending key) list of up to 64*R entries spread over the lanes of one wavefront:
/// entry e lives in register v[e / 64] of lane e % 64.  `thr` is the current k-th best key (wave-uniform).
template <int R>
struct WaveTopK
{
    uint64_t v[R];
    uint64_t thr;

    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int r = 0; r < R; r++)
            v[r] = KEY_NONE;
        thr = KEY_NONE;
    }

    __device__ __forceinline__ void insert(uint64_t ckey, uint32_t k, uint32_t lane)
    {
        uint32_t pos = 0;
#pragma unroll
        for (int r = 0; r < R; r++)
            pos += __popcll(__ballot(v[r] < ckey));
#pragma unroll
        for (int r = R - 1; r >= 0; r--)
        {
            uint64_t up = shfl_up64(v[r]);
            if (r > 0)
            {
                uint64_t carry = readlane64(v[r - 1], 63);
                if (lane == 0)
                    up = carry;
            }
            uint32_t e = r * 64 + lane;
            v[r] = e < pos ? v[r] : (e == pos ? ckey : up);
        }
        uint32_t kr = (k - 1) >> 6, kl = (k - 1) & 63;
#pragma unroll
        for (int r = 0; r < R; r++)
            if (r == (int)kr)
                thr = readlane64(v[r], kl);
    }

    /// Offer one candidate per lane (KEY_NONE = nothing).
    __device__ __forceinline__ void offer(uint64_t key, uint32_t k, uint32_t lane)
    {
        uint64_t m = __ballot(key < thr);
        while (m)
        {
            int b = __builtin_ctzll(m);
            m &= m - 1;
            uint64_t ck = readlane64(key, b);
            if (ck < thr)
                insert(ck, k, lane);
        }
    }

    __device__ __forceinline__ void store(uint64_t * dst, uint32_t k, uint32_t lane) const
    {
#pragma unroll
        for (int r = 0; r < R; r++)
        {
            uint32_t e = r * 64 + lane;
            if (e < k)
                dst[e] = v[r];
        }
    }
};

/// Merge the 4 per-wave sorted lists lists[w][0..k) (LDS) into out[0..k) (LDS) by rank:
/// rank = own index + #keys of the other lists that sort before it (equal keys: lower list first).
/// Must be called by all BLOCK threads; ends with a barrier.
__device__ __forceinline__ void block_rank_merge(const uint64_t * lists, uint32_t stride, uint64_t * out, uint32_t k,
                                                 uint32_t tid)
{
    for (uint32_t i = tid; i < k; i += BLOCK)
        out[i] = KEY_NONE;
    __syncthreads();
    for (uint32_t i = tid; i < 4 * k; i += BLOCK)
    {
        uint32_t w = i / k, e = i - w * k;
        uint64_t key = lists[w * stride + e];
        if (key == KEY_NONE)
            continue;
        uint32_t pos = e;
        for (uint32_t o = 0; o < 4; o++)
        {
            if (o == w)
                continue;
            // keys of list o that sort before this one: the smaller ones, and -- for equal keys (the same id at the same
            // distance in two lists: overlapping parts, a row added twice) -- those of the lower-numbered list, so that
            // every copy gets its own rank instead of two copies landing in one slot and leaving a hole
            const uint64_t * l = lists + o * stride;
            const bool ties_first = o < w;
            uint32_t lo = 0, hi = k; // first index with l[idx] >= key (> key when ties_first)
            while (lo < hi)
            {
                uint32_t mid = (lo + hi) >> 1;
                if (l[mid] < key || (ties_first && l[mid] == key))
                    lo = mid + 1;
                else
                    hi = mid;
            }
            pos += lo;
        }
        if (pos < k)
            out[pos] = key;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------ the scan

struct ScanParams
{
    const float4 * Y;       // base rows, ld4 float4 per row (zero padded to a multiple of 4 floats)
    const uint32_t * ids;   // nullable: id of stored row r (else r + id_base)
    const uint32_t * rowmap; // nullable: the scan runs over a COMPACTED view -- view row r is stored row rowmap[r]
                             // (the rows a selective filter lets through, filter_kernels.hpp); list offsets, row
                             // ranges and segments are in view rows, Y / ids are indexed by the stored row
    const uint64_t * alive; // nullable filter bitmap indexed by id, LSB-first
    const float4 * Q;       // queries, ld4 float4 per row
    uint64_t * partial;     // output keys
    uint32_t id_base;
    uint32_t nbits;
    uint32_t ld4;
    uint32_t k;
    uint32_t nq;
    // FLAT front end
    uint32_t n_rows;
    const uint32_t * n_rows_dev; // nullable: the row count lives on the device (a compacted view: n_rows is its upper bound)
    uint32_t rows_per_block;
    uint32_t n_blocks; // partial lists per query
    // IVF front end
    const int32_t * probes;   // [nq][nprobe] list ids (-1 = none)
    const int64_t * list_off; // [nlist+1]
    const int64_t * list_end; // candidate pass only, nullable: rows of list l end here instead of at list_off[l+1]
    uint32_t nprobe;
    uint32_t seg_max;
    // list-batched IVF front end (see IvfPlanParams)
    const uint32_t * pairs;
    const uint32_t * pair_off;
    const uint32_t * work_off;
    uint32_t nlist;
    uint32_t xcd_order; // 1: XCD-contiguous work ranges (see ivf_batched_scan_kernel)
    // matrix-core candidate pass and its fallback (mfma_scan_kernels.hpp)
    const float * qnorm;     // [nq] |q|^2
    const float * xnorm;     // [n_rows] |x|^2
    const uint32_t * qmap;   // subset kernels: the queries to process ...
    const uint32_t * qcount; // ... and how many of them (device side)
    uint32_t slot_base, slot_cap; // subset kernels: this round's window [slot_base, slot_base + slot_cap) of the fail list;
                                  // the partial lists of entry f sit in slot f - slot_base
    uint32_t * qthr;         // big-tile candidate pass: per-query running cut (ordered distance word), see there
    uint32_t * qcnt;         // big-tile candidate pass: keys emitted so far per query (append cursor into `partial`)
    uint32_t cand_cap;       // ... whose row for query q is partial[q * cand_cap ...]
    uint32_t tile_q;         // candidate pass: queries per tile of the plan (0: the kernel's full tile); a multiple of 32
    const float4 * Qsplit;   // candidate pass: the queries in split-bf16 step layout (split_queries_kernel)
};

/// Scans rows [row_begin,row_end) for T queries already staged in LDS (qs[t*ld4 + c]) and leaves the block's
/// merged top-k of query t in out[t][0..k) (global).  All BLOCK threads participate.
template <int METRIC, int T, int R>
__device__ __forceinline__ void scan_rows(const ScanParams & a, uint32_t row_begin, uint32_t row_end,
                                          const float4 * qs, uint64_t * lds_merge, uint64_t * const * out)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4, g = lane & 15;
    const uint32_t ld4 = a.ld4, k = a.k;
    const uint32_t jfull = ld4 >> 4, jtail = ld4 & 15;

    WaveTopK<R> top[T];
#pragma unroll
    for (int t = 0; t < T; t++)
        top[t].init();

    // one (query, float4 column) update; the accumulator of component c of query t only ever sees its own
    // columns in ascending order, whatever the loop nest around it looks like
    auto fma4 = [&](float4 & s, const float4 q, const float4 y) {
        if (METRIC == M_L2)
        {
            float dx = __fsub_rn(q.x, y.x), dy = __fsub_rn(q.y, y.y), dz = __fsub_rn(q.z, y.z),
                  dw = __fsub_rn(q.w, y.w);
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
    };


    // id + filter of a finished row (same value in the 16 lanes of the row; only lane g == 0 offers it), then the
    // cross-lane sum in tree order and the offer to each query's top-k
    auto finish_row = [&](uint32_t r, bool rv, const float4 * acc) {
        uint32_t id = 0;
        bool ok = rv && g == 0;
        if (ok)
        {
            const uint32_t rs = a.rowmap ? a.rowmap[r] : r;
            id = a.ids ? a.ids[rs] : rs + a.id_base;
            if (a.alive)
                ok = id < a.nbits && ((a.alive[id >> 6] >> (id & 63)) & 1);
        }
#pragma unroll
        for (int t = 0; t < T; t++)
        {
            float s = __fadd_rn(__fadd_rn(acc[t].x, acc[t].y), __fadd_rn(acc[t].z, acc[t].w));
            s = row16_tree_sum(s);
            uint64_t key = ok ? make_key<METRIC>(s, id) : KEY_NONE;
            top[t].offer(key, k, lane);
        }
    };

    // One query, rows of exactly 12 float4 per lane (d = 768): the rows of the NEXT step are in flight while this step's
    // are consumed (two named register sets, swapped by unrolling: a copy of a register waits for its load).  With the
    // plain loop below every 16-row step pays a full memory round trip; a few-query search is a handful of steps per
    // block and nothing else hides them.  Same arithmetic, same order.
    if (T == 1 && jfull == 12 && jtail == 0)
    {
        const float4 * qrow = qs + g;
        auto load_rows = [&](uint32_t base, float4 * y) {
            const uint32_t r = base + grp, rc = r < row_end ? r : row_end - 1;
            const float4 * yrow = a.Y + (size_t)(a.rowmap ? a.rowmap[rc] : rc) * ld4 + g;
#pragma unroll
            for (int u = 0; u < 12; u++)
                y[u] = yrow[u * 16];
        };
        auto consume = [&](uint32_t base, const float4 * y) {
            float4 acc[T];
            acc[0] = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 q[12];
#pragma unroll
            for (int u = 0; u < 12; u++)
                q[u] = qrow[u * 16];
#pragma unroll
            for (int u = 0; u < 12; u++)
                fma4(acc[0], q[u], y[u]);
            finish_row(base + grp, base + grp < row_end, acc);
        };
        float4 ya[12], yb[12];
        uint32_t base = row_begin + wave * 4;
        if (base < row_end)
            load_rows(base, ya);
        while (base < row_end)
        {
            if (base + 16 < row_end)
                load_rows(base + 16, yb);
            __builtin_amdgcn_sched_barrier(0);
            consume(base, ya);
            base += 16;
            if (base >= row_end)
                break;
            if (base + 16 < row_end)
                load_rows(base + 16, ya);
            __builtin_amdgcn_sched_barrier(0);
            consume(base, yb);
            base += 16;
        }
    }
    else
    for (uint32_t base = row_begin + wave * 4; base < row_end; base += 16)
    {
        const uint32_t r = base + grp;
        const bool rv = r < row_end;
        const uint32_t rc = rv ? r : row_end - 1;
        const float4 * yrow = a.Y + (size_t)(a.rowmap ? a.rowmap[rc] : rc) * ld4 + g;
        const float4 * qrow = qs + g;

        float4 acc[T];
#pragma unroll
        for (int t = 0; t < T; t++)
            acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);

        // The row is consumed in chunks of JC float4 columns per lane: all JC 16-byte loads are issued first
        // (JC KiB in flight per wave), then the queries are walked OUTSIDE the columns, so only one query's
        // LDS operands are live at a time (keeps T = 8 near 128 VGPRs instead of 245).
#ifndef MSVS_JC8
#define MSVS_JC8 6
#endif
        constexpr int JC = T >= 8 ? MSVS_JC8 : (T == 1 ? 12 : 6); // one query: registers to spare, 12 KiB in flight per wave
        uint32_t j = 0;
        for (; j + JC <= jfull; j += JC)
        {
            float4 y[JC];
#pragma unroll
            for (int u = 0; u < JC; u++)
                y[u] = yrow[(j + u) * 16];
#pragma unroll
            for (int t = 0; t < T; t++)
            {
                const float4 * qp = qrow + t * ld4 + j * 16;
                float4 q[JC];
#pragma unroll
                for (int u = 0; u < JC; u++)
                    q[u] = qp[u * 16];
#pragma unroll
                for (int u = 0; u < JC; u++)
                    fma4(acc[t], q[u], y[u]);
                // do not let the scheduler hoist the next query's LDS reads above this one's arithmetic
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (; j + 2 <= jfull; j += 2)
        {
            const float4 y0 = yrow[j * 16], y1 = yrow[(j + 1) * 16];
#pragma unroll
            for (int t = 0; t < T; t++)
            {
                fma4(acc[t], qrow[t * ld4 + j * 16], y0);
                fma4(acc[t], qrow[t * ld4 + (j + 1) * 16], y1);
            }
        }
        for (; j < jfull; j++)
        {
            const float4 y1 = yrow[j * 16];
#pragma unroll
            for (int t = 0; t < T; t++)
                fma4(acc[t], qrow[t * ld4 + j * 16], y1);
        }
        if (g < jtail)
        {
            const float4 y1 = yrow[jfull * 16];
#pragma unroll
            for (int t = 0; t < T; t++)
                fma4(acc[t], qrow[t * ld4 + jfull * 16], y1);
        }

        finish_row(r, rv, acc);
    }

    // 4 wave lists -> 1 block list per query, all T queries in one pass (3 barriers per work item):
    // lists at lds_merge[(t*4 + wave)*k + e], merged lists at lds_merge[T*4*k + t*k + e]
    uint64_t * merged = lds_merge + (size_t)T * 4 * k;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < T; t++)
        top[t].store(lds_merge + (t * 4 + wave) * k, k, lane);
    for (uint32_t i = tid; i < T * k; i += BLOCK)
        merged[i] = KEY_NONE;
    __syncthreads();
    for (uint32_t i = tid; i < T * 4 * k; i += BLOCK)
    {
        const uint32_t t = i / (4 * k), rem = i - t * 4 * k, w = rem / k, e = rem - w * k;
        const uint64_t * lists = lds_merge + (size_t)t * 4 * k;
        const uint64_t key = lists[w * k + e];
        if (key == KEY_NONE)
            continue;
        uint32_t pos = e; // rank = own index + #smaller keys in the other three lists (keys are unique)
        for (uint32_t o = 0; o < 4; o++)
        {
            if (o == w)
                continue;
            const uint64_t * l = lists + o * k;
            uint32_t lo = 0, hi = k;
            while (lo < hi)
            {
                uint32_t mid = (lo + hi) >> 1;
                if (l[mid] < key)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            pos += lo;
        }
        if (pos < k)
            merged[t * k + pos] = key;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < T; t++)
        for (uint32_t e = tid; e < k; e += BLOCK)
            out[t][e] = merged[t * k + e];
}

/// Stage T query rows (zero padded) into LDS.  Queries beyond nq repeat the last valid one.
template <int T>
__device__ __forceinline__ void stage_queries(const ScanParams & a, const uint32_t * qidx, float4 * qs)
{
#pragma unroll
    for (int t = 0; t < T; t++) // static index into qidx: keeps it in registers
    {
        const float4 * src = a.Q + (size_t)qidx[t] * a.ld4;
        for (uint32_t c = threadIdx.x; c < a.ld4; c += BLOCK)
            qs[t * a.ld4 + c] = src[c];
    }
    __syncthreads();
}

extern __shared__ __attribute__((aligned(16))) unsigned char msvs_smem[];

/// LDS bytes a scan block needs.
inline size_t scan_lds_bytes(uint32_t T, uint32_t ld4, uint32_t k)
{
    return (size_t)T * ld4 * 16 + (size_t)T * 5 * k * 8;
}
constexpr size_t SCAN_LDS_BUDGET = 64 * 1024; // what the planners let a canonical scan block use

/// FLAT: grid (n_blocks, ceil(nq/T)); block bx scans rows [bx*rows_per_block, ...) for queries by*T...
/// partial layout: [nq][n_blocks][k].
template <int METRIC, int T, int R>
__global__ __launch_bounds__(BLOCK) void flat_scan_kernel(const ScanParams a)
{
    float4 * qs = reinterpret_cast<float4 *>(msvs_smem);
    uint64_t * lds_merge = reinterpret_cast<uint64_t *>(msvs_smem + (size_t)T * a.ld4 * 16);
    const uint32_t q0 = blockIdx.y * T;
    uint32_t qidx[T];
    uint64_t * out[T];
    uint64_t * dummy = lds_merge + 5 * 0; // never used: duplicates write to their own (valid) slot of the last query
    (void)dummy;
#pragma unroll
    for (int t = 0; t < T; t++)
    {
        uint32_t q = q0 + t < a.nq ? q0 + t : a.nq - 1;
        qidx[t] = q;
        out[t] = a.partial + ((size_t)q * a.n_blocks + blockIdx.x) * a.k;
    }
    stage_queries<T>(a, qidx, qs);
    const uint32_t row_begin = blockIdx.x * a.rows_per_block;
    uint32_t row_end = row_begin + a.rows_per_block;
    const uint32_t n_rows = a.n_rows_dev ? *a.n_rows_dev : a.n_rows;
    if (row_end > n_rows)
        row_end = n_rows;
    scan_rows<METRIC, T, R>(a, row_begin < row_end ? row_begin : row_end, row_end, qs, lds_merge, out);
}

/// IVF, one query per block (the low-latency form used for very small batches): grid (seg_max, nprobe, nq).
/// Block (s, p, q) scans segment s of the p-th probed list of query q and writes its top-k to
/// partial[((q*nprobe + p)*seg_max + s)*k ...]; blocks past the end of the list exit without writing (the merge
/// kernel derives the number of valid segments from the list length).
template <int METRIC, int R>
__global__ __launch_bounds__(BLOCK) void ivf_scan_kernel(const ScanParams a)
{
    const uint32_t s = blockIdx.x, p = blockIdx.y, q = blockIdx.z;
    uint64_t * out0 = a.partial + (((size_t)q * a.nprobe + p) * a.seg_max + s) * a.k;
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
        return;
    float4 * qs = reinterpret_cast<float4 *>(msvs_smem);
    uint64_t * lds_merge = reinterpret_cast<uint64_t *>(msvs_smem + (size_t)a.ld4 * 16);
    uint32_t qidx[1] = {q};
    uint64_t * out[1] = {out0};
    stage_queries<1>(a, qidx, qs);
    scan_rows<METRIC, 1, R>(a, (uint32_t)lb, (uint32_t)le, qs, lds_merge, out);
}

// ------------------------------------------------------------------------------------------ list-batched IVF scan
//
// For a batch of queries the (query, probed list) pairs are grouped BY LIST on the device, so that one pass over a
// list's rows serves a tile of up to T queries that probe it (the rows are streamed from HBM once per tile instead
// of once per query).  Work item = (list, query tile, row segment); a fixed-size grid walks the work items.

struct IvfPlanParams
{
    const int32_t * probes;   // [n_pairs] list id of pair i = q*nprobe + p (-1 = none)
    const int64_t * list_off; // [nlist+1]
    const int64_t * list_end; // nullable: end of list l (else list_off[l+1])
    const int64_t * whole_off; // nullable [nlist+1]: pairs whose list is EMPTY here are dropped from the plan (a shard of a
                               // multi-GPU index holds 1/world of the lists; 7 of 8 probes point at lists it does not own)
    uint32_t n_pairs;
    uint32_t nlist;
    uint32_t rows_per_block;
    uint32_t T;
    uint32_t * cnt;      // [nlist] pairs per list            (zeroed by the caller)
    uint32_t * fill;     // [nlist] scatter cursors            (zeroed by the caller)
    uint32_t * pair_off; // [nlist+1] exclusive scan of cnt
    uint32_t * work_off; // [nlist+1] exclusive scan of ceil(cnt/T) * ceil(len/rows_per_block)
    uint32_t * pairs;    // [n_pairs] pair indices grouped by list
    // optional second work partition of the SAME pairs over another row range (the sample launch of the shadow scan: block 0
    // of every list, tiles of T2 queries), computed by the same scan launch
    const int64_t * list_off2 = nullptr;
    const int64_t * list_end2 = nullptr;
    uint32_t T2 = 0;
    uint32_t * work_off2 = nullptr; // [nlist+1]
    uint32_t * zero = nullptr;      // a region of 32-bit words the scan launch clears on the way (the consumer's counters), ...
    uint32_t nzero = 0;             // ... and its length
    // measurement (option rerank_stats; nullable): += the rows of the lists that have pairs -- [0] in the first partition's row range
    // (when stat_first), [1] in the second's: what the launches over this plan read
    unsigned long long * stat_rows = nullptr;
    int stat_first = 1;
    // Row segments chosen ON THE DEVICE (the shadow list scan's main launch; nullable): the first partition's lists are cut into
    // segments of ~seg_out[0] 32-row blocks so that the launch has about seg_target_items work items whatever survived the probe
    // pruning -- a batch of 64 queries keeps ~64 lists, one item per list would leave three quarters of the CUs idle (plan_nseg)
    uint32_t * seg_out = nullptr;
    uint32_t seg_target_items = 0;
    uint32_t * pairs_out = nullptr; // nullable, HOST-visible (pinned): the scan launch leaves the number of pairs in the plan here (a hint for the
                                    // next search's choice of stages: read by the host without any synchronisation, stale by design)
};

/// Segments of a list of `blocks` 32-row blocks at a segment size of sb blocks (rounded to the nearest count, at least one;
/// an empty range has none), and the blocks per segment that go with it: shared by the plan kernel and the scan kernel.
__host__ __device__ inline uint32_t plan_nseg(uint32_t blocks, uint32_t sb)
{
    if (blocks == 0)
        return 0;
    const uint32_t n = (blocks + sb / 2) / sb;
    return n ? n : 1;
}

static __global__ void ivf_hist_kernel(const IvfPlanParams p)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < p.n_pairs)
    {
        int32_t l = p.probes[i];
        if (l >= 0 && (!p.whole_off || p.whole_off[l + 1] > p.whole_off[l]))
            atomicAdd(&p.cnt[l], 1u);
    }
}

constexpr uint32_t PLAN_LDS_LISTS = 8192; // 32 KB of LDS counters
constexpr uint32_t PLAN_CHUNK = 2048;     // pairs per block (8 per thread)

static __device__ __forceinline__ bool plan_pair_list(const IvfPlanParams & p, uint32_t i, uint32_t end, int32_t & l)
{
    l = i < end ? p.probes[i] : -1;
    return l >= 0 && (!p.whole_off || p.whole_off[l + 1] > p.whole_off[l]);
}

/// One block of 1024 threads: the exclusive scans over the lists (pairs, work items, and the second partition's work items).
/// FUSED (a small batch: n_pairs <= PLAN_FUSED_PAIRS, nlist <= PLAN_LDS_LISTS): the whole plan in this one launch -- the pairs are
/// counted per list in LDS first (each thread keeps its <= 8 pairs and their ranks inside their lists), the scans run on the LDS
/// counts, and the pairs go to pair_off[list] + rank.  p.cnt is written (the counts, for a later rescan), not read: nothing has to be
/// zeroed beforehand, and p.zero clears the consumer's counters on the way -- a batch of 32 queries spent 5 launches of ~5 us each on
/// this (a memset in two fills, histogram, scans, scatter), a fifth of its device time.
constexpr uint32_t PLAN_FUSED_PAIRS = 8192;
template <bool FUSED>
static __global__ __launch_bounds__(1024) void ivf_plan_scan_kernel(const IvfPlanParams p)
{
    uint32_t * h = nullptr;
    int32_t list[FUSED ? PLAN_FUSED_PAIRS / 1024 : 1];
    uint32_t rank[FUSED ? PLAN_FUSED_PAIRS / 1024 : 1];
    if constexpr (FUSED)
    {
        __shared__ uint32_t s_h[PLAN_LDS_LISTS];
        h = s_h;
        for (uint32_t l = threadIdx.x; l < p.nlist; l += 1024)
            h[l] = 0;
        __syncthreads();
#pragma unroll
        for (uint32_t u = 0; u < PLAN_FUSED_PAIRS / 1024; u++)
        {
            int32_t l;
            const bool ok = plan_pair_list(p, u * 1024 + threadIdx.x, p.n_pairs, l);
            list[u] = ok ? l : -1;
            rank[u] = ok ? atomicAdd(&h[l], 1u) : 0;
        }
        __syncthreads();
    }
    __shared__ uint32_t sp[2][1024];
    __shared__ uint32_t sw[2][1024];
    __shared__ uint32_t sv[2][1024];
    __shared__ uint32_t carry[3];
    __shared__ unsigned long long s_rows[2];
    const uint32_t tid = threadIdx.x;
    if (tid < 3)
        carry[tid] = 0;
    if (tid < 2)
        s_rows[tid] = 0;
    for (uint32_t i = tid; i < p.nzero; i += 1024)
        p.zero[i] = 0;
    __syncthreads();
    uint32_t seg_blocks = 0; // (0: segments of rows_per_block rows as given)
    // the first 1024 lists' counts and lengths stay in registers for the scan below
    uint32_t c_first = 0, len_first = 0;
    if (tid < p.nlist)
    {
        c_first = FUSED ? h[tid] : p.cnt[tid];
        len_first = (uint32_t)((p.list_end ? p.list_end[tid] : p.list_off[tid + 1]) - p.list_off[tid]);
    }
    if (p.seg_out)
    {
        // block-tiles of the whole launch -> blocks per segment for ~seg_target_items items, a multiple of 8 (one per wavefront)
        __shared__ unsigned long long s_units;
        if (tid == 0)
            s_units = 0;
        __syncthreads();
        unsigned long long mine = (unsigned long long)((c_first + p.T - 1) / p.T) * ((len_first + 31) >> 5);
        for (uint32_t l = tid + 1024; l < p.nlist; l += 1024)
        {
            const uint32_t c = FUSED ? h[l] : p.cnt[l];
            const uint32_t len = (uint32_t)((p.list_end ? p.list_end[l] : p.list_off[l + 1]) - p.list_off[l]);
            mine += (unsigned long long)((c + p.T - 1) / p.T) * ((len + 31) >> 5);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1)
            mine += __shfl_xor(mine, o);
        if ((tid & 63) == 0 && mine)
            atomicAdd(&s_units, mine);
        __syncthreads();
        const unsigned long long per = (s_units + p.seg_target_items - 1) / (p.seg_target_items ? p.seg_target_items : 1);
        seg_blocks = per >= 0x40000000ull ? 0x40000000u : (uint32_t)((per + 7) / 8 * 8);
        if (seg_blocks < 8)
            seg_blocks = 8;
        if (tid == 0)
            p.seg_out[0] = seg_blocks;
    }
    for (uint32_t base = 0; base < p.nlist; base += 1024)
    {
        const uint32_t l = base + tid;
        uint32_t c = 0, w = 0, v = 0;
        if (l < p.nlist)
        {
            c = base ? (FUSED ? h[l] : p.cnt[l]) : c_first;
            uint32_t len = base ? (uint32_t)((p.list_end ? p.list_end[l] : p.list_off[l + 1]) - p.list_off[l]) : len_first;
            w = ((c + p.T - 1) / p.T) * (seg_blocks ? plan_nseg((len + 31) >> 5, seg_blocks) : (len + p.rows_per_block - 1) / p.rows_per_block);
            if (p.stat_rows && c && p.stat_first)
                atomicAdd(&s_rows[0], (unsigned long long)len);
            if (p.work_off2)
            {
                const uint32_t len2 = (uint32_t)((p.list_end2 ? p.list_end2[l] : p.list_off2[l + 1]) - p.list_off2[l]);
                v = ((c + p.T2 - 1) / p.T2) * ((len2 + p.rows_per_block - 1) / p.rows_per_block);
                if (p.stat_rows && c)
                    atomicAdd(&s_rows[1], (unsigned long long)len2);
            }
        }
        int cur = 0;
        sp[0][tid] = c;
        sw[0][tid] = w;
        sv[0][tid] = v;
        __syncthreads();
        for (uint32_t d = 1; d < 1024; d <<= 1)
        {
            uint32_t vp = sp[cur][tid], vw = sw[cur][tid], vv = sv[cur][tid];
            if (tid >= d)
            {
                vp += sp[cur][tid - d];
                vw += sw[cur][tid - d];
                vv += sv[cur][tid - d];
            }
            sp[cur ^ 1][tid] = vp;
            sw[cur ^ 1][tid] = vw;
            sv[cur ^ 1][tid] = vv;
            cur ^= 1;
            __syncthreads();
        }
        const uint32_t cp = carry[0], cw = carry[1], cv = carry[2];
        if (l < p.nlist)
        {
            p.pair_off[l] = cp + sp[cur][tid] - c;
            if constexpr (FUSED)
            {
                h[l] = cp + sp[cur][tid] - c; // (this thread read the count of list l above: now the start of its range)
                p.cnt[l] = c;
            }
            p.work_off[l] = cw + sw[cur][tid] - w;
            if (p.work_off2)
                p.work_off2[l] = cv + sv[cur][tid] - v;
        }
        __syncthreads();
        if (tid == 1023)
        {
            carry[0] = cp + sp[cur][1023];
            carry[1] = cw + sw[cur][1023];
            carry[2] = cv + sv[cur][1023];
        }
        __syncthreads();
    }
    if (tid == 0)
    {
        p.pair_off[p.nlist] = carry[0];
        p.work_off[p.nlist] = carry[1];
        if (p.pairs_out)
            *p.pairs_out = carry[0];
        if (p.work_off2)
            p.work_off2[p.nlist] = carry[2];
        if (p.stat_rows)
        {
            atomicAdd(p.stat_rows, s_rows[0]);
            atomicAdd(p.stat_rows + 1, s_rows[1]);
        }
    }
    if constexpr (FUSED)
    {
        // (the last round's barriers are behind every thread: h holds the range starts)
#pragma unroll
        for (uint32_t u = 0; u < PLAN_FUSED_PAIRS / 1024; u++)
            if (list[u] >= 0)
                p.pairs[h[list[u]] + rank[u]] = u * 1024 + tid;
    }
}

static __global__ void ivf_scatter_kernel(const IvfPlanParams p)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < p.n_pairs)
    {
        int32_t l = p.probes[i];
        if (l >= 0 && (!p.whole_off || p.whole_off[l + 1] > p.whole_off[l]))
            p.pairs[p.pair_off[l] + atomicAdd(&p.fill[l], 1u)] = i;
    }
}

// The same two passes with the per-pair atomics moved into LDS: a block takes PLAN_CHUNK consecutive pairs, counts them
// per list in LDS, and touches global memory once per (block, non-empty list).  On the bench step (131 072 pairs over
// 1024 lists, hot lists holding > 1000 pairs) the one-atomic-per-pair kernels spend 36 us each queueing on the hot
// addresses; the scatter also hands every block ONE contiguous range per list (base from a single atomicAdd of the
// block's count, rank inside the block from the LDS atomic).
static __global__ __launch_bounds__(256) void ivf_hist_lds_kernel(const IvfPlanParams p)
{
    __shared__ uint32_t h[PLAN_LDS_LISTS];
    const uint32_t tid = threadIdx.x;
    for (uint32_t l = tid; l < p.nlist; l += 256)
        h[l] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * PLAN_CHUNK, end = min(base + PLAN_CHUNK, p.n_pairs);
#pragma unroll
    for (uint32_t u = 0; u < PLAN_CHUNK / 256; u++)
    {
        int32_t l;
        if (plan_pair_list(p, base + u * 256 + tid, end, l))
            atomicAdd(&h[l], 1u);
    }
    __syncthreads();
    for (uint32_t l = tid; l < p.nlist; l += 256)
        if (h[l])
            atomicAdd(&p.cnt[l], h[l]);
}

static __global__ __launch_bounds__(256) void ivf_scatter_lds_kernel(const IvfPlanParams p)
{
    __shared__ uint32_t h[PLAN_LDS_LISTS];
    const uint32_t tid = threadIdx.x;
    for (uint32_t l = tid; l < p.nlist; l += 256)
        h[l] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * PLAN_CHUNK, end = min(base + PLAN_CHUNK, p.n_pairs);
    int32_t list[PLAN_CHUNK / 256];
    uint32_t rank[PLAN_CHUNK / 256];
#pragma unroll
    for (uint32_t u = 0; u < PLAN_CHUNK / 256; u++)
    {
        int32_t l;
        const bool ok = plan_pair_list(p, base + u * 256 + tid, end, l);
        list[u] = ok ? l : -1;
        rank[u] = ok ? atomicAdd(&h[l], 1u) : 0;
    }
    __syncthreads();
    for (uint32_t l = tid; l < p.nlist; l += 256) // count -> start of this block's range inside list l
        if (h[l])
            h[l] = p.pair_off[l] + atomicAdd(&p.fill[l], h[l]);
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < PLAN_CHUNK / 256; u++)
        if (list[u] >= 0)
            p.pairs[h[list[u]] + rank[u]] = base + u * 256 + tid;
}

/// grid: any size; block b handles work items b, b + gridDim.x, ...
template <int METRIC, int T, int R>
__global__ __launch_bounds__(BLOCK) void ivf_batched_scan_kernel(const ScanParams a)
{
    float4 * qs = reinterpret_cast<float4 *>(msvs_smem);
    uint64_t * lds_merge = reinterpret_cast<uint64_t *>(msvs_smem + (size_t)T * a.ld4 * 16);
    const uint32_t total = a.work_off[a.nlist];
    // Work order inside a list is [segment][query tile] (tile fastest): consecutive work items read the SAME rows
    // for different query tiles.  Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, a speed
    // assumption only), so slot s = b + i*gridDim.x is mapped to work item (s % 8) * per_xcd + s / 8: every XCD
    // walks one contiguous range, and the tiles sharing a row segment run back to back on ONE XCD's L2.
    const uint32_t per_xcd = (total + 7) / 8;
    for (uint32_t s = blockIdx.x; s < 8 * per_xcd; s += gridDim.x)
    {
        const uint32_t w = a.xcd_order ? (s & 7) * per_xcd + (s >> 3) : s;
        if (w >= total || (a.xcd_order && (s >> 3) >= per_xcd))
            continue;
        // the list owning work item w: work_off[l] <= w < work_off[l+1]
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
        const int64_t lbeg = a.list_off[l], lend = a.list_off[l + 1];
        const uint32_t nseg = ((uint32_t)(lend - lbeg) + a.rows_per_block - 1) / a.rows_per_block;
        const uint32_t local = w - a.work_off[l];
        const uint32_t pe = a.pair_off[l + 1];
        const uint32_t ntile = (pe - a.pair_off[l] + T - 1) / T;
        const uint32_t seg = local / ntile, tile = local - seg * ntile;
        (void)nseg;
        const uint32_t pb = a.pair_off[l] + tile * T;
        uint32_t qidx[T];
        uint64_t * out[T];
#pragma unroll
        for (int t = 0; t < T; t++)
        {
            uint32_t pi = pb + t < pe ? pb + t : pe - 1; // short tiles repeat their last pair (same slot, same values)
            uint32_t qp = a.pairs[pi];
            qidx[t] = qp / a.nprobe;
            out[t] = a.partial + ((size_t)qp * a.seg_max + seg) * a.k;
        }
        __syncthreads();
        stage_queries<T>(a, qidx, qs);
        const int64_t rb = lbeg + (int64_t)seg * a.rows_per_block;
        const int64_t re = rb + a.rows_per_block < lend ? rb + a.rows_per_block : lend;
        scan_rows<METRIC, T, R>(a, (uint32_t)rb, (uint32_t)re, qs, lds_merge, out);
    }
}

// ------------------------------------------------------------------------------------------ merges of sorted lists

constexpr uint32_t HEADS_CAP = 6144; // keys a merge block stages in LDS (48 KiB; the block stays below 64 KiB)

/// LDS bytes of a merge block: staged keys + output + per-list cursors + per-probe list prefix (the 5*k key slots of
/// the threshold-method fallback alias the staged-key region).
inline size_t merge_lds_bytes(uint32_t k, uint32_t nprobe)
{
    size_t heads = (size_t)HEADS_CAP * 8 + (size_t)k * 8 + (size_t)HEADS_CAP * 2 + ((size_t)nprobe + 1) * 4 + 64;
    size_t fallback = (size_t)5 * k * 8;
    return heads > fallback ? heads : fallback;
}

template <int CTRL>
__device__ __forceinline__ uint64_t dpp64(uint64_t v)
{
    uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xF, 0xF, false);
    uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xF, 0xF, false);
    return (uint64_t)hi << 32 | lo;
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}

/// Wave-wide minimum of a u32, result uniform: 4 DPP steps inside each 16-lane row (v_min_u32 with a DPP operand each),
/// then 4 readlanes + scalar mins across the rows.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    v = min(v, dpp32<0xB1>(v));  // lane ^ 1
    v = min(v, dpp32<0x4E>(v));  // lane ^ 2
    v = min(v, dpp32<0x141>(v)); // row_half_mirror
    v = min(v, dpp32<0x140>(v)); // row_mirror
    const uint32_t r0 = __builtin_amdgcn_readlane((int)v, 0), r1 = __builtin_amdgcn_readlane((int)v, 16),
                   r2 = __builtin_amdgcn_readlane((int)v, 32), r3 = __builtin_amdgcn_readlane((int)v, 48);
    return min(min(r0, r1), min(r2, r3));
}

/// Wave-wide maximum of a u32, result uniform.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    v = max(v, dpp32<0xB1>(v));
    v = max(v, dpp32<0x4E>(v));
    v = max(v, dpp32<0x141>(v));
    v = max(v, dpp32<0x140>(v));
    const uint32_t r0 = __builtin_amdgcn_readlane((int)v, 0), r1 = __builtin_amdgcn_readlane((int)v, 16),
                   r2 = __builtin_amdgcn_readlane((int)v, 32), r3 = __builtin_amdgcn_readlane((int)v, 48);
    return max(max(r0, r1), max(r2, r3));
}

/// Wave-wide sum of a u32, result uniform (the same butterfly: every step pairs disjoint groups).
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    v += dpp32<0xB1>(v);
    v += dpp32<0x4E>(v);
    v += dpp32<0x141>(v);
    v += dpp32<0x140>(v);
    const uint32_t r0 = __builtin_amdgcn_readlane((int)v, 0), r1 = __builtin_amdgcn_readlane((int)v, 16),
                   r2 = __builtin_amdgcn_readlane((int)v, 32), r3 = __builtin_amdgcn_readlane((int)v, 48);
    return r0 + r1 + r2 + r3;
}

/// Wave-wide minimum of a u64, result uniform: the minimum of the high words, then the minimum of the low words among
/// the lanes that hold it (two 32-bit reductions of single-instruction steps instead of one chain of 64-bit
/// compare-and-selects: this sits on the serial path of every heads-merge round).
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v)
{
    const uint32_t hi = (uint32_t)(v >> 32), mh = wave_min_u32(hi);
    const uint32_t ml = wave_min_u32(hi == mh ? (uint32_t)v : 0xFFFFFFFFu);
    return (uint64_t)mh << 32 | ml;
}

/// P sorted lists of length L staged in LDS (list i at keys[i*L ...]) -> the k smallest keys in out[0..k) (LDS), by
/// ONE wavefront: lane l owns lists l, l+64, ... and keeps the smallest unconsumed head among them; each of the k
/// rounds is a wave-wide minimum + a rescan by the winning lane only.  Cost ~ k * (P/64 + log 64) LDS reads, against
/// ~ P*L*ln(...) serialised insertions for the threshold method when most partial lists are short.
/// idx: u16[P] LDS scratch.  Equal keys (duplicates across lists) are all kept.
__device__ __forceinline__ void wave_heads_merge(const uint64_t * keys, uint32_t P, uint32_t L, uint16_t * idx,
                                                 uint64_t * out, uint32_t k, uint32_t lane)
{
    for (uint32_t i = lane; i < P; i += WAVE)
        idx[i] = 0;
    uint64_t best = KEY_NONE;
    uint32_t bl = 0;
    auto rescan = [&]() {
        best = KEY_NONE;
        for (uint32_t i = lane; i < P; i += WAVE)
        {
            const uint32_t h = idx[i];
            const uint64_t key = h < L ? keys[i * L + h] : KEY_NONE;
            if (key < best)
            {
                best = key;
                bl = i;
            }
        }
    };
    rescan();
    for (uint32_t r = 0; r < k; r++)
    {
        const uint64_t m = wave_min_u64(best);
        if (lane == 0)
            out[r] = m;
        // equal keys in two lanes (the same id at the same distance in two lists): one copy per round, lowest lane first
        const uint64_t tie = __ballot(m != KEY_NONE && best == m);
        if (tie && lane == (uint32_t)__builtin_ctzll(tie))
        {
            idx[bl] = (uint16_t)(idx[bl] + 1);
            rescan();
        }
    }
}

struct IvfMergeParams
{
    const uint64_t * partial; // [(q*nprobe + p)*seg_max + s][k]
    const int32_t * probes;   // [nq][nprobe]
    const int64_t * list_off;
    uint32_t nprobe, seg_max, rows_per_block, k;
    int64_t * out_ids;
    float * out_dis;
    int cosine;
    uint64_t * out_keys;     // non-null: write the merged keys [nq][k] instead of (ids, distances)
    int32_t * out_probes;    // non-null: write [nq][k] int32 ids instead (the coarse quantiser's probe lists)
    const uint32_t * qmap;   // subset kernel: the queries to merge ...
    const uint32_t * qcount; // ... and how many of them (device side)
    uint32_t slot_base, slot_cap; // subset kernel: the round's window of the fail list; partial is indexed by f - slot_base
};

/// Top-k of query q over the valid segments of its probed lists, whose partial lists sit in slot `slot` of a.partial
/// (the query itself, or its rank in a fail list).  All BLOCK threads; uniform control flow.
template <int METRIC, int R>
__device__ __forceinline__ void ivf_merge_query(const IvfMergeParams & a, const uint32_t q, const uint32_t slot)
{
    uint64_t * lds = reinterpret_cast<uint64_t *>(msvs_smem);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, k = a.k;
    auto emit = [&](uint32_t i, uint64_t key) {
        const size_t o = (size_t)q * k + i;
        if (a.out_keys)
            a.out_keys[o] = key;
        else if (a.out_probes)
            a.out_probes[o] = key == KEY_NONE ? -1 : (int32_t)(uint32_t)key;
        else
        {
            a.out_ids[o] = key == KEY_NONE ? -1 : (int64_t)(uint32_t)key;
            const float v = key_value<METRIC>(key);
            a.out_dis[o] = a.cosine ? __fsub_rn(1.0f, v) : v;
        }
    };

    // Fast path: all valid partial lists of the query fit in LDS -> compact them and run the heads merge.
    {
        uint64_t * keys = lds;                                                  // [HEADS_CAP]
        uint64_t * outk = lds + HEADS_CAP;                                      // [k]
        uint16_t * idx = reinterpret_cast<uint16_t *>(outk + k);                // [HEADS_CAP]
        uint32_t * lbase = reinterpret_cast<uint32_t *>(idx + HEADS_CAP);       // [nprobe + 1] list index prefix
        for (uint32_t p = tid; p < a.nprobe; p += BLOCK) // segment counts: independent global loads, all in flight
        {
            const int32_t l = a.probes[(size_t)q * a.nprobe + p];
            uint32_t ns = 0;
            if (l >= 0)
            {
                const uint32_t len = (uint32_t)(a.list_off[l + 1] - a.list_off[l]);
                ns = (len + a.rows_per_block - 1) / a.rows_per_block;
            }
            lbase[p + 1] = ns;
        }
        __syncthreads();
        if (tid == 0) // exclusive prefix over <= 256 LDS values
        {
            uint32_t acc = 0;
            for (uint32_t p = 0; p < a.nprobe; p++)
            {
                const uint32_t ns = lbase[p + 1];
                lbase[p] = acc;
                acc += ns;
            }
            lbase[a.nprobe] = acc;
        }
        __syncthreads();
        const uint32_t P = lbase[a.nprobe];
        if ((uint64_t)P * k <= HEADS_CAP)
        {
            // one flat loop over the padded (probe, segment-slot, rank) space: every load is independent, so the
            // copy costs one memory latency instead of one per probe
            const uint32_t slots = a.seg_max * k, padded = a.nprobe * slots;
            const uint64_t * src = a.partial + (size_t)slot * padded;
            for (uint32_t i = tid; i < padded; i += BLOCK)
            {
                const uint32_t p = i / slots, r = i - p * slots;
                if (r < (lbase[p + 1] - lbase[p]) * k)
                    keys[(size_t)lbase[p] * k + r] = src[i];
            }
            __syncthreads();
            if (wave == 0)
                wave_heads_merge(keys, P, k, idx, outk, k, lane);
            __syncthreads();
            for (uint32_t i = tid; i < k; i += BLOCK)
                emit(i, outk[i]);
            return;
        }
        __syncthreads();
    }

    WaveTopK<R> top;
    top.init();
    for (uint32_t p = wave; p < a.nprobe; p += 4)
    {
        const int32_t l = __builtin_amdgcn_readfirstlane(a.probes[(size_t)q * a.nprobe + p]);
        if (l < 0)
            continue;
        const uint32_t len = (uint32_t)(a.list_off[l + 1] - a.list_off[l]);
        const uint32_t n = ((len + a.rows_per_block - 1) / a.rows_per_block) * k;
        const uint64_t * src = a.partial + ((size_t)slot * a.nprobe + p) * a.seg_max * k;
        for (uint32_t base = 0; base < n; base += 4 * WAVE)
        {
            uint64_t key[4];
#pragma unroll
            for (int u = 0; u < 4; u++)
            {
                uint32_t i = base + u * WAVE + lane;
                key[u] = i < n ? src[i] : KEY_NONE;
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                top.offer(key[u], k, lane);
        }
    }
    top.store(lds + wave * k, k, lane);
    __syncthreads();
    uint64_t * merged = lds + 4 * k;
    block_rank_merge(lds, k, merged, k, tid);
    for (uint32_t i = tid; i < k; i += BLOCK)
        emit(i, merged[i]);
}

/// One block per query.
template <int METRIC, int R>
__global__ __launch_bounds__(BLOCK) void ivf_merge_kernel(const IvfMergeParams a)
{
    ivf_merge_query<METRIC, R>(a, blockIdx.x, blockIdx.x);
}

/// Block b merges queries a.qmap[b], a.qmap[b + gridDim.x], ... up to *a.qcount (usually 0).
template <int METRIC, int R>
__global__ __launch_bounds__(BLOCK) void ivf_merge_subset_kernel(const IvfMergeParams a)
{
    // fail-list entries [slot_base, slot_base + slot_cap): one round of the fallback (its buffers hold slot_cap queries)
    const uint32_t nf = *a.qcount < a.slot_base + a.slot_cap ? *a.qcount : a.slot_base + a.slot_cap;
    for (uint32_t f = a.slot_base + blockIdx.x; f < nf; f += gridDim.x)
    {
        ivf_merge_query<METRIC, R>(a, a.qmap[f], f - a.slot_base);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ final merge

struct MergeParams
{
    const uint64_t * partial; // [nq][n_lists][k]
    uint32_t n_lists;
    uint32_t k;
    int64_t * out_ids;    // [nq][k] (mode 0)
    float * out_dis;      // [nq][k] (mode 0)
    int32_t * out_probes; // [nq][k] (mode 1)
    float * out_probe_dis = nullptr; // nullable [nq][k] (mode 1): the keys' values -- the canonical distance of the query to every probe
    uint64_t * out_keys;  // [nq][k] (mode 2: keep keys, for multi-level merges)
    int mode;
    int cosine; // mode 0: report 1 - ip
};

/// One block per query: top-k of n_lists*k keys.
template <int METRIC, int R>
__global__ __launch_bounds__(BLOCK) void merge_kernel(const MergeParams a)
{
    uint64_t * lds = reinterpret_cast<uint64_t *>(msvs_smem);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, k = a.k, q = blockIdx.x;
    const uint64_t * src = a.partial + (size_t)q * a.n_lists * k;
    const uint64_t total = (uint64_t)a.n_lists * k;

    if (total <= HEADS_CAP)
    {
        // Fast path: stage all lists in LDS and run the heads merge (see wave_heads_merge)
        uint64_t * keys = lds;
        uint64_t * outk = lds + HEADS_CAP;
        uint16_t * idx = reinterpret_cast<uint16_t *>(outk + k);
        for (uint32_t i = tid; i < (uint32_t)total; i += BLOCK)
            keys[i] = src[i];
        __syncthreads();
        if (wave == 0)
            wave_heads_merge(keys, a.n_lists, k, idx, outk, k, lane);
        __syncthreads();
        for (uint32_t i = tid; i < k; i += BLOCK)
        {
            uint64_t key = outk[i];
            size_t o = (size_t)q * k + i;
            if (a.mode == 1)
            {
                a.out_probes[o] = key == KEY_NONE ? -1 : (int32_t)(uint32_t)key;
                if (a.out_probe_dis)
                    a.out_probe_dis[o] = key_value<METRIC>(key);
            }
            else if (a.mode == 2)
                a.out_keys[o] = key;
            else
            {
                a.out_ids[o] = key == KEY_NONE ? -1 : (int64_t)(uint32_t)key;
                float v = key_value<METRIC>(key);
                a.out_dis[o] = a.cosine ? __fsub_rn(1.0f, v) : v;
            }
        }
        return;
    }

    WaveTopK<R> top;
    top.init();
    for (uint64_t base = 0; base < total; base += 4 * BLOCK)
    {
        uint64_t key[4]; // 4 independent loads in flight before the first (serialising) offer
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            uint64_t i = base + u * BLOCK + tid;
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
        uint64_t key = merged[i];
        size_t o = (size_t)q * k + i;
        if (a.mode == 1)
        {
            a.out_probes[o] = key == KEY_NONE ? -1 : (int32_t)(uint32_t)key;
            if (a.out_probe_dis)
                a.out_probe_dis[o] = key_value<METRIC>(key);
        }
        else if (a.mode == 2)
            a.out_keys[o] = key;
        else
        {
            a.out_ids[o] = key == KEY_NONE ? -1 : (int64_t)(uint32_t)key;
            float v = key_value<METRIC>(key);
            a.out_dis[o] = a.cosine ? __fsub_rn(1.0f, v) : v;
        }
    }
}

/// The merge of a FEW short lists (nparts * k <= 256: the result exchange of a sharded search, the per-part lists of a multi-part
/// search) straight from the (ids, dis) arrays, one WAVEFRONT per query: every key ranks itself among the others through a
/// wave-private LDS slab and writes itself to its output slot -- no packing pass, no block per query (the general pair of launches costs
/// ~40 us per 4096 queries whatever the lists hold: eight barriers around eighty keys).  Same keys, same order, same outputs as
/// pack_keys_kernel + merge_kernel (mode 0): ascending (distance | descending for IP, id); an id < 0 or a distance that is not
/// strictly better than the neutral value is no entry; unfilled slots are id -1, distance +-FLT_MAX.
template <int METRIC>
__global__ __launch_bounds__(BLOCK) void merge_small_kernel(const int64_t * ids, size_t ids_stride, const float * dis, size_t dis_stride,
                                                            uint32_t nparts, uint32_t nq, uint32_t k, int64_t * out_ids, float * out_dis)
{
    __shared__ uint64_t s_keys[BLOCK / WAVE][256];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t q = blockIdx.x * (BLOCK / WAVE) + wave;
    if (q >= nq)
        return;
    const uint32_t n = nparts * k;
    uint64_t mine[4];
    uint32_t valid = 0;
#pragma unroll
    for (int u = 0; u < 4; u++)
    {
        const uint32_t i = (uint32_t)u * WAVE + lane;
        uint64_t key = KEY_NONE;
        if (i < n)
        {
            const uint32_t p = i / k, j = i - p * k;
            const size_t rem = (size_t)q * k + j;
            const int64_t id = ids[p * ids_stride + rem];
            const float d = dis[p * dis_stride + rem];
            key = id < 0 ? KEY_NONE : make_key<METRIC>(d, (uint32_t)id);
        }
        mine[u] = key;
        s_keys[wave][i] = key;
        valid += (uint32_t)__popcll(__ballot(key != KEY_NONE));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // wave-private LDS: the wavefront's own writes, in order
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < 4; u++)
    {
        const uint32_t i = (uint32_t)u * WAVE + lane;
        if ((uint32_t)u * WAVE >= n) // (uniform)
            break;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; j++)
        {
            const uint64_t kj = s_keys[wave][j];
            rank += kj < mine[u] || (kj == mine[u] && j < i) ? 1u : 0u;
        }
        if (mine[u] != KEY_NONE && rank < k)
        {
            const size_t o = (size_t)q * k + rank;
            out_ids[o] = (int64_t)(uint32_t)mine[u];
            out_dis[o] = key_value<METRIC>(mine[u]);
        }
    }
    for (uint32_t r = valid + lane; r < k; r += WAVE)
    {
        out_ids[(size_t)q * k + r] = -1;
        out_dis[(size_t)q * k + r] = key_value<METRIC>(KEY_NONE);
    }
}

/// (ids, dis) lists of `nparts` shards -> keys in the merge kernel's layout [nq][nparts][k].
/// Shard p's lists start at ids + p * ids_stride / dis + p * dis_stride (element strides), each [nq][k].
template <int METRIC>
__global__ void pack_keys_kernel(const int64_t * ids, size_t ids_stride, const float * dis, size_t dis_stride,
                                 uint64_t * keys, uint32_t nparts, uint32_t nq, uint32_t k)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per_part = (size_t)nq * k;
    if (i >= per_part * nparts)
        return;
    const uint32_t p = (uint32_t)(i / per_part);
    const size_t rem = i - (size_t)p * per_part;
    const uint32_t q = (uint32_t)(rem / k), j = (uint32_t)(rem - (size_t)q * k);
    const int64_t id = ids[p * ids_stride + rem];
    const float d = dis[p * dis_stride + rem];
    keys[((size_t)q * nparts + p) * k + j] = id < 0 ? KEY_NONE : make_key<METRIC>(d, (uint32_t)id);
}

// ------------------------------------------------------------------------------------------ normalisation

/// VectorDataset::normalize(): strictly sequential f32 sum of squares (the reference's order), rows with sum <
/// FLT_EPSILON untouched.  ld = row stride in floats (>= d).  One WAVEFRONT per row: the lanes stage the row in LDS
/// (coalesced), lane 0 walks it in order, all lanes divide -- a thread per row read its 768 elements one dependent global
/// load at a time: 166 us for the 64 queries of a cosine batch (12 % of the 10M-row step).  dynamic LDS: d * 4 bytes.
static __global__ __launch_bounds__(WAVE) void normalize_rows_kernel(float * x, size_t n, uint32_t d, uint32_t ld)
{
    float * row = reinterpret_cast<float *>(msvs_smem);
    __shared__ float s_sum;
    const uint32_t lane = threadIdx.x;
    for (size_t r = blockIdx.x; r < n; r += gridDim.x)
    {
        float * p = x + r * ld;
        for (uint32_t j = lane; j < d; j += WAVE)
            row[j] = p[j];
        __syncthreads();
        if (lane == 0)
        {
            float sum = 0.f;
            for (uint32_t j = 0; j < d; j++)
                sum = __fadd_rn(sum, __fmul_rn(row[j], row[j]));
            s_sum = sum;
        }
        __syncthreads();
        const float sum = s_sum;
        if (!(sum < 1.1920928955078125e-7f))
        {
            // NOT __fsqrt_rn: without OCML_BASIC_ROUNDED_OPERATIONS that is v_sqrt_f32 (1 ulp); sqrtf() is IEEE-correct
            // under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt, and so is operator/ behind __fdiv_rn.
            const float s = sqrtf(sum);
            for (uint32_t j = lane; j < d; j += WAVE)
                p[j] = __fdiv_rn(row[j], s);
        }
        __syncthreads();
    }
}

/// grid of normalize_rows_kernel: one block per row up to a few waves of the device
inline unsigned normalize_rows_grid(size_t n) { return (unsigned)(n < 262144 ? (n ? n : 1) : 262144); }

/// Copies rows with stride conversion (d -> ld, zero padding) on the device.
static __global__ void pad_rows_kernel(const float * src, float * dst, size_t n, uint32_t d, uint32_t ld)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = n * ld;
    if (i >= total)
        return;
    size_t r = i / ld;
    uint32_t c = (uint32_t)(i - r * ld);
    dst[i] = c < d ? src[r * d + c] : 0.f;
}

}
