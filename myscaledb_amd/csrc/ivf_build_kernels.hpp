// ivf_build_kernels.hpp -- device code of the IVFFLAT build feed (k-means training and list assignment).
//
// This is NOT the parity-critical scan: the trained structure of an IVF index is "parity unpinned"
// (only recall, not structure, can be compared), so nearest-centroid assignment is done as a
// GEMM on the FP32 matrix cores: score(j,i) = |c_j|^2 - 2 <c_j, x_i> (L2) or -<c_j, x_i> (IP) with
// v_mfma_f32_32x32x2_f32 (exact f32 products, one rounding per fma), LDS-tiled 128 centroids x 128 rows per block.
// The search path re-derives everything it returns with the canonical arithmetic of scan_kernels.hpp.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace msvs
{

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int AS_TM = 128; // centroids per block tile (MFMA rows)
constexpr int AS_TN = 128; // data rows per block tile (MFMA cols)
constexpr int AS_TK = 16;
constexpr int AS_LD = AS_TM + 4;

/// cnorm[j] = sum_k c[j][k]^2
static __global__ void row_sqnorm_kernel(const float * c, float * out, uint32_t n, uint32_t d, uint32_t ld)
{
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n)
        return;
    const float * p = c + (size_t)r * ld;
    float s = 0.f;
    for (uint32_t j = 0; j < d; j++)
        s = fmaf(p[j], p[j], s);
    out[r] = s;
}

/// assign[i] = argmin_j score(j, i), ties -> lowest j.  grid: ceil(n / 128) blocks of 256 threads.
template <bool IP>
__global__ __launch_bounds__(256) void assign_kernel(const float * __restrict__ X, size_t n, const float * __restrict__ C,
                                                     const float * __restrict__ cnorm, uint32_t nlist, uint32_t d,
                                                     uint32_t ld, int32_t * __restrict__ assign,
                                                     float * __restrict__ best_score /* nullable */)
{
    __shared__ float As[AS_TK * AS_LD];
    __shared__ float Bs[AS_TK * AS_LD];
    __shared__ float red_v[2][AS_TN];
    __shared__ int red_i[2][AS_TN];

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t wm = wave >> 1, wn = wave & 1; // wave tile: centroids [wm*64, +64), rows [wn*64, +64)
    const uint32_t h = lane >> 5, l32 = lane & 31;
    const size_t row0 = (size_t)blockIdx.x * AS_TN;

    float bestv[2] = {3.402823466e+38f, 3.402823466e+38f};
    int besti[2] = {0x7fffffff, 0x7fffffff};

    const uint32_t lr = tid >> 2, lk = (tid & 3) * 4; // loader: rows lr, lr+64; k offset lk..lk+3

    for (uint32_t c0 = 0; c0 < nlist; c0 += AS_TM)
    {
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    acc[a][b][r] = 0.f;

        for (uint32_t k0 = 0; k0 < d; k0 += AS_TK)
        {
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 2; it++)
            {
                uint32_t r = lr + it * 64;
                uint32_t cj = c0 + r < nlist ? c0 + r : nlist - 1;
                size_t xi = row0 + r < n ? row0 + r : n - 1;
                float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = a4;
                if (k0 + lk < ld)
                {
                    a4 = *reinterpret_cast<const float4 *>(C + (size_t)cj * ld + k0 + lk);
                    b4 = *reinterpret_cast<const float4 *>(X + xi * ld + k0 + lk);
                }
                As[(lk + 0) * AS_LD + r] = a4.x;
                As[(lk + 1) * AS_LD + r] = a4.y;
                As[(lk + 2) * AS_LD + r] = a4.z;
                As[(lk + 3) * AS_LD + r] = a4.w;
                Bs[(lk + 0) * AS_LD + r] = b4.x;
                Bs[(lk + 1) * AS_LD + r] = b4.y;
                Bs[(lk + 2) * AS_LD + r] = b4.z;
                Bs[(lk + 3) * AS_LD + r] = b4.w;
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < AS_TK / 2; kk++)
            {
                const float * ap = As + (kk * 2 + h) * AS_LD + wm * 64 + l32;
                const float * bp = Bs + (kk * 2 + h) * AS_LD + wn * 64 + l32;
                float a0 = ap[0], a1 = ap[32], b0 = bp[0], b1 = bp[32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        // D[row = centroid][col = data row]: lane holds col l32, rows (r&3) + 8*(r>>2) + 4*h
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int r = 0; r < 16; r++)
            {
                uint32_t cj = c0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (cj < nlist)
                {
                    float cn = IP ? 0.f : cnorm[cj];
#pragma unroll
                    for (int nt = 0; nt < 2; nt++)
                    {
                        float dot = acc[mt][nt][r];
                        float sc = IP ? -dot : fmaf(-2.f, dot, cn);
                        if (sc < bestv[nt] || (sc == bestv[nt] && (int)cj < besti[nt]))
                        {
                            bestv[nt] = sc;
                            besti[nt] = (int)cj;
                        }
                    }
                }
            }
    }
    // combine the two half-waves (same column, different row subsets)
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
    {
        float ov = __shfl_xor(bestv[nt], 32, 64);
        int oi = __shfl_xor(besti[nt], 32, 64);
        if (ov < bestv[nt] || (ov == bestv[nt] && oi < besti[nt]))
        {
            bestv[nt] = ov;
            besti[nt] = oi;
        }
    }
    __syncthreads();
    if (h == 0)
    {
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
        {
            red_v[wm][wn * 64 + nt * 32 + l32] = bestv[nt];
            red_i[wm][wn * 64 + nt * 32 + l32] = besti[nt];
        }
    }
    __syncthreads();
    if (tid < AS_TN && row0 + tid < n)
    {
        float v0 = red_v[0][tid], v1 = red_v[1][tid];
        int i0 = red_i[0][tid], i1 = red_i[1][tid];
        bool take1 = v1 < v0 || (v1 == v0 && i1 < i0);
        const int best = take1 ? i1 : i0;
        assign[row0 + tid] = best == 0x7fffffff ? 0 : best; // rows with NaN / overflowing scores go to list 0
        if (best_score)
            best_score[row0 + tid] = take1 ? v1 : v0;
    }
}

/// New centroid = mean of its members, summed in member order (deterministic).  One block per list,
/// thread per dimension (strided).  members[off[j] .. off[j+1]) are row indices into X.
static __global__ void centroid_update_kernel(const float * __restrict__ X, uint32_t d, uint32_t ld,
                                       const int64_t * __restrict__ off, const uint32_t * __restrict__ members,
                                       float * __restrict__ C)
{
    const uint32_t j = blockIdx.x;
    const int64_t b = off[j], e = off[j + 1];
    if (e <= b)
        return; // empty cluster keeps its previous centroid
    const float inv = 1.0f / (float)(e - b);
    for (uint32_t c = threadIdx.x; c < ld; c += blockDim.x)
    {
        float s = 0.f;
        if (c < d)
            for (int64_t m = b; m < e; m++)
                s += X[(size_t)members[m] * ld + c];
        C[(size_t)j * ld + c] = c < d ? s * inv : 0.f;
    }
}

/// dst[pos[i]] = src[i] (rows of ld floats, float4 granularity): lays rows out list-major.
static __global__ void scatter_rows_kernel(const float4 * __restrict__ src, float4 * __restrict__ dst,
                                    const uint32_t * __restrict__ pos, size_t n, uint32_t ld4)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = n * ld4;
    if (i >= total)
        return;
    size_t r = i / ld4;
    uint32_t c = (uint32_t)(i - r * ld4);
    dst[(size_t)pos[r] * ld4 + c] = src[i];
}

/// dst[i] = src[idx[i]] (gather rows)
static __global__ void gather_rows_kernel(const float4 * __restrict__ src, float4 * __restrict__ dst,
                                   const uint32_t * __restrict__ idx, size_t n, uint32_t ld4)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = n * ld4;
    if (i >= total)
        return;
    size_t r = i / ld4;
    uint32_t c = (uint32_t)(i - r * ld4);
    dst[i] = src[(size_t)idx[r] * ld4 + c];
}


/// radius[l] = an upper bound of max over the rows x of list l of ||x - c_l|| (Euclidean), one block per list: differences and sums
/// in double (the inputs are exact f32: the error is ~1e-13 relative), the square root rounded up and widened by 1e-6.  What the
/// probe pruning of the shadow list scan subtracts from a query's distance to the centroid (h16_scan_kernels.hpp).
static __global__ __launch_bounds__(256) void list_radius_kernel(const float * vecs, const float * centroids, const int64_t * list_off,
                                                                  uint32_t d, uint32_t ld, float * radius)
{
    __shared__ double s_max[4];
    const uint32_t l = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t lbeg = list_off[l], lend = list_off[l + 1];
    const float * c = centroids + (size_t)l * ld;
    double mx = 0.0;
    for (int64_t r = lbeg + wave; r < lend; r += 4) // a wavefront per row
    {
        const float * x = vecs + (size_t)r * ld;
        double s = 0.0;
        for (uint32_t e = lane; e < d; e += 64)
        {
            const double df = (double)x[e] - (double)c[e];
            s += df * df;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1)
            s += __shfl_xor(s, o);
        mx = s > mx || !(s == s) ? s : mx; // NaN sticks: the list is then never pruned
    }
    if (lane == 0)
        s_max[wave] = mx;
    __syncthreads();
    if (tid == 0)
    {
        double m = s_max[0];
        for (int w = 1; w < 4; w++)
            m = s_max[w] > m || !(s_max[w] == s_max[w]) ? s_max[w] : m;
        const double r = sqrt(m) * (1.0 + 1e-6) + 1e-30;
        float rf = (float)r;
        if ((double)rf < r)
            rf = nextafterf(rf, INFINITY);
        radius[l] = r == r ? rf : INFINITY;
    }
}

}
