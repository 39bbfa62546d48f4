// Feasibility: the exhaustive pass with the queries' A operands RESIDENT IN REGISTERS (2 x 32 queries x 768 dims per wavefront = 384 registers,
// one wavefront per SIMD) and the rows staged once per workgroup in LDS (double-buffered 48 KB row blocks, one barrier per block).
// hipcc --offload-arch=gfx950 -O3 -o tools/micro/areg tools/micro/areg.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int NS = 48; // 16-element steps per row (d = 768)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k(const u32x4 * __restrict__ Qa /* [qblock][NS][64] */, const u32x4 * __restrict__ H /* [block][NS][64] */, float * out, int nblk, unsigned span_mask)
{
    __shared__ __attribute__((aligned(16))) u32x4 rows_s[2][NS * 64]; // 2 x 48 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    half8 a[2][NS];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int s = 0; s < NS; s++)
            a[t][s] = __builtin_bit_cast(half8, Qa[((size_t)(blockIdx.x * 8 + wave * 2 + t) * NS + s) * 64 + lane]);
    f32x16 total[2];
    for (int t = 0; t < 2; t++)
        for (int r = 0; r < 16; r++)
            total[t][r] = 0.f;
    // stage block 0
    const u32x4 * src = H + (size_t)((blockIdx.x * 37u) & span_mask) * NS * 64;
#pragma unroll
    for (int i = 0; i < 12; i++)
        rows_s[0][i * 256 + tid] = src[i * 256 + tid];
    __syncthreads();
    for (int b = 0; b < nblk; b++)
    {
        const int cur = b & 1;
        // next block -> registers (12 x 16 B per thread), written to LDS after the multiply
        const u32x4 * nx = H + (size_t)((blockIdx.x * 37u + b + 1) & span_mask) * NS * 64;
        u32x4 st[12];
#pragma unroll
        for (int i = 0; i < 12; i++)
            st[i] = nx[i * 256 + tid];
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                acc[t][r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; s++)
        {
            const half8 bf = __builtin_bit_cast(half8, rows_s[cur][s * 64 + lane]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][s], bf, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][s], bf, acc[1], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                total[t][r] += acc[t][r] < 0.5f ? 1.f : 0.f; // (a stand-in for the threshold test)
#pragma unroll
        for (int i = 0; i < 12; i++)
            rows_s[cur ^ 1][i * 256 + tid] = st[i];
        __syncthreads();
    }
    float s = 0.f;
    for (int t = 0; t < 2; t++)
        for (int r = 0; r < 16; r++)
            s += total[t][r];
    out[blockIdx.x * 256 + tid] = s;
}


// Variant: EIGHT wavefronts (two per SIMD), ONE 32-query block per wavefront in registers (192): every B fragment read from LDS feeds one MFMA.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k8(const u32x4 * __restrict__ Qa, const u32x4 * __restrict__ H, float * out, int nblk, unsigned span_mask)
{
    __shared__ __attribute__((aligned(16))) u32x4 rows_s[2][NS * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    half8 a[NS];
#pragma unroll
    for (int s = 0; s < NS; s++)
        a[s] = __builtin_bit_cast(half8, Qa[((size_t)(blockIdx.x * 8 + wave) * NS + s) * 64 + lane]);
    f32x16 total;
    for (int r = 0; r < 16; r++)
        total[r] = 0.f;
    const u32x4 * src = H + (size_t)((blockIdx.x * 37u) & span_mask) * NS * 64;
#pragma unroll
    for (int i = 0; i < 6; i++)
        rows_s[0][i * 512 + tid] = src[i * 512 + tid];
    __syncthreads();
    for (int b = 0; b < nblk; b++)
    {
        const int cur = b & 1;
        const u32x4 * nx = H + (size_t)((blockIdx.x * 37u + b + 1) & span_mask) * NS * 64;
        u32x4 st[6];
#pragma unroll
        for (int i = 0; i < 6; i++)
            st[i] = nx[i * 512 + tid];
        f32x16 acc0, acc1; // two accumulators (even / odd steps): no back-to-back dependence
#pragma unroll
        for (int r = 0; r < 16; r++)
            acc0[r] = acc1[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; s += 2)
        {
            const half8 b0 = __builtin_bit_cast(half8, rows_s[cur][s * 64 + lane]);
            const half8 b1 = __builtin_bit_cast(half8, rows_s[cur][(s + 1) * 64 + lane]);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s + 1], b1, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; r++)
            total[r] += acc0[r] + acc1[r] < 0.5f ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < 6; i++)
            rows_s[cur ^ 1][i * 512 + tid] = st[i];
        __syncthreads();
    }
    float s = 0.f;
    for (int r = 0; r < 16; r++)
        s += total[r];
    out[blockIdx.x * 512 + tid] = s;
}

int main()
{
    const size_t tbl_bytes = (size_t)256 << 20;
    u32x4 *Qa, *H;
    float * out;
    hipMalloc(&Qa, (size_t)256 * 8 * NS * 64 * 16);
    hipMalloc(&H, tbl_bytes);
    hipMalloc(&out, 256 * 512 * 4);
    std::vector<_Float16> r((size_t)16 << 20);
    srand(3);
    for (auto & v : r)
        v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    for (size_t off = 0; off < tbl_bytes; off += r.size() * 2)
        hipMemcpy((char *)H + off, r.data(), r.size() * 2, hipMemcpyHostToDevice);
    for (size_t off = 0; off < (size_t)256 * 8 * NS * 64 * 16; off += r.size() * 2)
        hipMemcpy((char *)Qa + off, r.data(), std::min(r.size() * 2, (size_t)256 * 8 * NS * 64 * 16 - off), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int nblk = 400;
    for (unsigned mask : {63u, 4095u})
    {
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, Qa, H, out, nblk, mask);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 5; i++)
            hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, Qa, H, out, nblk, mask);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double fl = 5.0 * 256 * 4 * (double)nblk * NS * 2 * 32768.0;
        printf("A in registers, rows through LDS, window of %u blocks: %.3f ms per launch, %.0f TF/s (%.3f of 2500)\n", mask + 1, ms / 5, fl / (ms * 1e-3) / 1e12, fl / (ms * 1e-3) / 2.5e15);
    }
    for (unsigned mask : {63u, 4095u})
    {
        hipLaunchKernelGGL(k8, dim3(256), dim3(512), 0, 0, Qa, H, out, nblk, mask);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 5; i++)
            hipLaunchKernelGGL(k8, dim3(256), dim3(512), 0, 0, Qa, H, out, nblk, mask);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double fl = 5.0 * 256 * 8 * (double)nblk * NS * 32768.0;
        printf("8 waves, 32 queries per wave in registers, window of %u blocks: %.3f ms per launch, %.0f TF/s (%.3f of 2500)\n", mask + 1, ms / 5, fl / (ms * 1e-3) / 1e12, fl / (ms * 1e-3) / 2.5e15);
    }
    return 0;
}
