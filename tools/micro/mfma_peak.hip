// Ceiling of v_mfma_f32_32x32x16_f16 on this chip under its power budget: NACC independent accumulators per wavefront, operands from
// registers only (random fp16 values), WPS wavefronts per SIMD.  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/micro/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(512) void k(const half8 * in, float * out, int iters)
{
    __shared__ half8 tile[LDS ? 4096 : 1];
    const int lane = threadIdx.x & 63;
    half8 a[3], b[2];
    for (int i = 0; i < 3; i++)
        a[i] = in[(threadIdx.x + 64 * i) & 1023];
    for (int i = 0; i < 2; i++)
        b[i] = in[(threadIdx.x * 7 + 64 * i) & 1023];
    if (LDS)
    {
        for (int i = threadIdx.x; i < 4096; i += 512)
            tile[i] = in[i & 1023];
        __syncthreads();
    }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++)
        for (int r = 0; r < 16; r++)
            acc[i][r] = 0.f;
    for (int it = 0; it < iters; it++)
    {
        if (LDS)
        {
#pragma unroll
            for (int i = 0; i < 3; i++)
                a[i] = tile[(lane + 64 * i + 192 * (it & 15)) & 4095];
        }
#pragma unroll
        for (int i = 0; i < NACC; i++)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i % 3], b[i / 3 % 2], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; i++)
        for (int r = 0; r < 16; r++)
            s += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(const char * name, const half8 * d_in, float * d_out, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(512), 0, 0, d_in, d_out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++)
        hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(512), 0, 0, d_in, d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = 5.0 * blocks * 8.0 * iters * NACC * 32768.0;
    printf("%-34s blocks %4d: %.3f ms per launch, %.0f TF/s (%.3f of 2500)\n", name, blocks, ms / 5, fl / (ms * 1e-3) / 1e12, fl / (ms * 1e-3) / 2.5e15);
}


// The exhaustive pass's loop in miniature: per chunk 24 MFMAs = 4 steps x (3 A fragments from LDS x 2 row blocks), the B operands
// arriving through a 3-slot register ring of 2 x 4 global_load_dwordx4 per chunk from a `span_blocks`-block window of a shadow table.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int LOADS>
__global__ __launch_bounds__(512) void kring(const half8 * in, const u32x4 * tbl, float * out, int chunks, unsigned span_mask)
{
    __shared__ half8 tile[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 512)
        tile[i] = in[i & 1023];
    __syncthreads();
    f32x16 acc[6];
    for (int i = 0; i < 6; i++)
        for (int r = 0; r < 16; r++)
            acc[i][r] = 0.f;
    u32x4 ring[2][3][4];
    const u32x4 * base = tbl + lane;
    unsigned pos = (blockIdx.x * 8 + wave) * 2 * 12; // chunk index of this wavefront's first block pair (12 chunks per block)
    auto load = [&](int slot, unsigned c) {
#pragma unroll
        for (int rb = 0; rb < 2; rb++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (LOADS)
                    ring[rb][slot][j] = base[(size_t)(((c + rb * 12) & span_mask) * 4 + j) * 64];
    };
    if (!LOADS)
        for (int rb = 0; rb < 2; rb++)
            for (int sl = 0; sl < 3; sl++)
                for (int j = 0; j < 4; j++)
                    ring[rb][sl][j] = base[(rb * 12 + sl * 4 + j) * 64];
    load(0, pos);
    load(1, pos + 1);
    __builtin_amdgcn_sched_barrier(0);
    for (int c0 = 0; c0 < chunks; c0 += 3)
    {
#pragma unroll
        for (int u = 0; u < 3; u++)
        {
            load((u + 2) % 3, pos + c0 + u + 2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                half8 a[3];
#pragma unroll
                for (int cb = 0; cb < 3; cb++)
                    a[cb] = tile[(lane + 64 * cb + 192 * ((c0 + u + j) & 15)) & 4095];
#pragma unroll
                for (int rb = 0; rb < 2; rb++)
#pragma unroll
                    for (int cb = 0; cb < 3; cb++)
                        acc[rb * 3 + cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cb], __builtin_bit_cast(half8, ring[rb][u][j]), acc[rb * 3 + cb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 6; i++)
        for (int r = 0; r < 16; r++)
            s += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int LOADS>
void run_ring(const char * name, const half8 * d_in, const u32x4 * tbl, float * d_out, int chunks, unsigned span_mask)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((kring<LOADS>), dim3(256), dim3(512), 0, 0, d_in, tbl, d_out, chunks, span_mask);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++)
        hipLaunchKernelGGL((kring<LOADS>), dim3(256), dim3(512), 0, 0, d_in, tbl, d_out, chunks, span_mask);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = 5.0 * 256 * 8.0 * chunks * 24 * 32768.0;
    printf("%-52s: %.3f ms per launch, %.0f TF/s (%.3f of 2500)\n", name, ms / 5, fl / (ms * 1e-3) / 1e12, fl / (ms * 1e-3) / 2.5e15);
}

int main()
{
    std::vector<_Float16> h(8192);
    srand(1);
    for (auto & v : h)
        v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    half8 * d_in;
    float * d_out;
    hipMalloc(&d_in, 8192 * 2);
    hipMalloc(&d_out, 1024 * 512 * 4);
    hipMemcpy(d_in, h.data(), 8192 * 2, hipMemcpyHostToDevice);
    run<6, false>("6 acc, registers only", d_in, d_out, 256, 40000);
    run<6, false>("6 acc, registers only", d_in, d_out, 512, 20000);
    run<6, true>("6 acc, 3 ds_read_b128 per 6 MFMA", d_in, d_out, 256, 40000);
    run<12, false>("12 acc, registers only", d_in, d_out, 256, 20000);
    run<6, false>("6 acc, registers only (again)", d_in, d_out, 256, 40000);
    u32x4 * tbl;
    const size_t tbl_bytes = (size_t)1 << 30; // 1 GiB of "shadow": 4 KiB per chunk
    hipMalloc(&tbl, tbl_bytes);
    {
        // random fp16 rows (|x| < 1) over the first 64 MB (the windows below stay inside it except the streaming case)
        std::vector<_Float16> r((size_t)32 << 20);
        for (auto & v : r)
            v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
        for (size_t off = 0; off < tbl_bytes; off += r.size() * 2)
            hipMemcpy((char *)tbl + off, r.data(), r.size() * 2, hipMemcpyHostToDevice);
    }
    run_ring<0>("ring loop, B from registers loaded once", d_in, tbl, d_out, 6000, 0);
    run_ring<1>("ring loop, 8 x dwordx4 per 24 MFMA, 96 KB window", d_in, tbl, d_out, 6000, 23);
    run_ring<1>("ring loop, 8 x dwordx4 per 24 MFMA, 3 MB window", d_in, tbl, d_out, 6000, 767);
    run_ring<1>("ring loop, 8 x dwordx4 per 24 MFMA, 1 GB stream", d_in, tbl, d_out, 6000, 262143);
    return 0;
}
