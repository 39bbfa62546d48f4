// What a small stream-ordered copy / fill costs between two kernels: hipMemcpyAsync (D2D, H2D from pinned), hipMemsetAsync, hipEventRecord
// against a copy KERNEL.  hipcc --offload-arch=gfx950 -O3 -o tools/micro/copy_gap tools/micro/copy_gap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void work(float * p) { p[threadIdx.x] += 1.f; }
__global__ void copyk(uint4 * d, const uint4 * s, size_t n) { for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i]; }
int main()
{
    const size_t bytes = 64 << 10;
    float * w; unsigned char *a, *b, *h;
    hipMalloc(&w, 4096); hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipHostMalloc(&h, bytes, hipHostMallocDefault);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1, ex; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreateWithFlags(&ex, hipEventDisableTiming);
    const int N = 200;
    for (int mode = 0; mode < 7; mode++)
    {
        for (int rep = 0; rep < 2; rep++)
        {
            hipEventRecord(e0, s);
            for (int i = 0; i < N; i++)
            {
                hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s, w);
                switch (mode)
                {
                    case 1: hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, s); break;
                    case 2: hipMemcpyAsync(b, h, bytes, hipMemcpyHostToDevice, s); break;
                    case 3: hipMemsetAsync(b, 0, bytes, s); break;
                    case 4: hipEventRecord(ex, s); break;
                    case 5: hipLaunchKernelGGL(copyk, dim3(16), dim3(256), 0, s, (uint4 *)b, (const uint4 *)a, bytes / 16); break;
                    case 6: hipLaunchKernelGGL(copyk, dim3(16), dim3(256), 0, s, (uint4 *)b, (const uint4 *)h, bytes / 16); break;
                    default: break;
                }
                hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s, w);
            }
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const char * names[] = {"two kernels", "+ hipMemcpyAsync D2D 64 KB", "+ hipMemcpyAsync H2D (pinned) 64 KB", "+ hipMemsetAsync 64 KB", "+ hipEventRecord", "+ copy kernel D2D 64 KB", "+ copy kernel pinned -> device 64 KB"};
        printf("%-40s %.2f us per iteration\n", names[mode], ms / N * 1e3);
    }
    return 0;
}
