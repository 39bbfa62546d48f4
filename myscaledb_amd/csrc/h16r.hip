// h16r.hip -- instantiations and launch of the register-tile list scan (h16r_scan_kernels.hpp); its own translation unit:
// 12 large kernels (metric x parts x chunks per part) compile beside msvs_capi.hip instead of inside it.
#include <hip/hip_runtime.h>

#include <mutex>

#include "h16r_scan_kernels.hpp"

namespace msvs
{

template <int METRIC, int KS, int CPPT>
static void h16r_launch(const H16Params & a, uint32_t grid, hipStream_t stream)
{
    // more than 64 KiB of dynamic LDS needs the attribute raised once per kernel
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&h16r_scan_kernel<METRIC, KS, CPPT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL((h16r_scan_kernel<METRIC, KS, CPPT>), dim3(grid), dim3(512), h16r_lds_bytes(KS, CPPT), stream, a);
}

template <int METRIC, int KS>
static void h16r_by_cpp(uint32_t cpp, const H16Params & a, uint32_t grid, hipStream_t stream)
{
    switch (cpp)
    {
        case 6: h16r_launch<METRIC, KS, 6>(a, grid, stream); break;
        case 8: h16r_launch<METRIC, KS, 8>(a, grid, stream); break;
        default: h16r_launch<METRIC, KS, 12>(a, grid, stream); break;
    }
}

void h16r_dispatch(int metric, uint32_t ks, const H16Params & a, uint32_t grid, hipStream_t stream)
{
    const uint32_t cpp = a.nch / ks;
    if (metric == M_IP)
        ks == 1 ? h16r_by_cpp<M_IP, 1>(cpp, a, grid, stream) : h16r_by_cpp<M_IP, 2>(cpp, a, grid, stream);
    else
        ks == 1 ? h16r_by_cpp<M_L2, 1>(cpp, a, grid, stream) : h16r_by_cpp<M_L2, 2>(cpp, a, grid, stream);
}

}
