// binary.hip -- seam A2 for BinaryVector: msvs_knn_bin (include/msvs.h), see bin_kernels.hpp.
#include <algorithm>

#include "bin_kernels.hpp"
#include "device_ops.hpp"

namespace msvs
{

template <int METRIC, int G>
static void bin_dispatch_r(const BinParams & a, hipStream_t stream)
{
    const dim3 grid(a.n_blocks, a.nq);
    const size_t lds = (size_t)a.ld16 * 16 + (size_t)5 * a.k * 8;
    if (a.k <= 64)
        hipLaunchKernelGGL((bin_scan_kernel<METRIC, G, 1>), grid, dim3(BLOCK), lds, stream, a);
    else
        hipLaunchKernelGGL((bin_scan_kernel<METRIC, G, 4>), grid, dim3(BLOCK), lds, stream, a);
}

template <int METRIC>
static void bin_dispatch_g(uint32_t g, const BinParams & a, hipStream_t stream)
{
    switch (g)
    {
        case 1:
            bin_dispatch_r<METRIC, 1>(a, stream);
            break;
        case 2:
            bin_dispatch_r<METRIC, 2>(a, stream);
            break;
        case 4:
            bin_dispatch_r<METRIC, 4>(a, stream);
            break;
        case 8:
            bin_dispatch_r<METRIC, 8>(a, stream);
            break;
        default:
            bin_dispatch_r<METRIC, 16>(a, stream);
            break;
    }
}

}

using namespace msvs;

extern "C" int msvs_knn_bin(const uint8_t * x, const uint8_t * y, size_t nbytes, size_t k, size_t nx, size_t ny, int metric,
                            const uint64_t * alive_bits, int64_t * ids, float * dis)
{
    return guarded([&] {
        if (metric != MSVS_METRIC_HAMMING && metric != MSVS_METRIC_JACCARD)
            fail(MSVS_ERR_NOT_IMPLEMENTED, "Metric not implemented in brute force search for Binary Vector");
        if (nx == 0 || k == 0)
            return;
        if (!x || !ids || !dis || (ny && !y) || nbytes == 0)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null buffer or zero dimension");
        if (k > MSVS_MAX_K)
            fail(MSVS_ERR_UNSUPPORTED_K, "k = %zu exceeds the device top-k limit %d", k, MSVS_MAX_K);
        if (ny > 0xfffffff0ull)
            fail(MSVS_ERR_ID_RANGE, "ny exceeds the u32 id range");
        hipStream_t stream = thread_stream();
        const uint32_t ld16 = (uint32_t)ceil_div(nbytes, (size_t)16);
        const size_t ldb = (size_t)ld16 * 16;
        if (ldb + 5 * k * 8 > SCAN_LDS_BUDGET)
            fail(MSVS_ERR_INVALID_ARGUMENT, "binary vectors of %zu bytes are too long for the LDS query stage", nbytes);
        uint32_t g = 1; // lanes per row: one 16-byte word each per step
        while (g < 16 && g * 2 <= ld16)
            g *= 2;
        // row ranges: ~2048 blocks over the chip, at least one wavefront step each
        const size_t rows_step = 4 * (64 / g);
        const size_t want = std::max<size_t>(1, 2048 / nx);
        const size_t nb = std::max<size_t>(1, std::min(want, ceil_div(std::max<size_t>(ny, 1), rows_step)));
        const uint32_t rpb = (uint32_t)round_up(ceil_div(std::max<size_t>(ny, 1), nb), rows_step);
        const uint32_t n_blocks = (uint32_t)ceil_div(std::max<size_t>(ny, 1), (size_t)rpb);
        const size_t words = alive_bits ? std::max<size_t>(1, ceil_div(ny, (size_t)64)) : 0;
        Scratch & scr = scratch_for(stream);
        scr.reserve((nx + std::max<size_t>(ny, 1)) * ldb + nx * (size_t)n_blocks * k * 8 + nx * k * 12 + words * 8 + 8192,
                    stream);
        unsigned char * dq = scr.take<unsigned char>(nx * ldb);
        unsigned char * dy = scr.take<unsigned char>(std::max<size_t>(ny, 1) * ldb);
        uint64_t * partial = scr.take<uint64_t>(nx * (size_t)n_blocks * k);
        int64_t * d_ids = scr.take<int64_t>(nx * k);
        float * d_dis = scr.take<float>(nx * k);
        uint64_t * d_alive = words ? scr.take<uint64_t>(words) : nullptr;
        if (ldb != nbytes)
        {
            MSVS_HIP(hipMemsetAsync(dq, 0, nx * ldb, stream));
            MSVS_HIP(hipMemsetAsync(dy, 0, std::max<size_t>(ny, 1) * ldb, stream));
        }
        MSVS_HIP(hipMemcpy2DAsync(dq, ldb, x, nbytes, nbytes, nx, hipMemcpyHostToDevice, stream));
        if (ny)
            MSVS_HIP(hipMemcpy2DAsync(dy, ldb, y, nbytes, nbytes, ny, hipMemcpyHostToDevice, stream));
        if (words)
            MSVS_HIP(hipMemcpyAsync(d_alive, alive_bits, words * 8, hipMemcpyHostToDevice, stream));
        BinParams a{};
        a.Y = reinterpret_cast<const uint4 *>(dy);
        a.Q = reinterpret_cast<const uint4 *>(dq);
        a.alive = d_alive;
        a.nbits = (uint32_t)ny;
        a.ld16 = ld16;
        a.n_rows = (uint32_t)ny;
        a.rows_per_block = rpb;
        a.n_blocks = n_blocks;
        a.k = (uint32_t)k;
        a.nq = (uint32_t)nx;
        a.partial = partial;
        {
            ProfileScope prof("bin_scan", stream);
            if (metric == MSVS_METRIC_HAMMING)
                bin_dispatch_g<B_HAMMING>(g, a, stream);
            else
                bin_dispatch_g<B_JACCARD>(g, a, stream);
            MSVS_HIP(hipGetLastError());
        }
        MergeParams m{};
        m.partial = partial;
        m.n_lists = n_blocks;
        m.k = (uint32_t)k;
        m.out_ids = d_ids;
        m.out_dis = d_dis;
        launch_merge(M_L2, m, (uint32_t)nx, stream);
        MSVS_HIP(hipMemcpyAsync(ids, d_ids, nx * k * 8, hipMemcpyDeviceToHost, stream));
        MSVS_HIP(hipMemcpyAsync(dis, d_dis, nx * k * 4, hipMemcpyDeviceToHost, stream));
        MSVS_HIP(hipStreamSynchronize(stream));
    });
}
