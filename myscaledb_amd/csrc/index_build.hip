// index_build.hip -- an index object comes into being (include/msvs.h: msvs_index_create / set_centroids / train / add / build,
// the size queries): k-means on the device, assignment of the rows to their lists, the list-major final storage, the row norms
// and the fp16 shadows the batched searches read (h16_scan_kernels.hpp).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#include "h16_scan_kernels.hpp"
#include "index_internal.hpp"
#include "ivf_build_kernels.hpp"

using namespace msvs;

/// The fp16 shadow of an IVF index (see h16_scan_kernels.hpp); called once the final storage and the norms are in place.
/// max |x' - x| / |x| over the rows of a table as the shadow stores them (h16_rho_kernel); < 0 when it cannot be measured.
static float measure_shadow_rho(const float * rows, size_t n, uint32_t ld, float scale, float inv_scale, hipStream_t stream)
{
    if (n == 0)
        return 0.f;
    DevBuf<uint32_t> rb(1);
    MSVS_HIP(hipMemsetAsync(rb.p, 0, 4, stream));
    hipLaunchKernelGGL(h16_rho_kernel, dim3((unsigned)ceil_div(n, (size_t)4)), dim3(256), 0, stream, rows, n, ld, scale, inv_scale, rb.p);
    MSVS_HIP(hipGetLastError());
    uint32_t bits = 0;
    MSVS_HIP(hipMemcpyAsync(&bits, rb.p, 4, hipMemcpyDeviceToHost, stream));
    MSVS_HIP(hipStreamSynchronize(stream));
    float rho;
    memcpy(&rho, &bits, 4);
    return rho >= 0.f && rho < 1.f ? rho : -1.f;
}

static void index_build_shadow(msvs_index & ix, hipStream_t stream)
{
    ix.shadow_ready = false;
    ix.h_rho = ix.c_rho = -1.f;
    if (ix.type == MSVS_INDEX_FLAT)
    {
        // a FLAT index is ONE list of n rows in id order: ceil(n / 32) blocks in the same operand layout (round 4): batches scan
        // it through h16_flat_kernel instead of converting the f32 rows to split bf16 on the fly
        if (!ix.want_shadow || ix.n == 0 || !(ix.xnorm_max < 1e30f) || ix.n > 0xfffffff0ull)
            return;
        DevBuf<uint32_t> mx(1);
        MSVS_HIP(hipMemsetAsync(mx.p, 0, 4, stream));
        const size_t n4 = ix.n * (size_t)(ix.ld / 4);
        hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)std::min<size_t>(ceil_div(n4, (size_t)256), 4096)), dim3(256), 0, stream,
                           reinterpret_cast<const float4 *>(ix.vecs.p), n4, mx.p);
        uint32_t bits = 0;
        MSVS_HIP(hipMemcpyAsync(&bits, mx.p, 4, hipMemcpyDeviceToHost, stream));
        MSVS_HIP(hipStreamSynchronize(stream));
        float maxabs;
        memcpy(&maxabs, &bits, 4);
        if (!(maxabs < 3.0e38f))
            return;
        int ex = 0;
        if (maxabs > 0.f)
            (void)frexpf(maxabs, &ex);
        const int sh = 14 - ex;
        if (sh > 100 || sh < -100)
            return;
        ix.h_scale = ldexpf(1.f, sh);
        ix.h_inv_scale = ldexpf(1.f, -sh);
        ix.h_nch = (uint32_t)ceil_div(ix.dim, (size_t)H_CHUNK);
        ix.h_nks = 4 * ix.h_nch;
        const size_t G = ceil_div(ix.n, (size_t)H_ROWS);
        const std::vector<uint32_t> one_hoff = {0u, (uint32_t)G};
        const std::vector<int64_t> one_off = {0, (int64_t)ix.n};
        DevBuf<uint32_t> d_one_hoff(2), d_blk(G);
        DevBuf<int64_t> d_one_off(2);
        MSVS_HIP(hipMemsetAsync(d_blk.p, 0, G * 4, stream)); // every block belongs to list 0
        MSVS_HIP(hipMemcpyAsync(d_one_hoff.p, one_hoff.data(), 8, hipMemcpyHostToDevice, stream));
        MSVS_HIP(hipMemcpyAsync(d_one_off.p, one_off.data(), 16, hipMemcpyHostToDevice, stream));
        const size_t npieces = G * (size_t)ix.h_nks * 64;
        ix.shadow.alloc(npieces);
        const size_t per_launch = (size_t)1 << 30;
        for (size_t p0 = 0; p0 < npieces; p0 += per_launch)
        {
            const size_t m = std::min(per_launch, npieces - p0);
            hipLaunchKernelGGL(h16_build_kernel, dim3((unsigned)ceil_div(m, (size_t)256)), dim3(256), 0, stream, ix.vecs.p, ix.ld,
                               d_one_off.p, d_blk.p, d_one_hoff.p, ix.h_nks, ix.h_scale, ix.shadow.p, p0, m);
        }
        MSVS_HIP(hipGetLastError());
        MSVS_HIP(hipStreamSynchronize(stream));
        ix.h_rho = measure_shadow_rho(ix.vecs.p, ix.n, ix.ld, ix.h_scale, ix.h_inv_scale, stream);
        ix.shadow_ready = true;
        return;
    }
    if (ix.type != MSVS_INDEX_IVFFLAT || !ix.want_shadow || ix.n == 0 || ix.nlist == 0 || !(ix.xnorm_max < 1e30f)
        || ix.n > 0xfffffff0ull)
        return;
    DevBuf<uint32_t> mx(1);
    MSVS_HIP(hipMemsetAsync(mx.p, 0, 4, stream));
    const size_t n4 = ix.n * (size_t)(ix.ld / 4);
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)std::min<size_t>(ceil_div(n4, (size_t)256), 4096)), dim3(256), 0,
                       stream, reinterpret_cast<const float4 *>(ix.vecs.p), n4, mx.p);
    MSVS_HIP(hipGetLastError());
    uint32_t bits = 0;
    MSVS_HIP(hipMemcpyAsync(&bits, mx.p, 4, hipMemcpyDeviceToHost, stream));
    MSVS_HIP(hipStreamSynchronize(stream));
    float maxabs;
    memcpy(&maxabs, &bits, 4);
    if (!(maxabs < 3.0e38f))
        return;
    // The centroid table shares the rows' scale (one set of query images serves the coarse pass and the list scan).  The centroids of
    // an index's OWN rows never exceed them -- but a shard of a sharded index holds every centroid and only its own lists' rows: a
    // component of somebody else's centroid beyond this shard's largest row cost the shard its centroid shadow, and with it the
    // approximate centroid distances the routed search's pre-pruning reads (W = 8 over the bench data: 3 of 8 ranks pruned nothing,
    // 4.5 instead of 1.0 ranks visited per query).  So the scale covers the centroids too, as long as that costs the rows at most two
    // bits (centroids far beyond every row -- user-supplied ones -- still go without a shadow).
    float cmax_tab = 0.f;
    if (ix.type == MSVS_INDEX_IVFFLAT && ix.nlist && ix.centroids.p && ix.cnorm_max < 1e30f)
    {
        MSVS_HIP(hipMemsetAsync(mx.p, 0, 4, stream));
        const size_t c4 = ix.nlist * (size_t)(ix.ld / 4);
        hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)std::min<size_t>(ceil_div(c4, (size_t)256), 4096)), dim3(256), 0, stream,
                           reinterpret_cast<const float4 *>(ix.centroids.p), c4, mx.p);
        MSVS_HIP(hipGetLastError());
        MSVS_HIP(hipMemcpyAsync(&bits, mx.p, 4, hipMemcpyDeviceToHost, stream));
        MSVS_HIP(hipStreamSynchronize(stream));
        memcpy(&cmax_tab, &bits, 4);
        if (cmax_tab > maxabs && cmax_tab <= 4.f * maxabs)
            maxabs = cmax_tab;
    }
    int ex = 0;
    if (maxabs > 0.f)
        (void)frexpf(maxabs, &ex); // maxabs < 2^ex  =>  |x| * 2^(14 - ex) < 2^14: no fp16 overflow
    const int sh = 14 - ex;
    if (sh > 100 || sh < -100)
        return; // the error model assumes the fp16 subnormal quantum is <= 2^-38 max|x|
    ix.h_scale = ldexpf(1.f, sh);
    ix.h_inv_scale = ldexpf(1.f, -sh);
    ix.h_nch = (uint32_t)ceil_div(ix.dim, (size_t)H_CHUNK);
    ix.h_nks = 4 * ix.h_nch; // whole chunks: the scan's inner loop has no tail
    std::vector<uint32_t> hoff(ix.nlist + 1, 0);
    std::vector<int64_t> mid(ix.nlist);
    for (size_t l = 0; l < ix.nlist; l++)
    {
        const size_t len = (size_t)(ix.h_list_off[l + 1] - ix.h_list_off[l]);
        const size_t nb = hoff[l] + ceil_div(len, (size_t)H_ROWS);
        if (nb > 0xfffffff0ull)
            return;
        hoff[l + 1] = (uint32_t)nb;
        mid[l] = std::min<int64_t>(ix.h_list_off[l] + H_ROWS, ix.h_list_off[l + 1]);
    }
    const size_t nblocks = hoff[ix.nlist];
    std::vector<uint32_t> blk_list(nblocks);
    for (size_t l = 0; l < ix.nlist; l++)
        std::fill(blk_list.begin() + hoff[l], blk_list.begin() + hoff[l + 1], (uint32_t)l);
    DevBuf<uint32_t> d_blk(std::max<size_t>(nblocks, 1));
    ix.hoff.alloc(ix.nlist + 1);
    ix.list_mid32.alloc(ix.nlist);
    const size_t npieces = nblocks * (size_t)ix.h_nks * 64;
    ix.shadow.alloc(npieces + 32768); // (+ 512 KiB of zeros past the last list)
    MSVS_HIP(hipMemsetAsync(ix.shadow.p + npieces, 0, 32768 * sizeof(uint4), stream));
    MSVS_HIP(hipMemcpyAsync(d_blk.p, blk_list.data(), nblocks * 4, hipMemcpyHostToDevice, stream));
    MSVS_HIP(hipMemcpyAsync(ix.hoff.p, hoff.data(), (ix.nlist + 1) * 4, hipMemcpyHostToDevice, stream));
    MSVS_HIP(hipMemcpyAsync(ix.list_mid32.p, mid.data(), ix.nlist * 8, hipMemcpyHostToDevice, stream));
    const size_t per_launch = (size_t)1 << 30; // pieces per launch (grid dimension limit)
    for (size_t p0 = 0; p0 < npieces; p0 += per_launch)
    {
        const size_t m = std::min(per_launch, npieces - p0);
        // p0 is a multiple of 2^30 pieces; the kernel indexes from the start of the shadow, so shift the base
        hipLaunchKernelGGL(h16_build_kernel, dim3((unsigned)ceil_div(m, (size_t)256)), dim3(256), 0, stream, ix.vecs.p,
                           ix.ld, ix.list_off.p, d_blk.p, ix.hoff.p, ix.h_nks, ix.h_scale, ix.shadow.p, p0, m);
    }
    MSVS_HIP(hipGetLastError());
    MSVS_HIP(hipStreamSynchronize(stream));
    ix.h_rho = measure_shadow_rho(ix.vecs.p, ix.n, ix.ld, ix.h_scale, ix.h_inv_scale, stream);
    ix.shadow_ready = true;
    // the centroid table in the same form (one list of nlist rows = G blocks), if it fits the rows' scale
    ix.c_shadow_ready = false;
    {
        MSVS_HIP(hipMemsetAsync(mx.p, 0, 4, stream));
        const size_t c4 = ix.nlist * (size_t)(ix.ld / 4);
        hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)std::min<size_t>(ceil_div(c4, (size_t)256), 4096)), dim3(256), 0, stream,
                           reinterpret_cast<const float4 *>(ix.centroids.p), c4, mx.p);
        MSVS_HIP(hipMemcpyAsync(&bits, mx.p, 4, hipMemcpyDeviceToHost, stream));
        MSVS_HIP(hipStreamSynchronize(stream));
        float cmax;
        memcpy(&cmax, &bits, 4);
        if (!(cmax <= maxabs) || !(ix.cnorm_max < 1e30f)) // user-supplied centroids may be larger than any row: no shadow then
            return;
        const size_t G = ceil_div(ix.nlist, (size_t)H_ROWS);
        std::vector<uint32_t> c_hoff(G + 1), one_hoff = {0u, (uint32_t)G};
        std::vector<int64_t> c_off(G + 1), one_off = {0, (int64_t)ix.nlist};
        for (size_t g = 0; g <= G; g++)
        {
            c_hoff[g] = (uint32_t)g;
            c_off[g] = (int64_t)std::min(g * H_ROWS, ix.nlist);
        }
        DevBuf<uint32_t> d_one_hoff(2), d_cblk(G);
        DevBuf<int64_t> d_one_off(2);
        MSVS_HIP(hipMemsetAsync(d_cblk.p, 0, G * 4, stream)); // every block belongs to list 0
        MSVS_HIP(hipMemcpyAsync(d_one_hoff.p, one_hoff.data(), 8, hipMemcpyHostToDevice, stream));
        MSVS_HIP(hipMemcpyAsync(d_one_off.p, one_off.data(), 16, hipMemcpyHostToDevice, stream));
        ix.c_hoff.alloc(G + 1);
        ix.c_list_off.alloc(G + 1);
        MSVS_HIP(hipMemcpyAsync(ix.c_hoff.p, c_hoff.data(), (G + 1) * 4, hipMemcpyHostToDevice, stream));
        MSVS_HIP(hipMemcpyAsync(ix.c_list_off.p, c_off.data(), (G + 1) * 8, hipMemcpyHostToDevice, stream));
        const size_t cpieces = G * (size_t)ix.h_nks * 64;
        ix.c_shadow.alloc(cpieces);
        hipLaunchKernelGGL(h16_build_kernel, dim3((unsigned)ceil_div(cpieces, (size_t)256)), dim3(256), 0, stream, ix.centroids.p, ix.ld,
                           d_one_off.p, d_cblk.p, d_one_hoff.p, ix.h_nks, ix.h_scale, ix.c_shadow.p, (size_t)0, cpieces);
        MSVS_HIP(hipGetLastError());
        MSVS_HIP(hipStreamSynchronize(stream));
        ix.c_rho = measure_shadow_rho(ix.centroids.p, ix.nlist, ix.ld, ix.h_scale, ix.h_inv_scale, stream);
        ix.c_shadow_ready = true;
    }
}

namespace msvs
{
/// Row norms for the approximate pass and its error bound; called once the final storage is in place.
void index_finalize_norms(msvs_index & ix, hipStream_t stream)
{
    if (!ix.plan_fb.pairs)
    {
        MSVS_HIP(hipHostMalloc(reinterpret_cast<void **>(&ix.plan_fb.pairs), 64, hipHostMallocDefault));
        *ix.plan_fb.pairs = 0xFFFFFFFFu;
        ix.plan_fb.pairs[1] = ix.plan_fb.pairs[2] = 0;
    }
    ix.xnorm.alloc(std::max<size_t>(ix.n, 1));
    ix.xnorm_max = 0.f;
    DevBuf<uint32_t> mx(1);
    if (ix.type == MSVS_INDEX_IVFFLAT && ix.nlist)
    {
        std::vector<int64_t> mid(ix.nlist);
        for (size_t l = 0; l < ix.nlist; l++)
            mid[l] = std::min<int64_t>(ix.h_list_off[l] + BG_ROWS, ix.h_list_off[l + 1]);
        ix.list_mid.alloc(ix.nlist);
        MSVS_HIP(hipMemcpyAsync(ix.list_mid.p, mid.data(), ix.nlist * 8, hipMemcpyHostToDevice, stream));
        MSVS_HIP(hipStreamSynchronize(stream));
        ix.cnorm.alloc(ix.nlist);
        MSVS_HIP(hipMemsetAsync(mx.p, 0, 4, stream));
        launch_row_sqnorm(ix.centroids.p, ix.cnorm.p, ix.nlist, ix.ld / 4, mx.p, stream);
        uint32_t cb = 0;
        MSVS_HIP(hipMemcpyAsync(&cb, mx.p, 4, hipMemcpyDeviceToHost, stream));
        MSVS_HIP(hipStreamSynchronize(stream));
        memcpy(&ix.cnorm_max, &cb, 4);
    }
    if (ix.n == 0)
        return;
    MSVS_HIP(hipMemsetAsync(mx.p, 0, 4, stream));
    launch_row_sqnorm(ix.vecs.p, ix.xnorm.p, ix.n, ix.ld / 4, mx.p, stream);
    uint32_t bits = 0;
    MSVS_HIP(hipMemcpyAsync(&bits, mx.p, 4, hipMemcpyDeviceToHost, stream));
    MSVS_HIP(hipStreamSynchronize(stream));
    memcpy(&ix.xnorm_max, &bits, 4); // NaN / inf / huge values switch the candidate pass off (see plan_ivf)
    {
        const uint32_t inf_bits = 0x7f800000u;
        MSVS_HIP(hipMemcpyAsync(mx.p, &inf_bits, 4, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(min_f32_kernel, dim3((unsigned)std::min<size_t>(ceil_div(ix.n, (size_t)256), 1024)), dim3(256), 0, stream, ix.xnorm.p,
                           ix.n, mx.p);
        MSVS_HIP(hipGetLastError());
        MSVS_HIP(hipMemcpyAsync(&bits, mx.p, 4, hipMemcpyDeviceToHost, stream));
        MSVS_HIP(hipStreamSynchronize(stream));
        memcpy(&ix.xnorm_min, &bits, 4);
    }
    if (ix.type == MSVS_INDEX_IVFFLAT && ix.nlist)
    {
        // how far a list's rows lie from its centroid at most: lets a batched search drop (query, list) pairs that provably
        // cannot hold one of the query's k nearest rows (triangle inequality; h16_list_scan)
        ix.list_radius.alloc(ix.nlist);
        hipLaunchKernelGGL(list_radius_kernel, dim3((unsigned)ix.nlist), dim3(256), 0, stream, ix.vecs.p, ix.centroids.p, ix.list_off.p,
                           (uint32_t)ix.dim, ix.ld, ix.list_radius.p);
        MSVS_HIP(hipGetLastError());
    }
    index_build_shadow(ix, stream);
}
}


static void index_assign(const msvs_index & ix, const float * d_x, size_t n, int32_t * d_assign, hipStream_t stream)
{
    DevBuf<float> cnorm(ix.nlist);
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)ceil_div(ix.nlist, 256)), dim3(256), 0, stream,
                       ix.centroids.p, cnorm.p, (uint32_t)ix.nlist, (uint32_t)ix.dim, ix.ld);
    unsigned grid = (unsigned)ceil_div(n, AS_TN);
    if (ix.metric == MSVS_METRIC_L2)
        hipLaunchKernelGGL((assign_kernel<false>), dim3(grid), dim3(256), 0, stream, d_x, n, ix.centroids.p, cnorm.p,
                           (uint32_t)ix.nlist, (uint32_t)ix.dim, ix.ld, d_assign, (float *)nullptr);
    else
        hipLaunchKernelGGL((assign_kernel<true>), dim3(grid), dim3(256), 0, stream, d_x, n, ix.centroids.p, cnorm.p,
                           (uint32_t)ix.nlist, (uint32_t)ix.dim, ix.ld, d_assign, (float *)nullptr);
    MSVS_HIP(hipGetLastError());
    MSVS_HIP(hipStreamSynchronize(stream));
}


extern "C" int msvs_index_create(int index_type, int metric, size_t dim, const char * params, msvs_index_t ** out)
{
    return guarded([&] {
        if (!out)
            fail(MSVS_ERR_INVALID_ARGUMENT, "out is null");
        *out = nullptr;
        if (index_type != MSVS_INDEX_FLAT && index_type != MSVS_INDEX_IVFFLAT)
            fail(MSVS_ERR_NOT_IMPLEMENTED, "index type %d is not implemented", index_type);
        if (metric != MSVS_METRIC_L2 && metric != MSVS_METRIC_IP && metric != MSVS_METRIC_COSINE)
            fail(MSVS_ERR_NOT_IMPLEMENTED, "metric %d is not implemented for Float32 vectors", metric);
        if (dim == 0 || dim > 8192)
            fail(MSVS_ERR_INVALID_ARGUMENT, "dimension %zu out of range [1, 8192]", dim);
        auto p = parse_params(params);
        std::unique_ptr<msvs_index> ix(new msvs_index);
        ix->type = index_type;
        ix->metric = metric;
        ix->dim = dim;
        ix->ld = padded_dim(dim);
        MSVS_HIP(hipGetDevice(&ix->device));
        ix->ncentroids = (size_t)param_int(p, "ncentroids", 1024);
        ix->kmeans_iters = (int)param_int(p, "kmeans_iters", 10);
        if (p.count("kmeans_split_big"))
            ix->split_big = std::max(1.1, atof(p.at("kmeans_split_big").c_str()));
        if (p.count("kmeans_split_small"))
            ix->split_small = std::min(1.0, std::max(0.0, atof(p.at("kmeans_split_small").c_str())));
        ix->train_sample = (size_t)param_int(p, "train_sample", 0);
        ix->seed = (uint64_t)param_int(p, "seed", 1234);
        ix->shard_rank = (int)param_int(p, "shard_rank", 0);
        ix->shard_world = (int)param_int(p, "shard_world", 1);
        ix->want_shadow = (int)param_int(p, "shadow", 1);
        if (ix->ncentroids == 0 || ix->shard_world < 1 || ix->shard_rank < 0 || ix->shard_rank >= ix->shard_world)
            fail(MSVS_ERR_INVALID_ARGUMENT, "bad ncentroids / shard parameters");
        *out = ix.release();
    });
}

extern "C" void msvs_index_free(msvs_index_t * index)
{
    if (index)
        combiner_forget(index);
    delete index;
}

extern "C" int msvs_index_set_centroids(msvs_index_t * ix, const float * centroids, size_t nlist, int mem)
{
    return guarded([&] {
        DeviceGuard on_device(ix ? ix->device : -1);
        if (!ix || !centroids || nlist == 0)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null index/centroids");
        if (ix->type != MSVS_INDEX_IVFFLAT)
            return;
        if (ix->staged || ix->ready)
            fail(MSVS_ERR_INVALID_ARGUMENT, "centroids must be set before data is added");
        ix->nlist = nlist;
        ix->centroids.alloc(nlist * ix->ld);
        upload_rows(ix->centroids.p, centroids, nlist, (uint32_t)ix->dim, ix->ld, mem, nullptr);
        MSVS_HIP(hipStreamSynchronize(nullptr));
    });
}

extern "C" int msvs_index_train(msvs_index_t * ix, const float * x, size_t n, int mem)
{
    return guarded([&] {
        DeviceGuard on_device(ix ? ix->device : -1);
        if (!ix || (n && !x))
            fail(MSVS_ERR_INVALID_ARGUMENT, "null index/data");
        if (ix->type != MSVS_INDEX_IVFFLAT)
            return;
        if (ix->staged || ix->ready)
            fail(MSVS_ERR_INVALID_ARGUMENT, "train must precede add");
        if (n == 0)
            fail(MSVS_ERR_INVALID_ARGUMENT, "no training data");
        hipStream_t stream = nullptr;
        const uint32_t d = (uint32_t)ix->dim, ld = ix->ld;
        size_t nlist = std::min(ix->ncentroids, n);
        size_t ns = ix->train_sample ? ix->train_sample : nlist * 64;
        ns = std::min(ns, n);
        // deterministic sample: a seeded partial Fisher-Yates over row indices
        std::mt19937_64 rng(ix->seed);
        std::vector<uint32_t> perm(n);
        std::iota(perm.begin(), perm.end(), 0u);
        for (size_t i = 0; i < ns; i++)
        {
            size_t j = i + (size_t)(rng() % (n - i));
            std::swap(perm[i], perm[j]);
        }
        perm.resize(ns);
        // stage the sample on the device (rows padded to ld)
        DevBuf<float> xs(ns * ld);
        {
            DevBuf<uint32_t> d_idx(ns);
            MSVS_HIP(hipMemcpyAsync(d_idx.p, perm.data(), ns * 4, hipMemcpyHostToDevice, stream));
            if (mem == MSVS_MEM_DEVICE && ld == d)
            {
                size_t total = ns * (ld / 4);
                hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, stream,
                                   reinterpret_cast<const float4 *>(x), reinterpret_cast<float4 *>(xs.p), d_idx.p, ns,
                                   ld / 4);
                MSVS_HIP(hipGetLastError());
            }
            else
            {
                // host data (or odd dimension): gather on the host, then upload
                std::vector<float> hx;
                const float * src = x;
                if (mem == MSVS_MEM_DEVICE)
                {
                    hx.resize(n * d);
                    MSVS_HIP(hipMemcpy(hx.data(), x, n * d * 4, hipMemcpyDeviceToHost));
                    src = hx.data();
                }
                std::vector<float> g(ns * d);
                for (size_t i = 0; i < ns; i++)
                    memcpy(&g[i * d], src + (size_t)perm[i] * d, d * 4);
                upload_rows(xs.p, g.data(), ns, d, ld, MSVS_MEM_HOST, stream);
                MSVS_HIP(hipStreamSynchronize(stream));
            }
            MSVS_HIP(hipStreamSynchronize(stream));
        }
        if (ix->metric == MSVS_METRIC_COSINE)
            normalize_device_rows(xs.p, ns, d, ld, stream);
        // init: the first nlist sampled rows
        ix->nlist = nlist;
        ix->centroids.alloc(nlist * ld);
        MSVS_HIP(hipMemcpyAsync(ix->centroids.p, xs.p, nlist * ld * 4, hipMemcpyDeviceToDevice, stream));
        DevBuf<int32_t> d_assign(ns);
        DevBuf<int64_t> d_off(nlist + 1);
        DevBuf<uint32_t> d_members(ns);
        std::vector<int32_t> h_assign(ns);
        std::vector<int64_t> off(nlist + 1);
        std::vector<uint32_t> members(ns);
        for (int it = 0; it < ix->kmeans_iters; it++)
        {
            if (ix->is_cancelled && ix->is_cancelled(ix->cancel_ctx))
            {
                MSVS_HIP(hipStreamSynchronize(stream));
                ix->centroids.release(); // an index whose training was cut short is untrained again
                ix->nlist = 0;
                fail(MSVS_ERR_ABORTED, "Cancelled building vector index");
            }
            // Lloyd step: L2 assignment (IP/cosine indexes train on L2 too, like Faiss' default clustering)
            msvs_index tmp_view;
            (void)tmp_view;
            int saved = ix->metric;
            ix->metric = MSVS_METRIC_L2;
            index_assign(*ix, xs.p, ns, d_assign.p, stream);
            ix->metric = saved;
            MSVS_HIP(hipMemcpy(h_assign.data(), d_assign.p, ns * 4, hipMemcpyDeviceToHost));
            std::fill(off.begin(), off.end(), 0);
            for (size_t i = 0; i < ns; i++)
                off[h_assign[i] + 1]++;
            for (size_t j = 0; j < nlist; j++)
                off[j + 1] += off[j];
            std::vector<int64_t> cur(off.begin(), off.end() - 1);
            for (size_t i = 0; i < ns; i++)
                members[cur[h_assign[i]]++] = (uint32_t)i;
            MSVS_HIP(hipMemcpyAsync(d_off.p, off.data(), (nlist + 1) * 8, hipMemcpyHostToDevice, stream));
            MSVS_HIP(hipMemcpyAsync(d_members.p, members.data(), ns * 4, hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(centroid_update_kernel, dim3((unsigned)nlist), dim3(256), 0, stream, xs.p, d, ld,
                               d_off.p, d_members.p, ix->centroids.p);
            MSVS_HIP(hipGetLastError());
            MSVS_HIP(hipStreamSynchronize(stream));
            // Empty clusters are re-seeded the way Faiss' Clustering does (split_clusters; the library behind the reference's
            // IVF indexes bundles it: BruteForceSearch.h:17,32): an empty cluster takes the centroid of a cluster drawn with
            // probability ~ (size - 1), the two copies are pushed apart by a relative 1/1024 with alternating sign per dimension,
            // and the donor's members count as split in halves for the next draw.  Without it a cluster that loses its members
            // stays where it is for good (round 2: half of the lists <= 10 rows on iid data).
            // ... "empty" here includes the nearly empty: a cluster holding less than 1/16 of the average is a centroid that
            // fits a handful of sample points (on unstructured data it ends with a list of one or two rows); it is re-seeded
            // like an empty one, except in the last iteration (its members would be left without their mean)
            const bool last_it = it + 1 >= ix->kmeans_iters;
            const size_t tiny = last_it ? 0 : ns / nlist / 16;
            std::vector<char> give_up(nlist, 0); // clusters re-seeded although they have members
            size_t nempty = 0;
            for (size_t j = 0; j < nlist; j++)
                nempty += (size_t)(off[j + 1] - off[j]) <= tiny;
            // ... and the oversized: Lloyd's iteration cannot undo a seeding that put two centroids into one well-separated blob
            // and none into another (the orphan blobs merge into a neighbour's list: 12 blobs in one list on SURVEY 8d's
            // sigma-0.3 model).  While a cluster holds more than `split_big` x the average, the smallest cluster below
            // `split_small` x the average gives its centroid up to split it; its members fall to their next centroid (the twin
            // inside the same blob).  split_big = 1.7 (round 3: 2.5): a list that holds TWO blobs is 2 x the average, and round 4's
            // probe of the pruning (tools/prune_probe.py) found 141 such lists of 1024 on the sigma-0.3 model holding 93 % of the
            // (query, list) pairs the pruning could not drop -- their radius is half the distance between two blobs, and their
            // centroid, the mean of two, is closer to every query than a blob centre is.
            std::vector<std::pair<size_t, size_t>> forced; // (small cluster, the giant it splits)
            // (not in the last two iterations: a pair of twins needs a Lloyd step or two to part and settle)
            if (it + 2 < ix->kmeans_iters && ns > 4 * nlist)
            {
                const double avg = (double)ns / (double)nlist;
                std::vector<size_t> order(nlist);
                std::iota(order.begin(), order.end(), (size_t)0);
                std::sort(order.begin(), order.end(), [&](size_t x, size_t y) {
                    const int64_t sx = off[x + 1] - off[x], sy = off[y + 1] - off[y];
                    return sx != sy ? sx < sy : x < y;
                });
                size_t lo = 0, hi = nlist;
                while (lo + 1 < hi)
                {
                    const size_t small = order[lo], big = order[hi - 1];
                    const double ssz = (double)(off[small + 1] - off[small]), bsz = (double)(off[big + 1] - off[big]);
                    if (!(bsz > ix->split_big * avg && ssz < ix->split_small * avg))
                        break;
                    if (ssz > (double)tiny) // the tiny ones are re-seeded by the draw below anyway
                    {
                        forced.push_back({small, big});
                        give_up[small] = 1;
                        hi--;
                    }
                    lo++;
                }
            }
            if ((nempty || !forced.empty()) && ns > nlist)
            {
                std::vector<float> hc(nlist * ld);
                MSVS_HIP(hipMemcpy(hc.data(), ix->centroids.p, nlist * ld * 4, hipMemcpyDeviceToHost));
                std::vector<double> sz(nlist);
                for (size_t j = 0; j < nlist; j++)
                    sz[j] = (size_t)(off[j + 1] - off[j]) <= tiny ? 0.0 : (double)(off[j + 1] - off[j]);
                const float feps = 1.f / 1024.f;
                for (const auto & fs : forced)
                {
                    const size_t ci = fs.first, cj = fs.second;
                    for (uint32_t c = 0; c < d; c++)
                    {
                        const float v = hc[cj * ld + c];
                        hc[ci * ld + c] = v * (c % 2 == 0 ? 1 + feps : 1 - feps);
                        hc[cj * ld + c] = v * (c % 2 == 0 ? 1 - feps : 1 + feps);
                    }
                    sz[ci] = std::floor(sz[cj] / 2);
                    sz[cj] -= sz[ci];
                }
                std::mt19937_64 srng(ix->seed * 1315423911ull + (uint64_t)it);
                std::uniform_real_distribution<double> uni(0.0, 1.0);
                const double denom = (double)(ns - nlist);
                for (size_t ci = 0; ci < nlist; ci++)
                {
                    if (sz[ci] != 0)
                        continue;
                    size_t cj = 0;
                    for (size_t guard = 0; guard < 64 * nlist; guard++, cj = (cj + 1) % nlist)
                        if (uni(srng) < (sz[cj] - 1.0) / denom)
                            break;
                    if (sz[cj] < 2)
                        continue; // nothing left to split (more clusters than distinct points)
                    const float eps = 1.f / 1024.f;
                    for (uint32_t c = 0; c < d; c++)
                    {
                        const float v = hc[cj * ld + c];
                        hc[ci * ld + c] = v * (c % 2 == 0 ? 1 + eps : 1 - eps);
                        hc[cj * ld + c] = v * (c % 2 == 0 ? 1 - eps : 1 + eps);
                    }
                    sz[ci] = std::floor(sz[cj] / 2);
                    sz[cj] -= sz[ci];
                }
                MSVS_HIP(hipMemcpy(ix->centroids.p, hc.data(), nlist * ld * 4, hipMemcpyHostToDevice));
            }
            ix->train_empty_last = nempty;
        }
    });
}

/// The host's check_cancelled callback (VIWithDataPart.cpp:425-430), polled where a build can still stop cleanly.
static void poll_cancel(const msvs_index_t * ix)
{
    if (ix->is_cancelled && ix->is_cancelled(ix->cancel_ctx))
        fail(MSVS_ERR_ABORTED, "Cancelled building vector index");
}

extern "C" int msvs_index_set_cancel(msvs_index_t * ix, int (*is_cancelled)(void *), void * ctx)
{
    return guarded([&] {
        if (!ix)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null index");
        ix->is_cancelled = is_cancelled;
        ix->cancel_ctx = ctx;
    });
}

extern "C" int msvs_index_add(msvs_index_t * ix, const float * x, const int64_t * ids, size_t n, int mem)
{
    return guarded([&] {
        DeviceGuard on_device(ix ? ix->device : -1);
        if (!ix || (n && !x))
            fail(MSVS_ERR_INVALID_ARGUMENT, "null index/data");
        if (ix->ready)
            fail(MSVS_ERR_INVALID_ARGUMENT, "index already built");
        poll_cancel(ix);
        if (n == 0)
            return;
        if (ix->type == MSVS_INDEX_IVFFLAT && ix->nlist == 0)
            fail(MSVS_ERR_NOT_READY, "IVFFLAT index must be trained (or given centroids) before add");
        hipStream_t stream = nullptr;
        const uint32_t d = (uint32_t)ix->dim, ld = ix->ld;
        msvs_index::Chunk ch;
        ch.n = n;
        ch.x.alloc(n * ld);
        upload_rows(ch.x.p, x, n, d, ld, mem, stream);
        ch.ids.resize(n);
        if (ids)
        {
            if (mem == MSVS_MEM_DEVICE)
                MSVS_HIP(hipMemcpy(ch.ids.data(), ids, n * 8, hipMemcpyDeviceToHost));
            else
                memcpy(ch.ids.data(), ids, n * 8);
            // distinct as long as every chunk is strictly ascending and starts above everything before it (what a part's row
            // offsets look like); anything else may repeat a label
            for (size_t i = 0; i < n && !ix->ids_may_repeat; i++)
            {
                if (ch.ids[i] <= ix->last_id)
                    ix->ids_may_repeat = true;
                ix->last_id = ch.ids[i];
            }
        }
        else
        {
            for (size_t i = 0; i < n; i++)
                ch.ids[i] = (int64_t)(ix->staged + i);
            if ((int64_t)ix->staged <= ix->last_id)
                ix->ids_may_repeat = true;
            ix->last_id = (int64_t)(ix->staged + n - 1);
        }
        for (size_t i = 0; i < n; i++)
            if (ch.ids[i] < 0 || ch.ids[i] > 0xfffffff0ll)
                fail(MSVS_ERR_ID_RANGE, "id %lld does not fit the u32 label range", (long long)ch.ids[i]);
        if (ix->metric == MSVS_METRIC_COSINE)
            normalize_device_rows(ch.x.p, n, d, ld, stream);
        if (ix->type == MSVS_INDEX_IVFFLAT)
        {
            DevBuf<int32_t> d_assign(n);
            index_assign(*ix, ch.x.p, n, d_assign.p, stream);
            ch.assign.resize(n);
            MSVS_HIP(hipMemcpy(ch.assign.data(), d_assign.p, n * 4, hipMemcpyDeviceToHost));
            if (ix->shard_world > 1)
            {
                // a shard keeps the rows of ITS lists only, and drops the others NOW: a rank that is shown all 100M rows of an
                // 8-way sharded index stages 12.5M of them, not 100M (the labels were taken from the global staging order above)
                std::vector<uint32_t> keep;
                keep.reserve(n / (size_t)ix->shard_world + 16);
                for (size_t i = 0; i < n; i++)
                    if (ch.assign[i] % ix->shard_world == ix->shard_rank)
                        keep.push_back((uint32_t)i);
                if (keep.size() < n)
                {
                    const size_t m = keep.size();
                    DevBuf<float> kept(std::max<size_t>(m, 1) * ld);
                    if (m)
                    {
                        DevBuf<uint32_t> d_keep(m);
                        MSVS_HIP(hipMemcpyAsync(d_keep.p, keep.data(), m * 4, hipMemcpyHostToDevice, stream));
                        const size_t total = m * (ld / 4);
                        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)ceil_div(total, (size_t)256)), dim3(256), 0, stream,
                                           reinterpret_cast<const float4 *>(ch.x.p), reinterpret_cast<float4 *>(kept.p), d_keep.p, m, ld / 4);
                        MSVS_HIP(hipGetLastError());
                        MSVS_HIP(hipStreamSynchronize(stream));
                    }
                    for (size_t j = 0; j < m; j++)
                    {
                        ch.ids[j] = ch.ids[keep[j]];
                        ch.assign[j] = ch.assign[keep[j]];
                    }
                    ch.ids.resize(m);
                    ch.assign.resize(m);
                    ch.x = std::move(kept);
                    ch.n = m;
                }
            }
        }
        MSVS_HIP(hipStreamSynchronize(stream));
        ix->staged += n;
        if (ch.n)
            ix->chunks.push_back(std::move(ch));
    });
}

extern "C" int msvs_index_build(msvs_index_t * ix)
{
    return guarded([&] {
        DeviceGuard on_device(ix ? ix->device : -1);
        if (!ix)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null index");
        if (ix->ready)
            return;
        poll_cancel(ix);
        hipStream_t stream = nullptr;
        const uint32_t ld = ix->ld;
        const size_t nlist = ix->type == MSVS_INDEX_IVFFLAT ? ix->nlist : 1;
        if (ix->type == MSVS_INDEX_IVFFLAT && nlist == 0)
            fail(MSVS_ERR_NOT_READY, "IVFFLAT index has no centroids");
        // rows kept on this shard, ordered by (list, id)
        struct Ref
        {
            int32_t list;
            uint32_t id;
            uint32_t chunk;
            uint32_t row;
        };
        std::vector<Ref> refs;
        {
            size_t held = 0;
            for (const auto & ch : ix->chunks)
                held += ch.n;
            refs.reserve(held);
        }
        for (size_t c = 0; c < ix->chunks.size(); c++)
        {
            const auto & ch = ix->chunks[c];
            for (size_t i = 0; i < ch.n; i++)
            {
                int32_t l = ix->type == MSVS_INDEX_IVFFLAT ? ch.assign[i] : 0;
                if (ix->type == MSVS_INDEX_IVFFLAT && ix->shard_world > 1 && l % ix->shard_world != ix->shard_rank)
                    continue;
                if (ix->type == MSVS_INDEX_FLAT && ix->shard_world > 1)
                {
                    // FLAT shards by contiguous id ranges of the staged order
                    size_t g = 0;
                    for (size_t cc = 0; cc < c; cc++)
                        g += ix->chunks[cc].n;
                    g += i;
                    size_t per = ceil_div(ix->staged, (size_t)ix->shard_world);
                    if (g / per != (size_t)ix->shard_rank)
                        continue;
                }
                refs.push_back({l, (uint32_t)ch.ids[i], (uint32_t)c, (uint32_t)i});
            }
        }
        std::stable_sort(refs.begin(), refs.end(), [](const Ref & a, const Ref & b) {
            return a.list != b.list ? a.list < b.list : a.id < b.id;
        });
        const size_t n = refs.size();
        ix->n = n;
        ix->h_list_off.assign(nlist + 1, 0);
        ix->max_id = 0;
        for (const auto & r : refs)
        {
            ix->h_list_off[r.list + 1]++;
            ix->max_id = std::max<uint64_t>(ix->max_id, r.id);
        }
        ix->max_list_len = 0;
        for (size_t l = 0; l < nlist; l++)
        {
            ix->max_list_len = std::max<size_t>(ix->max_list_len, (size_t)ix->h_list_off[l + 1]);
            ix->h_list_off[l + 1] += ix->h_list_off[l];
        }
        ix->vecs.alloc(std::max<size_t>(n, 1) * ld);
        ix->row_ids.alloc(std::max<size_t>(n, 1));
        ix->list_off.alloc(nlist + 1);
        std::vector<uint32_t> h_ids(n);
        std::vector<std::vector<uint32_t>> pos(ix->chunks.size());
        std::vector<std::vector<uint32_t>> src(ix->chunks.size());
        for (size_t p = 0; p < n; p++)
        {
            h_ids[p] = refs[p].id;
            pos[refs[p].chunk].push_back((uint32_t)p);
            src[refs[p].chunk].push_back(refs[p].row);
        }
        for (size_t c = 0; c < ix->chunks.size(); c++)
        {
            size_t m = pos[c].size();
            if (m)
            {
                // gather the kept rows of the chunk, then scatter them to their list-major positions
                DevBuf<uint32_t> d_src(m), d_pos(m);
                DevBuf<float> tmp(m * ld);
                MSVS_HIP(hipMemcpyAsync(d_src.p, src[c].data(), m * 4, hipMemcpyHostToDevice, stream));
                MSVS_HIP(hipMemcpyAsync(d_pos.p, pos[c].data(), m * 4, hipMemcpyHostToDevice, stream));
                size_t total = m * (ld / 4);
                hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, stream,
                                   reinterpret_cast<const float4 *>(ix->chunks[c].x.p),
                                   reinterpret_cast<float4 *>(tmp.p), d_src.p, m, ld / 4);
                hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, stream,
                                   reinterpret_cast<const float4 *>(tmp.p), reinterpret_cast<float4 *>(ix->vecs.p),
                                   d_pos.p, m, ld / 4);
                MSVS_HIP(hipGetLastError());
                MSVS_HIP(hipStreamSynchronize(stream));
            }
            ix->chunks[c].x.release();
        }
        if (n)
            MSVS_HIP(hipMemcpy(ix->row_ids.p, h_ids.data(), n * 4, hipMemcpyHostToDevice));
        MSVS_HIP(hipMemcpy(ix->list_off.p, ix->h_list_off.data(), (nlist + 1) * 8, hipMemcpyHostToDevice));
        ix->chunks.clear();
        index_finalize_norms(*ix, stream);
        MSVS_HIP(hipDeviceSynchronize()); // searches run on other (per-thread, non-blocking) streams
        ix->ready = true;
    });
}

extern "C" int msvs_index_ready(const msvs_index_t * ix) { return ix && ix->ready ? 1 : 0; }
extern "C" size_t msvs_index_num_data(const msvs_index_t * ix) { return ix ? (ix->ready ? ix->n : ix->staged) : 0; }
extern "C" size_t msvs_index_num_lists(const msvs_index_t * ix)
{
    return ix ? (ix->type == MSVS_INDEX_IVFFLAT ? ix->nlist : 1) : 0;
}
extern "C" size_t msvs_index_memory_usage(const msvs_index_t * ix)
{
    return ix ? ix->vecs.bytes() + ix->row_ids.bytes() + ix->centroids.bytes() + ix->list_off.bytes()
            + ix->xnorm.bytes() + ix->cnorm.bytes() + ix->list_mid.bytes() + ix->shadow.bytes() + ix->hoff.bytes()
            + ix->list_mid32.bytes()
              : 0;
}

