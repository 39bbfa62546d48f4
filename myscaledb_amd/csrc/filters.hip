// filters.hip -- PREWHERE filters of libmsvs.so (include/msvs.h: msvs_filter_*, msvs_index_search_filter[_device]) and the state that
// rides on a cached index (delete bitmap, decoupled-part row-id maps): producer kernels in filter_kernels.hpp, the consumer is
// either a bit test inside the scan or a compacted view of the index built per search.
#include <algorithm>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "filter_kernels.hpp"
#include "index_internal.hpp"

using namespace msvs;

// ------------------------------------------------------------------------------------------ filters

struct msvs_filter
{
    DevBuf<uint64_t> bits;
    size_t nbits = 0;
    uint64_t count = 0; // passing rows (kept current by every operation: the search strategy reads it)
};

namespace
{
void filter_recount(msvs_filter & f, hipStream_t stream)
{
    const size_t words = std::max<size_t>(1, ceil_div(f.nbits, (size_t)64));
    Scratch & v = view_for(stream); // a counter word without a hipMalloc per filter
    v.reserve(256, stream);
    unsigned long long * c = v.take<unsigned long long>(1);
    MSVS_HIP(hipMemsetAsync(c, 0, 8, stream));
    hipLaunchKernelGGL(filter_count_kernel, dim3((unsigned)ceil_div(words, (size_t)256)), dim3(256), 0, stream, f.bits.p, words, f.nbits, c);
    unsigned long long h = 0;
    MSVS_HIP(hipMemcpyAsync(&h, c, 8, hipMemcpyDeviceToHost, stream));
    MSVS_HIP(hipStreamSynchronize(stream));
    f.count = h;
}

std::unique_ptr<msvs_filter> filter_alloc(size_t nbits, hipStream_t stream)
{
    std::unique_ptr<msvs_filter> f(new msvs_filter);
    f->nbits = nbits;
    const size_t words = std::max<size_t>(1, ceil_div(nbits, (size_t)64));
    f->bits.alloc(words);
    MSVS_HIP(hipMemsetAsync(f->bits.p, 0, words * 8, stream));
    return f;
}

/// Build the compacted view of `ix` under the effective filter (everything enqueued on `stream`, scratch from view_for()).
SearchView build_view(const msvs_index & ix, const uint64_t * d_alive, size_t nbits, size_t alive_upper, hipStream_t stream)
{
    const size_t n = ix.n, chunks = std::max<size_t>(1, ceil_div(n, (size_t)COMPACT_CHUNK));
    // rows that can pass: the filter's population count -- unless labels may repeat (three rows with label 5 pass one bit)
    const size_t upper = std::max<size_t>(1, ix.ids_may_repeat ? n : std::min(n, alive_upper));
    Scratch & v = view_for(stream);
    v.reserve((chunks + 2) * 4 + (n + 2) * 4 + upper * 4 + (ix.nlist + 2) * 8 + 4096, stream);
    CompactParams p{};
    p.ids = ix.row_ids.p;
    p.alive = d_alive;
    p.nbits = (uint32_t)std::min<size_t>(nbits, 0xffffffffu);
    p.n = (uint32_t)n;
    p.chunk_cnt = v.take<uint32_t>(chunks + 1);
    p.rank = v.take<uint32_t>(n + 1);
    p.rowmap = v.take<uint32_t>(upper);
    p.rowmap_cap = (uint32_t)upper;
    p.list_off = ix.type == MSVS_INDEX_IVFFLAT ? ix.list_off.p : nullptr;
    p.nlist = (uint32_t)ix.nlist;
    p.sel_off = v.take<int64_t>(ix.nlist + 1);
    ProfileScope prof("filter_view", stream);
    hipLaunchKernelGGL(compact_count_kernel, dim3((unsigned)chunks), dim3(BLOCK), 0, stream, p);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, stream, p, (uint32_t)chunks);
    hipLaunchKernelGGL(compact_fill_kernel, dim3((unsigned)chunks), dim3(BLOCK), 0, stream, p);
    if (p.list_off)
        hipLaunchKernelGGL(compact_offsets_kernel, dim3((unsigned)ceil_div(ix.nlist + 1, (size_t)256)), dim3(256), 0, stream, p);
    MSVS_HIP(hipGetLastError());
    SearchView view{};
    view.list_off = p.sel_off;
    view.rowmap = p.rowmap;
    view.n_rows = p.rank + n;
    view.n_upper = upper;
    return view;
}

}

namespace msvs
{
/// The search of a filtered batch: selective filters go through the compacted view, the others through the bit test.
/// alive_count: passing rows of the caller's filter (an upper bound of what passes the effective filter inside the index).
void index_search_filtered(const msvs_index & ix, const float * d_queries, size_t nq, uint32_t k, size_t nprobe, const uint64_t * eff,
                           size_t eff_bits, uint64_t alive_count, int64_t * d_ids, float * d_dis, hipStream_t stream)
{
    const double frac = ix.n ? (double)alive_count / (double)ix.n : 1.0;
    double below = options().filter_compact_below;
    if (below < 0)
    {
        // measured crossover (profiles/r02_filter.txt, 1M x 768): the view is scanned canonically, so it competes with the
        // matrix-core candidate pass once many queries share a list pass -- 16 queries: wins below ~50 % passing,
        // 256: below ~30 %, 4096: below ~3 %
        const double per_list = ix.type == MSVS_INDEX_IVFFLAT ? (double)nq * (double)std::max<size_t>(nprobe, 1) / (double)std::max<size_t>(ix.nlist, 1)
                                                               : (double)nq;
        below = per_list < 2 ? 0.4 : (per_list < 16 ? 0.25 : 0.03);
    }
    // the view costs ~5 B per stored row: for one or two queries that is more than the probed lists themselves
    const bool worth = nq * std::max<size_t>(nprobe, 1) * 4 >= ix.nlist || ix.type == MSVS_INDEX_FLAT;
    if (eff && ix.n && (below >= 1.0 || (frac < below && worth)) && alive_count <= 0xfffffff0ull && ix.n <= 0xfffffff0ull)
    {
        const SearchView view = build_view(ix, eff, eff_bits, (size_t)alive_count, stream);
        index_search_device(ix, d_queries, nq, k, nprobe, nullptr, 0, d_ids, d_dis, stream, nullptr, nullptr, &view);
        return;
    }
    index_search_device(ix, d_queries, nq, k, nprobe, eff, eff_bits, d_ids, d_dis, stream);
}
}

extern "C" int msvs_filter_from_bits(const uint64_t * bits, size_t nbits, msvs_filter_t ** out)
{
    return guarded([&] {
        if (!out || (nbits && !bits))
            fail(MSVS_ERR_INVALID_ARGUMENT, "null argument");
        hipStream_t stream = thread_stream();
        auto f = filter_alloc(nbits, stream);
        if (nbits)
            MSVS_HIP(hipMemcpyAsync(f->bits.p, bits, ceil_div(nbits, (size_t)64) * 8, hipMemcpyHostToDevice, stream));
        filter_recount(*f, stream);
        *out = f.release();
    });
}

/// getFilterFromPipeline (MergeTreeSelectWithHybridSearchProcessor.cpp:905-934): one bit per passing `_part_offset`.
extern "C" int msvs_filter_from_offsets(const uint64_t * part_offsets, size_t n, size_t nbits, int mem, msvs_filter_t ** out)
{
    return guarded([&] {
        if (!out || (n && !part_offsets))
            fail(MSVS_ERR_INVALID_ARGUMENT, "null argument");
        hipStream_t stream = thread_stream();
        auto f = filter_alloc(nbits, stream);
        if (n)
        {
            const uint64_t * d_off = part_offsets;
            DevBuf<uint64_t> tmp;
            if (mem != MSVS_MEM_DEVICE)
            {
                tmp.alloc(n);
                MSVS_HIP(hipMemcpyAsync(tmp.p, part_offsets, n * 8, hipMemcpyHostToDevice, stream));
                d_off = tmp.p;
            }
            hipLaunchKernelGGL(filter_from_offsets_kernel, dim3((unsigned)ceil_div(n, (size_t)256)), dim3(256), 0, stream, d_off, n, nbits,
                               reinterpret_cast<unsigned long long *>(f->bits.p));
            MSVS_HIP(hipGetLastError());
            filter_recount(*f, stream); // also orders tmp's release after the kernel
        }
        *out = f.release();
    });
}

namespace
{
template <typename T>
void launch_predicate(const void * col, size_t n, int mem, int op, T lo, T hi, msvs_filter & f, hipStream_t stream)
{
    const T * d_col = static_cast<const T *>(col);
    DevBuf<T> tmp;
    if (mem != MSVS_MEM_DEVICE)
    {
        tmp.alloc(std::max<size_t>(n, 1));
        MSVS_HIP(hipMemcpyAsync(tmp.p, col, n * sizeof(T), hipMemcpyHostToDevice, stream));
        d_col = tmp.p;
    }
    hipLaunchKernelGGL((filter_predicate_kernel<T>), dim3((unsigned)ceil_div(n, (size_t)256)), dim3(256), 0, stream, d_col, n, op, lo, hi,
                       f.bits.p);
    MSVS_HIP(hipGetLastError());
    filter_recount(f, stream);
}
}

/// A simple PREWHERE predicate `column OP constant` evaluated on the device: row i of the column is `_part_offset` i.
extern "C" int msvs_filter_from_predicate(const void * column, int dtype, size_t nrows, int mem, int op, msvs_scalar_t lo,
                                          msvs_scalar_t hi, msvs_filter_t ** out)
{
    return guarded([&] {
        if (!out || (nrows && !column))
            fail(MSVS_ERR_INVALID_ARGUMENT, "null argument");
        if (op < MSVS_OP_EQ || op > MSVS_OP_BETWEEN)
            fail(MSVS_ERR_INVALID_ARGUMENT, "unknown comparison %d", op);
        hipStream_t stream = thread_stream();
        auto f = filter_alloc(nrows, stream);
        if (nrows)
            switch (dtype)
            {
                case MSVS_DT_UINT8: launch_predicate<uint8_t>(column, nrows, mem, op, (uint8_t)lo.i, (uint8_t)hi.i, *f, stream); break;
                case MSVS_DT_UINT16: launch_predicate<uint16_t>(column, nrows, mem, op, (uint16_t)lo.i, (uint16_t)hi.i, *f, stream); break;
                case MSVS_DT_UINT32: launch_predicate<uint32_t>(column, nrows, mem, op, (uint32_t)lo.i, (uint32_t)hi.i, *f, stream); break;
                case MSVS_DT_UINT64: launch_predicate<uint64_t>(column, nrows, mem, op, (uint64_t)lo.i, (uint64_t)hi.i, *f, stream); break;
                case MSVS_DT_INT8: launch_predicate<int8_t>(column, nrows, mem, op, (int8_t)lo.i, (int8_t)hi.i, *f, stream); break;
                case MSVS_DT_INT16: launch_predicate<int16_t>(column, nrows, mem, op, (int16_t)lo.i, (int16_t)hi.i, *f, stream); break;
                case MSVS_DT_INT32: launch_predicate<int32_t>(column, nrows, mem, op, (int32_t)lo.i, (int32_t)hi.i, *f, stream); break;
                case MSVS_DT_INT64: launch_predicate<int64_t>(column, nrows, mem, op, lo.i, hi.i, *f, stream); break;
                case MSVS_DT_FLOAT32: launch_predicate<float>(column, nrows, mem, op, (float)lo.f, (float)hi.f, *f, stream); break;
                case MSVS_DT_FLOAT64: launch_predicate<double>(column, nrows, mem, op, lo.f, hi.f, *f, stream); break;
                default: fail(MSVS_ERR_INVALID_ARGUMENT, "unknown column type %d", dtype);
            }
        *out = f.release();
    });
}

extern "C" int msvs_filter_combine(msvs_filter_t * a, const msvs_filter_t * b, int mode)
{
    return guarded([&] {
        if (!a || !b || mode < 0 || mode > 2)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null filter / unknown mode");
        hipStream_t stream = thread_stream();
        const size_t wa = std::max<size_t>(1, ceil_div(a->nbits, (size_t)64)), wb = ceil_div(b->nbits, (size_t)64);
        hipLaunchKernelGGL(filter_combine_kernel, dim3((unsigned)ceil_div(wa, (size_t)256)), dim3(256), 0, stream, a->bits.p, wa, b->bits.p,
                           wb, mode);
        MSVS_HIP(hipGetLastError());
        filter_recount(*a, stream);
    });
}

extern "C" int msvs_filter_count(const msvs_filter_t * f, uint64_t * alive, size_t * nbits)
{
    return guarded([&] {
        if (!f)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null filter");
        if (alive)
            *alive = f->count;
        if (nbits)
            *nbits = f->nbits;
    });
}

extern "C" int msvs_filter_to_bits(const msvs_filter_t * f, uint64_t * bits_out)
{
    return guarded([&] {
        if (!f || !bits_out)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null argument");
        MSVS_HIP(hipMemcpy(bits_out, f->bits.p, std::max<size_t>(1, ceil_div(f->nbits, (size_t)64)) * 8, hipMemcpyDeviceToHost));
    });
}

extern "C" void msvs_filter_free(msvs_filter_t * f) { delete f; }

extern "C" int msvs_index_search_filter_device(const msvs_index_t * ix, const float * d_queries, size_t nq, int k, int nprobe,
                                               const msvs_filter_t * filter, int64_t * d_ids, float * d_dis, void * hip_stream)
{
    return guarded([&] {
        if (!ix || !filter || (nq && (!d_queries || !d_ids || !d_dis)) || k < 0)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null index/filter/buffer or negative k");
        if (nq == 0 || k == 0)
            return;
        check_k((size_t)k);
        hipStream_t stream = as_stream(hip_stream);
        const auto meta = ix->get_meta();
        size_t eff_bits = filter->nbits;
        const uint64_t * eff = effective_filter(*ix, meta.get(), filter->bits.p, filter->nbits, &eff_bits, stream);
        index_search_filtered(*ix, d_queries, nq, (uint32_t)k, (size_t)std::max(nprobe, 0), eff, eff_bits, filter->count, d_ids, d_dis,
                              stream);
        apply_row_ids_map(meta.get(), d_ids, nq * (size_t)k, stream);
    });
}

extern "C" int msvs_index_search_filter(const msvs_index_t * ix, const float * queries, size_t nq, int k, const char * params,
                                        const msvs_filter_t * filter, int64_t * ids, float * dis)
{
    return guarded([&] {
        DeviceGuard on_device(ix ? ix->device : -1);
        if (!ix || !filter || (nq && (!queries || !ids || !dis)) || k < 0)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null index/filter/buffer or negative k");
        if (nq == 0 || k == 0)
            return;
        check_k((size_t)k);
        auto p = parse_params(params);
        for (const auto & kv : p)
            if (kv.first != "nprobe")
                fail(MSVS_ERR_INVALID_ARGUMENT, "unknown search parameter `%s`", kv.first.c_str());
        const long nprobe = param_int(p, "nprobe", 1);
        if (nprobe < 1)
            fail(MSVS_ERR_INVALID_ARGUMENT, "nprobe must be >= 1");
        hipStream_t stream = thread_stream();
        Scratch & stg = staging_for(stream);
        stg.reserve(nq * ix->dim * 4 + nq * (size_t)k * 12 + 4096, stream);
        float * dq = stg.take<float>(nq * ix->dim);
        int64_t * d_ids = stg.take<int64_t>(nq * (size_t)k);
        float * d_dis = stg.take<float>(nq * (size_t)k);
        MSVS_HIP(hipMemcpyAsync(dq, queries, nq * ix->dim * 4, hipMemcpyHostToDevice, stream));
        const int rc = msvs_index_search_filter_device(ix, dq, nq, k, (int)nprobe, filter, d_ids, d_dis, stream);
        if (rc != MSVS_OK)
            fail(rc, "%s", msvs_last_error());
        MSVS_HIP(hipMemcpyAsync(ids, d_ids, nq * (size_t)k * 8, hipMemcpyDeviceToHost, stream));
        MSVS_HIP(hipMemcpyAsync(dis, d_dis, nq * (size_t)k * 4, hipMemcpyDeviceToHost, stream));
        MSVS_HIP(hipStreamSynchronize(stream));
    });
}

/// VIWithMeta::setDeleteBitmap (VICacheObject.h:100-102): the lightweight-delete state of a cached index, resident in HBM
/// and swapped atomically; every search ANDs it into its filter (VIWithDataPart.cpp:903-908).  alive_bits NULL clears it.
extern "C" int msvs_index_set_delete_bitmap(msvs_index_t * ix, const uint64_t * alive_bits, size_t nbits)
{
    return guarded([&] {
        DeviceGuard on_device(ix ? ix->device : -1);
        if (!ix)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null index");
        auto cur = ix->get_meta();
        auto next = std::make_shared<msvs_index::Meta>();
        if (alive_bits)
        {
            const size_t words = std::max<size_t>(1, ceil_div(nbits, (size_t)64));
            next->delete_alive.alloc(words);
            MSVS_HIP(hipMemset(next->delete_alive.p, 0, words * 8));
            if (nbits)
                MSVS_HIP(hipMemcpy(next->delete_alive.p, alive_bits, ceil_div(nbits, (size_t)64) * 8, hipMemcpyHostToDevice));
            next->delete_nbits = nbits;
        }
        if (cur) // the maps are immutable once set: share them by copying device to device
        {
            auto dup = [](auto & dst, const auto & src) {
                if (src.n)
                {
                    dst.alloc(src.n);
                    MSVS_HIP(hipMemcpy(dst.p, src.p, src.bytes(), hipMemcpyDeviceToDevice));
                }
            };
            dup(next->row_ids_map, cur->row_ids_map);
            dup(next->inv_row_ids, cur->inv_row_ids);
            dup(next->inv_sources, cur->inv_sources);
            next->row_ids_n = cur->row_ids_n;
            next->inv_n = cur->inv_n;
            next->own_id = cur->own_id;
        }
        std::lock_guard<std::mutex> lk(ix->meta_mu);
        ix->meta = next;
    });
}

