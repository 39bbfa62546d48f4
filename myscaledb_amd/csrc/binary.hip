// binary.hip -- seam A2 for BinaryVector: msvs_knn_bin (include/msvs.h), see bin_kernels.hpp.
#include <algorithm>

#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "bin_kernels.hpp"
#include "device_ops.hpp"
#include "io_stream.hpp"

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

/// The scan + merge over rows already on the device (dy: n rows of ldb bytes, zero padded to 16-byte words); x, alive_bits
/// (over labels, nbits of them), ids, dis on the HOST.
static void bin_search_rows(const unsigned char * dy, const uint32_t * d_labels, size_t ny, size_t nbytes, const uint8_t * x, size_t nx,
                            size_t k, int metric, const uint64_t * alive_bits, size_t nbits, int64_t * ids, float * dis, hipStream_t stream)
{
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
    const size_t words = alive_bits ? std::max<size_t>(1, ceil_div(nbits, (size_t)64)) : 0;
    Scratch & scr = scratch_for(stream);
    scr.reserve(nx * ldb + nx * (size_t)n_blocks * k * 8 + nx * k * 12 + words * 8 + 8192, stream);
    unsigned char * dq = scr.take<unsigned char>(nx * ldb);
    uint64_t * partial = scr.take<uint64_t>(nx * (size_t)n_blocks * k);
    int64_t * d_ids = scr.take<int64_t>(nx * k);
    float * d_dis = scr.take<float>(nx * k);
    uint64_t * d_alive = words ? scr.take<uint64_t>(words) : nullptr;
    if (ldb != nbytes)
        MSVS_HIP(hipMemsetAsync(dq, 0, nx * ldb, stream));
    MSVS_HIP(hipMemcpy2DAsync(dq, ldb, x, nbytes, nbytes, nx, hipMemcpyHostToDevice, stream));
    if (words)
        MSVS_HIP(hipMemcpyAsync(d_alive, alive_bits, words * 8, hipMemcpyHostToDevice, stream));
    BinParams a{};
    a.Y = reinterpret_cast<const uint4 *>(dy);
    a.Q = reinterpret_cast<const uint4 *>(dq);
    a.alive = d_alive;
    a.labels = d_labels;
    a.nbits = (uint32_t)std::min<size_t>(nbits, 0xffffffffu);
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
}

static void bin_check_args(size_t nbytes, size_t k, int metric)
{
    if (metric != MSVS_METRIC_HAMMING && metric != MSVS_METRIC_JACCARD)
        fail(MSVS_ERR_NOT_IMPLEMENTED, "Metric not implemented in brute force search for Binary Vector");
    if (nbytes == 0)
        fail(MSVS_ERR_INVALID_ARGUMENT, "zero dimension");
    if (k > MSVS_MAX_K)
        fail(MSVS_ERR_UNSUPPORTED_K, "k = %zu exceeds the device top-k limit %d", k, MSVS_MAX_K);
}

extern "C" int msvs_knn_bin(const uint8_t * x, const uint8_t * y, size_t nbytes, size_t k, size_t nx, size_t ny, int metric,
                            const uint64_t * alive_bits, int64_t * ids, float * dis)
{
    return guarded([&] {
        bin_check_args(std::max<size_t>(nbytes, 1), k, metric);
        if (nx == 0 || k == 0)
            return;
        if (!x || !ids || !dis || (ny && !y) || nbytes == 0)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null buffer or zero dimension");
        if (ny > 0xfffffff0ull)
            fail(MSVS_ERR_ID_RANGE, "ny exceeds the u32 id range");
        hipStream_t stream = thread_stream();
        const size_t ldb = round_up(nbytes, (size_t)16);
        DevBuf<unsigned char> dy(std::max<size_t>(ny, 1) * ldb);
        if (ldb != nbytes)
            MSVS_HIP(hipMemsetAsync(dy.p, 0, std::max<size_t>(ny, 1) * ldb, stream));
        if (ny)
            MSVS_HIP(hipMemcpy2DAsync(dy.p, ldb, y, nbytes, nbytes, ny, hipMemcpyHostToDevice, stream));
        bin_search_rows(dy.p, nullptr, ny, nbytes, x, nx, k, metric, alive_bits, ny, ids, dis, stream);
    });
}

// ------------------------------------------------------------------------------------------- seam A1 for BinaryVector
//
// Search::VectorIndex<IS, OS, Bitmap, BinaryVector> (VICommon.h:142-143; created at VIWithDataPart.cpp:431-446, searched at
// :928-935): BinaryFLAT -- and the partition scan of BinaryMSTG, whose algorithm is not in the reference -- as an exhaustive
// scan of RESIDENT rows: the part's FixedString(N) column is uploaded once at build / load, a search sends the query bits and
// the filter bitmap.  Labels are the part's row offsets (what VIPartReader hands out as ids); the filter is indexed by label.

struct msvs_bin_index
{
    size_t nbytes = 0;
    int metric = MSVS_METRIC_HAMMING;
    std::vector<uint8_t> rows;   // host copy: n x nbytes (the serialised form)
    std::vector<int64_t> labels; // n
    mutable std::mutex mu;
    /// The rows as a search reads them, on ONE device: n x round_up(nbytes, 16) + labels, uploaded at the first search (on that
    /// device) after an add.  Searches hold the image through a shared_ptr while they scan: an add that comes in meanwhile
    /// replaces the map entry, the buffers go when the last scan that uses them is done.
    struct Image
    {
        DevBuf<unsigned char> rows;
        DevBuf<uint32_t> labels;
        size_t n = 0;
    };
    mutable std::map<int, std::shared_ptr<Image>> images; // by device; cleared by add
};

extern "C" int msvs_bin_index_create(size_t nbytes, int metric, msvs_bin_index_t ** out)
{
    return guarded([&] {
        if (!out)
            fail(MSVS_ERR_INVALID_ARGUMENT, "out is null");
        bin_check_args(nbytes, 1, metric);
        std::unique_ptr<msvs_bin_index> ix(new msvs_bin_index);
        ix->nbytes = nbytes;
        ix->metric = metric;
        *out = ix.release();
    });
}

extern "C" void msvs_bin_index_free(msvs_bin_index_t * ix) { delete ix; }

extern "C" int msvs_bin_index_add(msvs_bin_index_t * ix, const uint8_t * rows, const int64_t * ids, size_t n)
{
    return guarded([&] {
        if (!ix || (n && !rows))
            fail(MSVS_ERR_INVALID_ARGUMENT, "null index / rows");
        std::lock_guard<std::mutex> lk(ix->mu);
        const size_t base = ix->labels.size();
        if (base + n > 0xfffffff0ull)
            fail(MSVS_ERR_ID_RANGE, "more rows than the u32 label range");
        for (size_t i = 0; i < n; i++)
        {
            const int64_t id = ids ? ids[i] : (int64_t)(base + i);
            if (id < 0 || id > 0xfffffff0ll)
                fail(MSVS_ERR_ID_RANGE, "id %lld does not fit the u32 label range", (long long)id);
            ix->labels.push_back(id);
        }
        ix->rows.insert(ix->rows.end(), rows, rows + n * ix->nbytes);
        ix->images.clear(); // (scans in flight keep theirs alive)
    });
}

extern "C" size_t msvs_bin_index_num_data(const msvs_bin_index_t * ix) { return ix ? ix->labels.size() : 0; }

extern "C" int msvs_bin_index_search(const msvs_bin_index_t * ix, const uint8_t * x, size_t nx, size_t k, const uint64_t * alive_bits,
                                     size_t nbits, int64_t * ids, float * dis)
{
    return guarded([&] {
        if (!ix)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null index");
        bin_check_args(ix->nbytes, k, ix->metric);
        if (nx == 0 || k == 0)
            return;
        if (!x || !ids || !dis)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null buffer");
        hipStream_t stream = thread_stream();
        const size_t ldb = round_up(ix->nbytes, (size_t)16);
        int dev = 0;
        MSVS_HIP(hipGetDevice(&dev));
        std::shared_ptr<msvs_bin_index::Image> img;
        {
            // the row count is read and the upload made under ONE lock: an add between the two would otherwise leave rows that
            // are never searched; the image is per device (a search thread bound to another GPU gets its own copy)
            std::lock_guard<std::mutex> lk(ix->mu);
            auto & slot = ix->images[dev];
            if (!slot)
            {
                const size_t n = ix->labels.size();
                auto fresh = std::make_shared<msvs_bin_index::Image>();
                fresh->n = n;
                fresh->rows.alloc(std::max<size_t>(n, 1) * ldb);
                fresh->labels.alloc(std::max<size_t>(n, 1));
                MSVS_HIP(hipMemsetAsync(fresh->rows.p, 0, std::max<size_t>(n, 1) * ldb, stream));
                if (n)
                {
                    MSVS_HIP(hipMemcpy2DAsync(fresh->rows.p, ldb, ix->rows.data(), ix->nbytes, ix->nbytes, n, hipMemcpyHostToDevice, stream));
                    std::vector<uint32_t> l32(n);
                    for (size_t i = 0; i < n; i++)
                        l32[i] = (uint32_t)ix->labels[i];
                    MSVS_HIP(hipMemcpyAsync(fresh->labels.p, l32.data(), n * 4, hipMemcpyHostToDevice, stream));
                    MSVS_HIP(hipStreamSynchronize(stream)); // l32 is about to go
                }
                MSVS_HIP(hipStreamSynchronize(stream));
                slot = fresh;
            }
            img = slot;
        }
        bin_search_rows(img->rows.p, img->labels.p, img->n, ix->nbytes, x, nx, k, ix->metric, alive_bits, nbits, ids, dis, stream);
    });
}

namespace
{
struct BinHeader // 48 bytes, little endian
{
    char magic[8]; // "MSVSBIN1"
    uint32_t version;
    int32_t metric;
    uint64_t nbytes, n;
    uint64_t reserved[2];
};
}

extern "C" int msvs_bin_index_serialize_io(const msvs_bin_index_t * ix, const msvs_io_t * io)
{
    return guarded([&] {
        if (!ix)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null index");
        std::lock_guard<std::mutex> lk(ix->mu);
        {
            IoStream f(io, "data_bin", 1);
            BinHeader h{};
            memcpy(h.magic, "MSVSBIN1", 8);
            h.version = 1;
            h.metric = ix->metric;
            h.nbytes = ix->nbytes;
            h.n = ix->labels.size();
            f.write(&h, sizeof(h));
            if (!ix->rows.empty())
                f.write(ix->rows.data(), ix->rows.size());
            f.finish();
        }
        {
            IoStream f(io, "id_list", 1);
            const uint64_t n = ix->labels.size();
            f.write(&n, 8);
            if (n)
                f.write(ix->labels.data(), n * 8);
            f.finish();
        }
    });
}

extern "C" int msvs_bin_index_load_io(const msvs_io_t * io, msvs_bin_index_t ** out)
{
    return guarded([&] {
        if (!out)
            fail(MSVS_ERR_INVALID_ARGUMENT, "out is null");
        std::unique_ptr<msvs_bin_index> ix(new msvs_bin_index);
        {
            IoStream f(io, "data_bin", 0);
            BinHeader h{};
            f.read(&h, sizeof(h));
            if (memcmp(h.magic, "MSVSBIN1", 8) != 0 || h.version != 1 || h.nbytes == 0 || h.nbytes > 65536 || h.n > 0xfffffff0ull
                || (h.metric != MSVS_METRIC_HAMMING && h.metric != MSVS_METRIC_JACCARD))
                fail(MSVS_ERR_IO, "corrupt msvs binary index header");
            ix->nbytes = h.nbytes;
            ix->metric = h.metric;
            // the header is untrusted: the rows arrive in pieces and the buffer grows with what has really been read, so a corrupt
            // or truncated file ends in MSVS_ERR_IO (a short read) instead of a 2.8e14-byte allocation
            if (h.nbytes != 0 && h.n > SIZE_MAX / h.nbytes) // (n <= 0xfffffff0 and nbytes <= 65536 above: cannot wrap in 64 bits -- kept explicit)
                fail(MSVS_ERR_IO, "corrupt msvs binary index header");
            const size_t total = (size_t)h.n * (size_t)h.nbytes, piece = (size_t)64 << 20;
            for (size_t got = 0; got < total;)
            {
                const size_t m = std::min(piece, total - got);
                ix->rows.resize(got + m);
                f.read(ix->rows.data() + got, m);
                got += m;
            }
            // (the labels are sized by the id list's own count below, piece by piece like the rows -- not by the header)
        }
        {
            IoStream f(io, "id_list", 0);
            uint64_t n = 0;
            f.read(&n, 8);
            const size_t rows_read = ix->nbytes ? ix->rows.size() / ix->nbytes : 0;
            if (n != rows_read)
                fail(MSVS_ERR_IO, "corrupt msvs binary index: %llu ids for %zu rows", (unsigned long long)n, rows_read);
            for (size_t got = 0; got < n;) // in pieces, like the rows: the buffer grows with what has really been read
            {
                const size_t m = std::min<size_t>((size_t)8 << 20, n - got);
                ix->labels.resize(got + m);
                f.read(ix->labels.data() + got, m * 8);
                got += m;
            }
            for (int64_t id : ix->labels)
                if (id < 0 || id > 0xfffffff0ll)
                    fail(MSVS_ERR_IO, "corrupt msvs binary index: row id %lld outside the u32 row-offset range", (long long)id);
        }
        *out = ix.release();
    });
}
