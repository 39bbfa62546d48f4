// index_io.hip -- an index leaves and enters libmsvs.so: export of the built structure (tests / oracle, list statistics),
// the named-file serialisation through caller-supplied stream openers, resource accounting (include/msvs.h).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "index_internal.hpp"
#include "io_stream.hpp"

using namespace msvs;

extern "C" int msvs_index_export(const msvs_index_t * ix, float * centroids, int64_t * list_off, float * vecs,
                                 int64_t * ids)
{
    return guarded([&] {
        DeviceGuard on_device(ix ? ix->device : -1);
        if (!ix || !ix->ready)
            fail(MSVS_ERR_NOT_READY, "index is not ready");
        const size_t d = ix->dim, ld = ix->ld;
        if (centroids && ix->type == MSVS_INDEX_IVFFLAT)
            MSVS_HIP(hipMemcpy2D(centroids, d * 4, ix->centroids.p, ld * 4, d * 4, ix->nlist, hipMemcpyDeviceToHost));
        if (list_off)
            memcpy(list_off, ix->h_list_off.data(), ix->h_list_off.size() * 8);
        if (vecs && ix->n)
            MSVS_HIP(hipMemcpy2D(vecs, d * 4, ix->vecs.p, ld * 4, d * 4, ix->n, hipMemcpyDeviceToHost));
        if (ids && ix->n)
        {
            std::vector<uint32_t> h(ix->n);
            MSVS_HIP(hipMemcpy(h.data(), ix->row_ids.p, ix->n * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < ix->n; i++)
                ids[i] = (int64_t)h[i];
        }
    });
}

extern "C" int msvs_index_export_list(const msvs_index_t * ix, size_t list, float * vecs, int64_t * ids)
{
    return guarded([&] {
        DeviceGuard on_device(ix ? ix->device : -1);
        if (!ix || !ix->ready || ix->type != MSVS_INDEX_IVFFLAT)
            fail(MSVS_ERR_NOT_READY, "not a built IVFFLAT index");
        if (list >= ix->nlist)
            fail(MSVS_ERR_INVALID_ARGUMENT, "list %zu of %zu", list, ix->nlist);
        const size_t b = (size_t)ix->h_list_off[list], len = (size_t)ix->h_list_off[list + 1] - b, d = ix->dim, ld = ix->ld;
        if (!len)
            return;
        if (vecs)
            MSVS_HIP(hipMemcpy2D(vecs, d * 4, ix->vecs.p + b * ld, ld * 4, d * 4, len, hipMemcpyDeviceToHost));
        if (ids)
        {
            std::vector<uint32_t> h(len);
            MSVS_HIP(hipMemcpy(h.data(), ix->row_ids.p + b, len * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < len; i++)
                ids[i] = (int64_t)h[i];
        }
    });
}

extern "C" int msvs_index_list_stats(const msvs_index_t * ix, size_t * nlist, size_t * min_len, size_t * max_len, double * imbalance,
                                     size_t * train_empty)
{
    return guarded([&] {
        if (!ix || !ix->ready)
            fail(MSVS_ERR_NOT_READY, "index is not ready");
        size_t mn = ~(size_t)0, mx = 0;
        double sq = 0;
        const size_t nl = ix->type == MSVS_INDEX_IVFFLAT ? ix->nlist : 0;
        for (size_t l = 0; l < nl; l++)
        {
            const size_t len = (size_t)(ix->h_list_off[l + 1] - ix->h_list_off[l]);
            mn = std::min(mn, len);
            mx = std::max(mx, len);
            sq += (double)len * (double)len;
        }
        if (nlist)
            *nlist = nl;
        if (min_len)
            *min_len = nl ? mn : 0;
        if (max_len)
            *max_len = mx;
        if (imbalance)
            *imbalance = nl && ix->n ? (double)nl * sq / ((double)ix->n * (double)ix->n) : 1.0;
        if (train_empty)
            *train_empty = ix->train_empty_last;
    });
}

// ------------------------------------------------------------------------------------------- serialisation

// The index is a set of NAMED files written / read through caller-supplied stream callbacks (msvs_io_t), which is how
// the reference's library does it: Search::IndexDataFileWriter<OS>(path_prefix, opener) opens every file of the set through
// the host's opener -- a VectorIndexWriter over IDisk::writeFile, local disk or S3 alike (VectorIndexIO.h:25-166,
// VIWithDataPart.cpp:461-473, :688-700).  Files (the shim turns NAME into <index_name>-NAME.vidx3):
//   data_bin : DataHeader, centroids [nlist][dim] f32 (IVFFLAT), list offsets [nlist + 1] i64, rows [n][dim] f32
//              list-major (cosine: normalised)              -- serialize() / load()
//   id_list  : u64 n, ids [n] i64 in storage order          -- saveDataID() / loadDataID()
// The fp16 shadow and the norms are derived data and are rebuilt at load.

namespace
{
struct DataHeader
{
    char magic[8]; // "MSVSIDX2"
    uint32_t version;
    int32_t type, metric;
    uint32_t shard_rank, shard_world;
    uint32_t reserved;
    uint64_t dim, nlist, n;
};
constexpr uint32_t DATA_VERSION = 2;
/// stdio implementation behind the path convenience calls: file NAME of the set is <prefix>-NAME.vidx3
struct StdioCtx
{
    std::string prefix;
};
void * stdio_open(void * ctx, const char * name, int write)
{
    const std::string path = static_cast<StdioCtx *>(ctx)->prefix + "-" + name + ".vidx3";
    return fopen(path.c_str(), write ? "wb" : "rb");
}
int64_t stdio_write(void *, void * s, const void * p, size_t n) { return (int64_t)fwrite(p, 1, n, static_cast<FILE *>(s)); }
int64_t stdio_read(void *, void * s, void * p, size_t n) { return (int64_t)fread(p, 1, n, static_cast<FILE *>(s)); }
int stdio_close(void *, void * s) { return fclose(static_cast<FILE *>(s)); }
msvs_io_t stdio_io(StdioCtx * c) { return msvs_io_t{c, stdio_open, stdio_write, stdio_read, stdio_close}; }
}

extern "C" int msvs_index_serialize_io(const msvs_index_t * ix, const msvs_io_t * io)
{
    return guarded([&] {
        DeviceGuard on_device(ix ? ix->device : -1);
        if (!ix || !ix->ready)
            fail(MSVS_ERR_NOT_READY, "index is not ready");
        const size_t nlist = msvs_index_num_lists(ix), d = ix->dim, ld = ix->ld;
        {
            IoStream f(io, "data_bin", 1);
            DataHeader h{};
            memcpy(h.magic, "MSVSIDX2", 8);
            h.version = DATA_VERSION;
            h.type = ix->type;
            h.metric = ix->metric;
            h.shard_rank = (uint32_t)ix->shard_rank;
            h.shard_world = (uint32_t)ix->shard_world;
            h.dim = d;
            h.nlist = nlist;
            h.n = ix->n;
            f.write(&h, sizeof(h));
            if (ix->type == MSVS_INDEX_IVFFLAT)
            {
                std::vector<float> cent(nlist * d);
                MSVS_HIP(hipMemcpy2D(cent.data(), d * 4, ix->centroids.p, ld * 4, d * 4, nlist, hipMemcpyDeviceToHost));
                f.write(cent.data(), cent.size() * 4);
            }
            f.write(ix->h_list_off.data(), (nlist + 1) * 8);
            // rows in chunks: a 77 GB shard never sits in host memory at once
            const size_t rows_per = std::max<size_t>(1, IO_CHUNK / (d * 4));
            std::vector<float> buf(std::min(rows_per, std::max<size_t>(ix->n, 1)) * d);
            for (size_t r0 = 0; r0 < ix->n; r0 += rows_per)
            {
                const size_t m = std::min(rows_per, ix->n - r0);
                MSVS_HIP(hipMemcpy2D(buf.data(), d * 4, ix->vecs.p + r0 * ld, ld * 4, d * 4, m, hipMemcpyDeviceToHost));
                f.write(buf.data(), m * d * 4);
            }
            f.finish();
        }
        {
            IoStream f(io, "id_list", 1);
            const uint64_t n = ix->n;
            f.write(&n, 8);
            std::vector<uint32_t> h32(ix->n);
            if (ix->n)
                MSVS_HIP(hipMemcpy(h32.data(), ix->row_ids.p, ix->n * 4, hipMemcpyDeviceToHost));
            std::vector<int64_t> ids(h32.begin(), h32.end());
            f.write(ids.data(), ids.size() * 8);
            f.finish();
        }
    });
}

extern "C" int msvs_index_load_io(const msvs_io_t * io, msvs_index_t ** out)
{
    return guarded([&] {
        if (!out)
            fail(MSVS_ERR_INVALID_ARGUMENT, "out is null");
        *out = nullptr;
        std::unique_ptr<msvs_index> ix(new msvs_index);
        MSVS_HIP(hipGetDevice(&ix->device));
        size_t nlist = 0, n = 0, d = 0;
        {
            IoStream f(io, "data_bin", 0);
            DataHeader h{};
            f.read(&h, sizeof(h));
            // nothing of the file is trusted: a corrupt header must not turn into out-of-bounds device reads or TB allocations
            if (memcmp(h.magic, "MSVSIDX2", 8) != 0 || h.version != DATA_VERSION)
                fail(MSVS_ERR_IO, "not an msvs index (data_bin: bad magic / version)");
            if ((h.type != MSVS_INDEX_FLAT && h.type != MSVS_INDEX_IVFFLAT)
                || (h.metric != MSVS_METRIC_L2 && h.metric != MSVS_METRIC_IP && h.metric != MSVS_METRIC_COSINE)
                || h.dim == 0 || h.dim > 8192 || h.nlist == 0 || h.nlist > 0x7fffffffull || h.n > 0xfffffff0ull
                || (h.type == MSVS_INDEX_FLAT && h.nlist != 1) || h.shard_world == 0 || h.shard_rank >= h.shard_world)
                fail(MSVS_ERR_IO, "corrupt msvs index header");
            ix->type = h.type;
            ix->metric = h.metric;
            ix->dim = d = h.dim;
            ix->ld = padded_dim(d);
            ix->shard_rank = (int)h.shard_rank;
            ix->shard_world = (int)h.shard_world;
            nlist = h.nlist;
            n = h.n;
            const uint32_t ld = ix->ld;
            if (ix->type == MSVS_INDEX_IVFFLAT)
            {
                std::vector<float> cent(nlist * d);
                f.read(cent.data(), cent.size() * 4);
                ix->nlist = nlist;
                ix->centroids.alloc(nlist * ld);
                upload_rows(ix->centroids.p, cent.data(), nlist, (uint32_t)d, ld, MSVS_MEM_HOST, nullptr);
                MSVS_HIP(hipStreamSynchronize(nullptr));
            }
            std::vector<int64_t> off(nlist + 1);
            f.read(off.data(), off.size() * 8);
            bool ok = off[0] == 0 && off[nlist] == (int64_t)n;
            for (size_t l = 0; ok && l < nlist; l++)
                ok = off[l + 1] >= off[l];
            if (!ok)
                fail(MSVS_ERR_IO, "corrupt msvs index: list offsets are not a partition of the rows");
            ix->n = n;
            ix->h_list_off = off;
            ix->max_list_len = 0;
            for (size_t l = 0; l < nlist; l++)
                ix->max_list_len = std::max<size_t>(ix->max_list_len, (size_t)(off[l + 1] - off[l]));
            ix->vecs.alloc(std::max<size_t>(n, 1) * ld);
            ix->list_off.alloc(nlist + 1);
            MSVS_HIP(hipMemcpy(ix->list_off.p, off.data(), (nlist + 1) * 8, hipMemcpyHostToDevice));
            const size_t rows_per = std::max<size_t>(1, IO_CHUNK / (d * 4));
            std::vector<float> buf(std::min(rows_per, std::max<size_t>(n, 1)) * d);
            for (size_t r0 = 0; r0 < n; r0 += rows_per)
            {
                const size_t m = std::min(rows_per, n - r0);
                f.read(buf.data(), m * d * 4); // a short file fails here, before anything is searched
                upload_rows(ix->vecs.p + r0 * ld, buf.data(), m, (uint32_t)d, ld, MSVS_MEM_HOST, nullptr);
                MSVS_HIP(hipStreamSynchronize(nullptr));
            }
        }
        {
            IoStream f(io, "id_list", 0);
            uint64_t nid = 0;
            f.read(&nid, 8);
            if (nid != n)
                fail(MSVS_ERR_IO, "corrupt msvs index: id_list holds %llu ids for %zu rows", (unsigned long long)nid, n);
            std::vector<int64_t> ids(n);
            f.read(ids.data(), n * 8);
            std::vector<uint32_t> h32(n);
            ix->max_id = 0;
            for (size_t i = 0; i < n; i++)
            {
                if (ids[i] < 0 || ids[i] > 0xfffffff0ll)
                    fail(MSVS_ERR_IO, "corrupt msvs index: row id %lld outside the u32 row-offset range", (long long)ids[i]);
                h32[i] = (uint32_t)ids[i];
                ix->max_id = std::max<uint64_t>(ix->max_id, (uint64_t)ids[i]);
            }
            ix->row_ids.alloc(std::max<size_t>(n, 1));
            if (n)
                MSVS_HIP(hipMemcpy(ix->row_ids.p, h32.data(), n * 4, hipMemcpyHostToDevice));
            // list-major storage order says nothing about the labels: distinct or not is decided by looking (once per load)
            std::sort(h32.begin(), h32.end());
            ix->ids_may_repeat = std::adjacent_find(h32.begin(), h32.end()) != h32.end();
        }
        index_finalize_norms(*ix, nullptr);
        MSVS_HIP(hipDeviceSynchronize());
        ix->ready = true;
        *out = ix.release();
    });
}

/// Convenience over stdio: the file set <path_prefix>-data_bin.vidx3, <path_prefix>-id_list.vidx3.
extern "C" int msvs_index_serialize(const msvs_index_t * ix, const char * path_prefix)
{
    if (!path_prefix)
        return guarded([] { fail(MSVS_ERR_INVALID_ARGUMENT, "null path"); });
    StdioCtx c{path_prefix};
    const msvs_io_t io = stdio_io(&c);
    return msvs_index_serialize_io(ix, &io);
}

extern "C" int msvs_index_load(const char * path_prefix, msvs_index_t ** out)
{
    if (!path_prefix)
        return guarded([] { fail(MSVS_ERR_INVALID_ARGUMENT, "null path"); });
    StdioCtx c{path_prefix};
    const msvs_io_t io = stdio_io(&c);
    return msvs_index_load_io(&io, out);
}

/// Search::VectorIndex::getVersion().toString() -- what VIMetadata records as `version:` and hands back as the
/// `load_index_version` parameter at load (VIWithDataPart.cpp:485, :645).
extern "C" const char * msvs_index_version(void) { return "msvs-2"; }

/// Search::VectorIndex::getResourceUsage() (VIWithDataPart.cpp:368-385, :486-488): bytes resident in HBM, bytes of the
/// serialised file set, and the peak of the build (staging chunks + final storage).
extern "C" int msvs_index_resource_usage(const msvs_index_t * ix, size_t * memory_usage_bytes, size_t * disk_usage_bytes,
                                         size_t * build_memory_usage_bytes)
{
    return guarded([&] {
        if (!ix)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null index");
        const size_t n = ix->ready ? ix->n : ix->staged, nlist = msvs_index_num_lists(ix);
        const size_t disk = sizeof(DataHeader) + (ix->type == MSVS_INDEX_IVFFLAT ? nlist * ix->dim * 4 : 0) + (nlist + 1) * 8
            + n * ix->dim * 4 + 8 + n * 8;
        if (memory_usage_bytes)
            *memory_usage_bytes = msvs_index_memory_usage(ix);
        if (disk_usage_bytes)
            *disk_usage_bytes = disk;
        if (build_memory_usage_bytes)
            *build_memory_usage_bytes = 2 * n * (size_t)ix->ld * 4 + n * 6 * ix->dim / 4 + n * 24;
    });
}
