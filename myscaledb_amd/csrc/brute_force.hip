// brute_force.hip -- seam A2 of libmsvs.so (include/msvs.h): the exhaustive scans the host reaches through faiss::knn_* today
// (msvs_knn_f32[_filtered], msvs_normalize_f32) and the resident row blocks that keep a part's marks in HBM across calls
// (msvs_cache_*, msvs_block_*, msvs_knn_resident: VICacheObject semantics -- LRU by bytes, pins).
#include <algorithm>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "index_internal.hpp"

using namespace msvs;

// =========================================================================================== seam A2

namespace msvs
{
/// Shared body of msvs_knn_f32 / msvs_knn_f32_filtered: host buffers in, host buffers out.
/// d_resident (nullable): the base rows already in HBM (row stride = padded_dim(d) floats, zero padded) -- a block of
/// the resident cache (msvs_block_t); y is then ignored and nothing but the queries crosses PCIe.
static void knn_host(const float * x, const float * y, size_t d, size_t k, size_t nx, size_t ny, int metric,
                     const uint64_t * alive_bits, int64_t * ids, float * dis, const float * d_resident = nullptr)
{
    if (metric != MSVS_METRIC_L2 && metric != MSVS_METRIC_IP)
        fail(MSVS_ERR_NOT_IMPLEMENTED, "Metric not implemented in brute force search for Float32 Vector");
    if (nx == 0 || k == 0)
        return;
    if (!x || !ids || !dis || (ny && !y && !d_resident) || d == 0)
        fail(MSVS_ERR_INVALID_ARGUMENT, "null buffer or zero dimension");
    if (k > MSVS_MAX_K_ROUNDS)
        fail(MSVS_ERR_UNSUPPORTED_K, "k = %zu exceeds the limit %d", k, MSVS_MAX_K_ROUNDS);
    if (ny > 0xfffffff0ull)
        fail(MSVS_ERR_ID_RANGE, "ny exceeds the u32 id range");
    hipStream_t stream = thread_stream();
    const uint32_t ld = padded_dim(d);
    const uint32_t kpass = (uint32_t)std::min<size_t>(k, MSVS_MAX_K);
    const size_t bw = ceil_div(std::max<size_t>(ny, 1), 64);
    const bool rounds = k > MSVS_MAX_K;
    Scratch & scr = scratch_for(stream);
    size_t need = (nx + (d_resident ? 0 : ny)) * (size_t)ld * 4 + nx * k * 12 + bw * 8
        + flat_scratch_bytes(ny, rounds ? 1 : nx, kpass, ld) + 16384;
    scr.reserve(need, stream);
    float * dq = scr.take<float>(nx * ld);
    const float * dy = d_resident;
    if (!d_resident)
    {
        float * up = scr.take<float>(std::max<size_t>(ny, 1) * ld);
        upload_rows(up, y, ny, (uint32_t)d, ld, MSVS_MEM_HOST, stream);
        dy = up;
    }
    int64_t * d_ids = scr.take<int64_t>(nx * k);
    float * d_dis = scr.take<float>(nx * k);
    uint64_t * bm = (alive_bits || rounds) ? scr.take<uint64_t>(bw) : nullptr;
    const size_t mark = scr.used; // everything taken after this point is per-pass scratch
    upload_rows(dq, x, nx, (uint32_t)d, ld, MSVS_MEM_HOST, stream);
    auto load_filter = [&]() {
        if (alive_bits)
            MSVS_HIP(hipMemcpyAsync(bm, alive_bits, bw * 8, hipMemcpyHostToDevice, stream));
        else if (bm)
            MSVS_HIP(hipMemsetAsync(bm, 0xFF, bw * 8, stream));
    };
    MergeParams out{};
    if (!rounds)
    {
        load_filter();
        out.out_ids = d_ids;
        out.out_dis = d_dis;
        flat_search_device(scr, metric, dy, nullptr, ny, ld, dq, nx, (uint32_t)k, bm, ny, out, stream);
    }
    else
    {
        // rounds of MSVS_MAX_K per query, excluding what was already returned (see msvs_index_search)
        for (size_t q = 0; q < nx; q++)
        {
            load_filter();
            for (size_t done = 0; done < k; done += MSVS_MAX_K)
            {
                const uint32_t kr = (uint32_t)std::min<size_t>(MSVS_MAX_K, k - done);
                scr.used = mark;
                out.out_ids = d_ids + q * k + done;
                out.out_dis = d_dis + q * k + done;
                flat_search_device(scr, metric, dy, nullptr, ny, ld, dq + q * ld, 1, kr, bm, ny, out, stream);
                hipLaunchKernelGGL(clear_bits_kernel, dim3(1), dim3(256), 0, stream, bm, out.out_ids, kr);
                MSVS_HIP(hipGetLastError());
            }
        }
    }
    MSVS_HIP(hipMemcpyAsync(ids, d_ids, nx * k * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    MSVS_HIP(hipMemcpyAsync(dis, d_dis, nx * k * sizeof(float), hipMemcpyDeviceToHost, stream));
    MSVS_HIP(hipStreamSynchronize(stream));
}
}

extern "C" int msvs_knn_f32(const float * x, const float * y, size_t d, size_t k, size_t nx, size_t ny, int metric,
                            int64_t * ids, float * dis)
{
    return guarded([&] { knn_host(x, y, d, k, nx, ny, metric, nullptr, ids, dis); });
}

extern "C" int msvs_knn_f32_filtered(const float * x, const float * y, size_t d, size_t k, size_t nx, size_t ny,
                                     int metric, const uint64_t * alive_bits, int64_t * ids, float * dis)
{
    return guarded([&] { knn_host(x, y, d, k, nx, ny, metric, alive_bits, ids, dis); });
}

namespace msvs
{
void normalize_device_rows(float * d_x, size_t n, uint32_t d, uint32_t ld, hipStream_t stream)
{
    if (n == 0)
        return;
    if ((size_t)d * 4 > 60 * 1024) // the row is staged in LDS
        fail(MSVS_ERR_INVALID_ARGUMENT, "dimension %u too large to normalise on the device", d);
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(normalize_rows_grid(n)), dim3(WAVE), (size_t)d * 4, stream, d_x, n, d, ld);
    MSVS_HIP(hipGetLastError());
}
}

extern "C" int msvs_normalize_f32(float * x, size_t n, size_t d)
{
    return guarded([&] {
        if (n == 0)
            return;
        if (!x || d == 0)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null buffer or zero dimension");
        hipStream_t stream = nullptr;
        Scratch & scr = scratch_for(stream);
        scr.reserve(n * d * 4 + 4096, stream);
        float * dx = scr.take<float>(n * d);
        MSVS_HIP(hipMemcpyAsync(dx, x, n * d * 4, hipMemcpyHostToDevice, stream));
        normalize_device_rows(dx, n, (uint32_t)d, (uint32_t)d, stream);
        MSVS_HIP(hipMemcpyAsync(x, dx, n * d * 4, hipMemcpyDeviceToHost, stream));
        MSVS_HIP(hipStreamSynchronize(stream));
    });
}

// =========================================================================================== resident blocks (f1)
//
// The GPU analogue of the reference's VICacheManager / VIWithMeta for the BRUTE-FORCE path: the
// dense block a mark of a part turns into (MergeTreeVSManager.cpp:1380-1392) is uploaded once, keyed by
// (part key, mark), and stays in HBM in an LRU bounded by bytes; later queries against the same part send only the
// query vectors over PCIe.  Blocks are immutable; lightweight deletes arrive per search as the row_exists bitmap, like in
// the reference.  A part that is dropped or mutated is evicted by key prefix (CacheKey: table path / part name, VICacheObject.h:119-162).

#include <list>
#include <unordered_map>
#include <condition_variable>
#include <deque>

struct msvs_block
{
    msvs_cache_t * owner = nullptr;
    std::string key;
    DevBuf<float> rows; // n x ld, zero padded; normalised when `normalized`
    size_t n = 0, d = 0;
    uint32_t ld = 0;
    int normalized = 0;
    int pins = 0;
    bool doomed = false; // evicted while pinned: freed at the last release
    std::list<msvs_block *>::iterator pos;
    size_t bytes() const { return rows.bytes(); }
};

struct msvs_cache
{
    std::mutex mu;
    size_t capacity = 0, used = 0;
    std::list<msvs_block *> lru; // front = most recently used
    std::unordered_map<std::string, msvs_block *> map;
    uint64_t hits = 0, misses = 0, evictions = 0;
    int device = 0;

    static std::string full_key(const char * key, uint64_t mark) { return std::string(key ? key : "") + "#" + std::to_string(mark); }
    void drop_locked(msvs_block * b)
    {
        map.erase(b->key);
        lru.erase(b->pos);
        used -= b->bytes();
        evictions++;
        if (b->pins == 0)
            delete b;
        else
            b->doomed = true;
    }
    void make_room_locked(size_t need)
    {
        // least recently used first; pinned blocks stay (the bound is soft while searches hold more than the capacity)
        std::vector<msvs_block *> victims;
        size_t freed = 0;
        for (auto it = lru.rbegin(); it != lru.rend() && used - freed + need > capacity; ++it)
            if ((*it)->pins == 0)
            {
                victims.push_back(*it);
                freed += (*it)->bytes();
            }
        for (msvs_block * b : victims)
            drop_locked(b);
    }
};

extern "C" int msvs_cache_create(size_t capacity_bytes, msvs_cache_t ** out)
{
    return guarded([&] {
        if (!out)
            fail(MSVS_ERR_INVALID_ARGUMENT, "out is null");
        std::unique_ptr<msvs_cache> c(new msvs_cache);
        c->capacity = capacity_bytes;
        MSVS_HIP(hipGetDevice(&c->device));
        *out = c.release();
    });
}

extern "C" void msvs_cache_free(msvs_cache_t * c)
{
    if (!c)
        return;
    for (msvs_block * b : c->lru)
        delete b;
    delete c;
}

extern "C" int msvs_block_lookup(msvs_cache_t * c, const char * key, uint64_t mark, msvs_block_t ** out)
{
    return guarded([&] {
        if (!c || !out)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null cache / out");
        *out = nullptr;
        std::lock_guard<std::mutex> lk(c->mu);
        auto it = c->map.find(msvs_cache::full_key(key, mark));
        if (it == c->map.end())
        {
            c->misses++;
            return;
        }
        msvs_block * b = it->second;
        c->lru.splice(c->lru.begin(), c->lru, b->pos);
        b->pins++;
        c->hits++;
        *out = b;
    });
}

extern "C" int msvs_block_upload(msvs_cache_t * c, const char * key, uint64_t mark, const float * rows, size_t n, size_t d,
                                 int normalize, msvs_block_t ** out)
{
    return guarded([&] {
        if (!c || !out || (n && !rows) || d == 0 || d > 8192)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null cache / rows / out or bad dimension");
        if (n > 0xfffffff0ull)
            fail(MSVS_ERR_ID_RANGE, "block exceeds the u32 row range");
        *out = nullptr;
        const std::string fk = msvs_cache::full_key(key, mark);
        {
            std::lock_guard<std::mutex> lk(c->mu);
            auto it = c->map.find(fk);
            if (it != c->map.end()) // another thread was faster: the resident copy wins
            {
                msvs_block * b = it->second;
                if (b->n != n || b->d != d || b->normalized != (normalize ? 1 : 0))
                    fail(MSVS_ERR_INVALID_ARGUMENT, "block `%s` is resident with another shape", fk.c_str());
                c->lru.splice(c->lru.begin(), c->lru, b->pos);
                b->pins++;
                *out = b;
                return;
            }
        }
        std::unique_ptr<msvs_block> b(new msvs_block);
        b->owner = c;
        b->key = fk;
        b->n = n;
        b->d = d;
        b->ld = padded_dim(d);
        b->normalized = normalize ? 1 : 0;
        b->rows.alloc(std::max<size_t>(n, 1) * b->ld);
        hipStream_t stream = nullptr;
        upload_rows(b->rows.p, rows, n, (uint32_t)d, b->ld, MSVS_MEM_HOST, stream);
        if (normalize && n)
        {
            normalize_device_rows(b->rows.p, n, (uint32_t)d, b->ld, stream);
        }
        MSVS_HIP(hipStreamSynchronize(stream));
        std::lock_guard<std::mutex> lk(c->mu);
        auto it = c->map.find(fk);
        if (it != c->map.end())
        {
            msvs_block * e = it->second;
            c->lru.splice(c->lru.begin(), c->lru, e->pos);
            e->pins++;
            *out = e;
            return; // b is dropped
        }
        c->make_room_locked(b->bytes());
        c->lru.push_front(b.get());
        b->pos = c->lru.begin();
        b->pins = 1;
        c->used += b->bytes();
        c->map[fk] = b.get();
        *out = b.release();
    });
}

extern "C" int msvs_block_info(const msvs_block_t * b, size_t * n, size_t * d, int * normalized)
{
    return guarded([&] {
        if (!b)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null block");
        if (n)
            *n = b->n;
        if (d)
            *d = b->d;
        if (normalized)
            *normalized = b->normalized;
    });
}

extern "C" void msvs_block_release(msvs_block_t * b)
{
    if (!b)
        return;
    msvs_cache_t * c = b->owner;
    std::lock_guard<std::mutex> lk(c->mu);
    if (--b->pins == 0 && b->doomed)
        delete b;
}

extern "C" int msvs_cache_evict(msvs_cache_t * c, const char * key_prefix, size_t * evicted)
{
    return guarded([&] {
        if (!c)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null cache");
        const std::string pre = key_prefix ? key_prefix : "";
        size_t cnt = 0;
        std::lock_guard<std::mutex> lk(c->mu);
        for (auto it = c->lru.begin(); it != c->lru.end();)
        {
            msvs_block * b = *it;
            ++it;
            if (b->key.compare(0, pre.size(), pre) == 0)
            {
                c->drop_locked(b);
                cnt++;
            }
        }
        if (evicted)
            *evicted = cnt;
    });
}

extern "C" int msvs_cache_stats(msvs_cache_t * c, size_t * bytes, size_t * blocks, uint64_t * hits, uint64_t * misses,
                                uint64_t * evictions)
{
    return guarded([&] {
        if (!c)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null cache");
        std::lock_guard<std::mutex> lk(c->mu);
        if (bytes)
            *bytes = c->used;
        if (blocks)
            *blocks = c->map.size();
        if (hits)
            *hits = c->hits;
        if (misses)
            *misses = c->misses;
        if (evictions)
            *evictions = c->evictions;
    });
}

extern "C" int msvs_knn_resident(const msvs_block_t * b, const float * x, size_t k, size_t nx, int metric,
                                 const uint64_t * alive_bits, int64_t * ids, float * dis)
{
    return guarded([&] {
        if (!b)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null block");
        knn_host(x, nullptr, b->d, k, nx, b->n, metric, alive_bits, ids, dis, b->rows.p);
    });
}

