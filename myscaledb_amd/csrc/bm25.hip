// bm25.hip -- seam B of include/msvs.h: BM25 scoring of exported posting lists on the GPU.
#include <atomic>
#include <cmath>
#include <map>
#include <memory>
#include <mutex>

#include "bm25_kernels.hpp"
#include "bm25p_kernels.hpp"
#include "bm25r_kernels.hpp"
#include "bm25l_kernels.hpp"
#include "device_ops.hpp"

using namespace msvs;

struct msvs_postings
{
    DevBuf<int64_t> post_off;
    DevBuf<uint32_t> doc_ids, tfs;
    DevBuf<uint8_t> fieldnorm_ids; // [num_fields][num_docs]
    DevBuf<uint8_t> term_field;    // empty: every term belongs to field 0
    std::vector<int64_t> h_post_off;
    std::vector<uint8_t> h_term_field; // empty: field 0
    size_t num_terms = 0, num_docs = 0, num_postings = 0, num_fields = 1;
    int device = 0; // where the postings live: the host-pointer entries run there whichever thread calls (DeviceGuard)
    // resident alive bitmap (lightweight deletes of the part): swapped under `mu`, a search keeps its own reference
    mutable std::mutex mu;
    std::shared_ptr<DevBuf<uint64_t>> alive;
    size_t alive_nbits = 0;
    // score-ready records (bm25r_kernels.hpp): derived from the postings and ONE fieldnorm cache (= the corpus statistics of the
    // searches that use it); rebuilt when a search arrives with other statistics, published under `mu`, a search keeps its own
    // reference (a replaced set is released by hipFree, which waits for the device)
    struct RecSet
    {
        DevBuf<uint2> rec;
        std::vector<float> key; // the cache it was built for, [num_fields][256]
    };
    mutable std::shared_ptr<RecSet> recs;
    // skip table of the frequent terms (bm25_skip_build_kernel): first posting at or past every 8192nd document, relative to the
    // term's first -- the sub-range bounds of a batch start their search inside one such stretch instead of the whole list
    DevBuf<int32_t> skip_row;  // [num_terms]: row of the term, -1 = none
    DevBuf<uint32_t> skip_tab; // [rows][skip_n + 1]
    uint32_t skip_n = 0;       // stretches per term (0: no table)
};

namespace
{
/// tantivy FIELD_NORMS_TABLE (tantivy/src/fieldnorm/code.rs): Lucene SmallFloat byte4 decoding.
uint32_t fieldnorm_of_id(uint32_t b)
{
    if (b < 24)
        return b;
    uint32_t i = b - 24, bits = i & 7;
    int shift = (int)(i >> 3) - 1;
    uint64_t dec = shift < 0 ? bits : ((uint64_t)(bits | 8) << shift);
    uint64_t v = 24 + dec;
    return v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v;
}

/// msvs_bm25_stats: queries that went through the sample / emit path, and (device side) how many of them had to take the
/// exact fallback.
std::atomic<unsigned long long> g_bm25_queries{0};
unsigned long long * bm25_fail_counter()
{
    static std::map<int, unsigned long long *> per_device;
    static std::mutex mu;
    int dev = 0;
    MSVS_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    auto it = per_device.find(dev);
    if (it != per_device.end())
        return it->second;
    unsigned long long * p = nullptr;
    MSVS_HIP(hipMalloc(&p, 8));
    MSVS_HIP(hipMemset(p, 0, 8));
    per_device[dev] = p;
    return p;
}

/// Host staging of a batch's argument blob: a small ring of pinned buffers per (host thread, stream).  The copy to the
/// device is truly asynchronous from pinned memory; a slot is reused only after the event recorded behind its copy has
/// passed (four calls later: normally long ago).  hipLaunchHostFunc to free a heap blob cost ~0.3 ms per call.
struct PinnedRing
{
    void * buf[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t cap[4] = {0, 0, 0, 0};
    // a slot is free again when the device has written want[slot] to done[slot] (pinned): the first launch behind the copy does it
    // (Bm25Params::slot_done: the first scorer launch) -- round 5 recorded an event per batch: a barrier packet, ~10 us of idle device
    uint32_t * done = nullptr;
    uint32_t want[4] = {0, 0, 0, 0};
    uint32_t seq = 0;
    int next = 0;
    void * take(size_t bytes, int & slot, hipStream_t stream)
    {
        slot = next;
        next = (next + 1) & 3;
        if (!done)
        {
            MSVS_HIP(hipHostMalloc(reinterpret_cast<void **>(&done), 64, hipHostMallocCoherent));
            for (int i = 0; i < 4; i++)
                done[i] = 0;
        }
        if (want[slot])
        {
            for (uint64_t spins = 1; __atomic_load_n(&done[slot], __ATOMIC_ACQUIRE) != want[slot]; spins++)
            {
                __builtin_ia32_pause();
                if ((spins & 0xfff) == 0 && hipStreamQuery(stream) != hipErrorNotReady)
                    break; // the stream ran dry (or failed: the next call reports it): nothing can still read the slot
            }
            want[slot] = 0;
        }
        if (cap[slot] < bytes)
        {
            if (buf[slot])
                MSVS_HIP(hipHostFree(buf[slot]));
            buf[slot] = nullptr;
            cap[slot] = 0;
            MSVS_HIP(hipHostMalloc(&buf[slot], bytes + bytes / 2 + 4096, hipHostMallocDefault));
            cap[slot] = bytes + bytes / 2 + 4096;
        }
        return buf[slot];
    }
    /// The sequence number the launch behind this slot's copy must write to done[slot].
    uint32_t arm(int slot)
    {
        seq = seq + 1 ? seq + 1 : 1;
        want[slot] = seq;
        return seq;
    }
};
PinnedRing & pinned_ring(hipStream_t stream)
{
    static thread_local std::map<std::pair<int, hipStream_t>, PinnedRing> rings;
    int dev = 0;
    MSVS_HIP(hipGetDevice(&dev));
    return rings[{dev, stream}];
}

/// Resident workgroups of the wave scorer per CU: 8 B of LDS per document (+ 1 per text column) and wavefront.
uint32_t bw_blocks_per_cu(bool one_field)
{
    const size_t per_block = (size_t)BW_WAVES * BW_DOCS * (8 + (one_field ? 1 : 4)) + 4096;
    return (uint32_t)std::max<size_t>(1, (160 * 1024) / per_block);
}

uint32_t bm25_cu_count()
{
    int dev = 0, n = 0;
    MSVS_HIP(hipGetDevice(&dev));
    MSVS_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    return (uint32_t)std::max(n, 1);
}

__global__ void and_words_kernel(const uint64_t * a, size_t na, const uint64_t * b, size_t nb, uint64_t * out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = (i < na ? a[i] : 0ull) & (i < nb ? b[i] : 0ull);
}
}

extern "C" int msvs_postings_create_fields(const int64_t * post_off, size_t num_terms, const uint8_t * term_field,
                                           const uint32_t * doc_ids, const uint32_t * tfs, const uint8_t * fieldnorm_ids,
                                           size_t num_fields, size_t num_docs, msvs_postings_t ** out)
{
    return guarded([&] {
        if (!out || !post_off || (num_docs && !fieldnorm_ids))
            fail(MSVS_ERR_INVALID_ARGUMENT, "null buffer");
        *out = nullptr;
        if (num_fields < 1 || num_fields > BM25_MAX_FIELDS)
            fail(MSVS_ERR_INVALID_ARGUMENT, "1 .. %u fields", BM25_MAX_FIELDS);
        const size_t np = (size_t)post_off[num_terms];
        if (np && (!doc_ids || !tfs))
            fail(MSVS_ERR_INVALID_ARGUMENT, "null postings");
        if (num_docs > 0xfffffff0ull)
            fail(MSVS_ERR_ID_RANGE, "num_docs exceeds the u32 row id range");
        for (size_t t = 0; t < num_terms; t++)
        {
            if (post_off[t + 1] < post_off[t] || (term_field && term_field[t] >= num_fields))
                fail(MSVS_ERR_INVALID_ARGUMENT, "posting offsets must ascend and term fields must be < num_fields");
            // the scorer indexes LDS by doc id and relies on one posting per (term, doc): never trust the export blindly
            for (int64_t p = post_off[t]; p < post_off[t + 1]; p++)
                if (doc_ids[p] >= num_docs || (p > post_off[t] && doc_ids[p] <= doc_ids[p - 1]))
                    fail(MSVS_ERR_INVALID_ARGUMENT, "doc ids must ascend inside a posting list and stay below num_docs");
        }
        std::unique_ptr<msvs_postings> p(new msvs_postings);
        MSVS_HIP(hipGetDevice(&p->device));
        p->num_terms = num_terms;
        p->num_docs = num_docs;
        p->num_postings = np;
        p->num_fields = num_fields;
        p->h_post_off.assign(post_off, post_off + num_terms + 1);
        p->post_off.alloc(num_terms + 1);
        p->doc_ids.alloc(std::max<size_t>(np, 1));
        p->tfs.alloc(std::max<size_t>(np, 1));
        p->fieldnorm_ids.alloc(std::max<size_t>(num_docs * num_fields, 1));
        MSVS_HIP(hipMemcpy(p->post_off.p, post_off, (num_terms + 1) * 8, hipMemcpyHostToDevice));
        if (np)
        {
            MSVS_HIP(hipMemcpy(p->doc_ids.p, doc_ids, np * 4, hipMemcpyHostToDevice));
            MSVS_HIP(hipMemcpy(p->tfs.p, tfs, np * 4, hipMemcpyHostToDevice));
        }
        if (num_docs)
            MSVS_HIP(hipMemcpy(p->fieldnorm_ids.p, fieldnorm_ids, num_docs * num_fields, hipMemcpyHostToDevice));
        if (term_field && num_terms)
        {
            p->h_term_field.assign(term_field, term_field + num_terms);
            p->term_field.alloc(num_terms);
            MSVS_HIP(hipMemcpy(p->term_field.p, term_field, num_terms, hipMemcpyHostToDevice));
        }
        // the skip table: terms with at least a posting per stretch, the most frequent first, at most 1/8 of the postings' bytes
        if (np && num_docs > BM25_SKIP_DOCS && options().bm25_skip != 0)
        {
            const uint32_t n_c = (uint32_t)ceil_div(num_docs, (size_t)BM25_SKIP_DOCS);
            std::vector<uint32_t> sel;
            for (size_t t = 0; t < num_terms; t++)
                if ((uint64_t)(post_off[t + 1] - post_off[t]) >= (uint64_t)n_c) // (a posting per stretch on average; the byte cap below decides the rest)
                    sel.push_back((uint32_t)t);
            std::sort(sel.begin(), sel.end(), [&](uint32_t x, uint32_t y) { return post_off[x + 1] - post_off[x] > post_off[y + 1] - post_off[y]; });
            const size_t max_rows = std::max<size_t>(1, np * 8 / 8 / ((size_t)(n_c + 1) * 4));
            if (sel.size() > max_rows)
                sel.resize(max_rows);
            if (!sel.empty())
            {
                std::vector<int32_t> row(num_terms, -1);
                for (size_t i = 0; i < sel.size(); i++)
                    row[sel[i]] = (int32_t)i;
                p->skip_row.alloc(num_terms);
                p->skip_tab.alloc(sel.size() * (size_t)(n_c + 1));
                DevBuf<uint32_t> d_sel(sel.size());
                MSVS_HIP(hipMemcpy(p->skip_row.p, row.data(), num_terms * 4, hipMemcpyHostToDevice));
                MSVS_HIP(hipMemcpy(d_sel.p, sel.data(), sel.size() * 4, hipMemcpyHostToDevice));
                const size_t n_e = sel.size() * (size_t)(n_c + 1);
                hipLaunchKernelGGL(bm25_skip_build_kernel, dim3((unsigned)ceil_div(n_e, (size_t)256)), dim3(256), 0, nullptr, p->post_off.p,
                                   p->doc_ids.p, d_sel.p, (uint32_t)sel.size(), n_c, p->skip_tab.p);
                MSVS_HIP(hipGetLastError());
                MSVS_HIP(hipDeviceSynchronize());
                p->skip_n = n_c;
            }
        }
        *out = p.release();
    });
}

extern "C" int msvs_postings_create(const int64_t * post_off, size_t num_terms, const uint32_t * doc_ids,
                                    const uint32_t * tfs, const uint8_t * fieldnorm_ids, size_t num_docs,
                                    msvs_postings_t ** out)
{
    return msvs_postings_create_fields(post_off, num_terms, nullptr, doc_ids, tfs, fieldnorm_ids, 1, num_docs, out);
}

extern "C" void msvs_postings_free(msvs_postings_t * postings) { delete postings; }

extern "C" int msvs_postings_set_alive(msvs_postings_t * ps, const uint64_t * alive_bits, size_t nbits)
{
    return guarded([&] {
        if (!ps)
            fail(MSVS_ERR_INVALID_ARGUMENT, "null postings");
        DeviceGuard on_device(ps->device);
        std::shared_ptr<DevBuf<uint64_t>> next;
        if (alive_bits)
        {
            const size_t words = std::max<size_t>(1, ceil_div(nbits, (size_t)64));
            next = std::make_shared<DevBuf<uint64_t>>(words);
            MSVS_HIP(hipMemset(next->p, 0, words * 8));
            if (nbits)
                MSVS_HIP(hipMemcpy(next->p, alive_bits, ceil_div(nbits, (size_t)64) * 8, hipMemcpyHostToDevice));
        }
        std::lock_guard<std::mutex> lk(ps->mu);
        ps->alive = next;
        ps->alive_nbits = alive_bits ? nbits : 0;
    });
}

namespace
{
/// The posting set's score-ready records for this fieldnorm cache (built on `stream` when the cached set is for other
/// statistics; the build is synchronised before it is published: other host threads may pick it up at once).
std::shared_ptr<msvs_postings::RecSet> records_for(const msvs_postings & ps, const std::vector<float> & cache, hipStream_t stream)
{
    {
        std::lock_guard<std::mutex> lk(ps.mu);
        if (ps.recs && ps.recs->key == cache)
            return ps.recs;
    }
    auto set = std::make_shared<msvs_postings::RecSet>();
    set->key = cache;
    // (one record of padding at either end: bm25l_kernel's dead lanes may read the neighbour of a slice, never use it)
    set->rec.alloc(ps.num_postings + 2);
    MSVS_HIP(hipMemsetAsync(set->rec.p, 0, (ps.num_postings + 2) * sizeof(uint2), stream));
    if (ps.num_postings)
    {
        DevBuf<float> d_cache(cache.size());
        MSVS_HIP(hipMemcpyAsync(d_cache.p, cache.data(), cache.size() * 4, hipMemcpyHostToDevice, stream));
        const unsigned grid = (unsigned)std::min<size_t>(ceil_div(ps.num_postings, (size_t)256), (size_t)bm25_cu_count() * 32);
        hipLaunchKernelGGL(bm25_rec_build_kernel, dim3(grid), dim3(256), 0, stream, ps.doc_ids.p, ps.tfs.p, ps.fieldnorm_ids.p,
                           ps.post_off.p, ps.h_term_field.empty() ? (const uint8_t *)nullptr : ps.term_field.p, (uint32_t)ps.num_terms,
                           (uint32_t)ps.num_docs, d_cache.p, set->rec.p + 1, (uint64_t)ps.num_postings);
        MSVS_HIP(hipGetLastError());
        MSVS_HIP(hipStreamSynchronize(stream)); // (d_cache is freed at scope exit; the set is complete when published)
    }
    std::lock_guard<std::mutex> lk(ps.mu);
    ps.recs = set;
    return set;
}

static __global__ void bm25_nop_kernel() {}

/// Documents per sub-range for a batch whose densest query has `rho` postings per document: that query's windows are single
/// sub-ranges ~3/4 full.  From 2048 on a multiple of the skip table's stretch (BM25_SKIP_DOCS): the bounds of the frequent terms are
/// then table entries (bm25_bounds8_kernel reads no posting for them).
static uint32_t bm25_sub_docs_for(double rho)
{
    uint32_t s = (uint32_t)std::min<double>(BP_MAX_DOCS, std::max<double>(BP_MIN_DOCS, std::floor(0.75 * BP_CAP / std::max(rho, 1e-9))));
    if (s >= BM25_SKIP_DOCS)
        s = s / BM25_SKIP_DOCS * BM25_SKIP_DOCS;
    return s;
}

/// Resident workgroups of bm25r_kernel per CU by its LDS (8192 hash slots: 38 KB per workgroup; 16384: 46 KB).
uint32_t br_blocks_per_cu(bool big_slots) { return big_slots ? 3u : 4u; }

/// One pass over a chunk of the batch.  Host inputs (term ids, df, statistics: a few numbers per query -- the weights
/// need libm's logf to match tantivy), device work on `stream`, results left in d_ids / d_scores ([nq][k], id -1 = no
/// hit).  d_alive: the effective filter, already on the device.
void bm25_chunk_device(const msvs_postings & ps, size_t nq, const uint32_t * qoff, const uint32_t * qterms,
                       const uint32_t * qgroups, const uint64_t * df, uint64_t total_docs, const float * cache,
                       int operator_or, const uint64_t * d_alive, size_t nbits, size_t k, int64_t * d_ids, float * d_scores,
                       const uint2 * d_rec, hipStream_t stream)
{
    const size_t f0 = qoff[0], n_flat = qoff[nq] - f0, nf1 = std::max<size_t>(n_flat, 1), nc = ps.num_fields * 256;
    const float K1 = 1.2f;
    // two kernels share this flow: the wave-private streaming scorer (default) and the block scorer it replaced (knob)
    const bool wave = options().bm25_wave != 0;
    // ... and, for batches of sparse terms, the posting-as-unit scorer (bm25p_kernels.hpp).  Its sub-range size follows the
    // densest query of the batch (postings of all its terms per document): that query's windows (<= BP_CAP postings) are
    // single sub-ranges, sparser queries take several per window.  Frequent terms (over 1/8 posting per document) keep the dense accumulator,
    // which they fill.
    bool posting = wave && options().bm25_posting != 0;
    static thread_local std::vector<uint64_t> q_postings;
    uint64_t all_postings = 0;
    uint32_t sub_docs = BW_DOCS;
    if (posting)
    {
        double rho = 0;
        q_postings.assign(nq, 0);
        for (size_t q = 0; q < nq; q++)
        {
            uint64_t sum = 0;
            for (size_t j = qoff[q]; j < qoff[q + 1] && qoff[q + 1] >= qoff[q]; j++)
            {
                if (qterms[j] >= ps.num_terms)
                    fail(MSVS_ERR_INVALID_ARGUMENT, "query term id %u out of range", qterms[j]);
                sum += (uint64_t)(ps.h_post_off[qterms[j] + 1] - ps.h_post_off[qterms[j]]);
            }
            q_postings[q] = sum;
            all_postings += sum;
            rho = std::max(rho, (double)sum / (double)std::max<size_t>(ps.num_docs, 1));
        }
        if (rho > 0.125 && options().bm25_posting != 2) // 2: always (tests: every sub-range of a frequent term is split)
            posting = false;
        else
        {
            // the densest query's windows are single sub-ranges filled to ~3/4 of the cap; sparser queries take several
            sub_docs = bm25_sub_docs_for(rho);
            if (options().bm25_sub_docs >= 16)
                sub_docs = (uint32_t)options().bm25_sub_docs;
        }
    }
    const uint32_t docs_per_block = posting ? sub_docs : (wave ? BW_DOCS : BM25_DOCS);
    const uint32_t n_blocks = (uint32_t)std::max<size_t>(1, ceil_div(ps.num_docs, (size_t)docs_per_block));
    // wave scorer: an item = spi consecutive sub-ranges of one query; enough items for ~4 per resident wavefront
    // (posting scorer: at least 32 chunks whenever there are 32 sub-ranges -- the sample / cut / emit flow below needs them, and a batch
    // whose densest query is sparse (8192-document sub-ranges, 1221 of them over 10M documents) must not fall off it: measured 2.2 ms
    // of per-chunk lists for a 1024-query batch against 0.95 ms)
    const uint32_t spi = (uint32_t)std::min<size_t>(std::min<size_t>(64, posting ? std::max<size_t>(1, n_blocks / 32) : 64),
                                                    std::max<size_t>(1, (size_t)n_blocks * nq / 8192));
    const uint32_t n_chunks = wave ? (uint32_t)ceil_div((size_t)n_blocks, (size_t)spi) : n_blocks;
    // long corpora: sample -> cut -> emit -> select (+ exact fallback); short ones: per-chunk top-k lists, one merge
    const bool emit = n_chunks >= (wave ? 32u : 64u) && ps.num_docs >= 500000 && options().bm25_emit != 0;
    const uint32_t cand_cap = options().bm25_cand_cap > 0 ? (uint32_t)std::min<double>(options().bm25_cand_cap, BM25_CAND_CAP) : BM25_CAND_CAP;
    unsigned long long * stat_fail = bm25_fail_counter();
    // posting scorer, EMIT pass: items of about equal postings (a uniform spread of a term over the documents assumed), twice
    // as many as resident wavefronts -- the launch walks them with a static stride
    const uint32_t cus = bm25_cu_count();
    // the record scorer (bm25r_kernel) when the posting set has its score-ready records for this call's statistics
    const bool recs = posting && d_rec != nullptr;
    const bool big_slots = options().bm25_slots >= 16384;
    // ... and its lean form (bm25l_kernels.hpp) when no query of the chunk has more than four terms
    bool lean = recs && options().bm25_lean != 0;
    for (size_t q = 0; q < nq && lean; q++)
        lean = qoff[q + 1] < qoff[q] || qoff[q + 1] - qoff[q] <= BL_NT;
    const uint32_t bpc = lean ? BL_WAVES_PER_SIMD : recs ? br_blocks_per_cu(big_slots) : BP_BLOCKS_PER_CU; // resident workgroups per CU of the posting scorer
    // items per resident wavefront of the EMIT pass: an item costs ~7 us of dependent loads before its first window, a wavefront with
    // one long item cannot even out the others' tails -- measured: 1 item at 64 queries (0.163 -> 0.146 ms), 2 at 256, 4 at 1024
    // (0.749 -> 0.698 ms): ~4600 postings per item, between 1 and 4 items
    const double ipw = options().bm25_items_per_wave > 0
        ? options().bm25_items_per_wave
        : std::min(4.0, std::max(1.0, (double)all_postings / (4600.0 * cus * (bpc * BP_WAVES))));
    const double per_item = std::max(800.0, (double)all_postings / (ipw * cus * (bpc * BP_WAVES)));
    auto spi_of = [&](size_t q) -> uint32_t {
        if (q_postings[q] == 0)
            return n_blocks;
        return (uint32_t)std::min<double>(n_blocks, std::max(1.0, std::floor(per_item * n_blocks / (double)q_postings[q])));
    };
    size_t n_items_e = 0;
    if (posting && emit)
        for (size_t q = 0; q < nq; q++)
            n_items_e += ceil_div((size_t)n_blocks, (size_t)spi_of(q));
    // the sample (every 16th chunk) walks a FINER partition in the wave scorer: with the emit pass's chunks (38 sub-ranges at
    // 64 queries over 10M documents) it was 576 items of 112 us each on 2048 resident wavefronts = one item's duration
    // (posting scorer: no finer than one item per resident wavefront -- a sample item costs ~30 us whatever it holds, and 5248
    // items on 4096 wavefront slots were two rounds of them)
    const uint32_t spi_s = wave && options().bm25_fine_sample != 0
        ? std::max<uint32_t>(std::max<uint32_t>(1, spi / 8),
                             posting ? (uint32_t)ceil_div((size_t)n_blocks * nq, (size_t)BM25_SAMPLE_STEP * cus * (bpc * BP_WAVES)) : 1u)
        : spi;
    const uint32_t n_chunks_s = wave ? (uint32_t)ceil_div((size_t)n_blocks, (size_t)spi_s) : n_blocks;
    uint32_t n_sb = (uint32_t)ceil_div((size_t)n_chunks_s, (size_t)BM25_SAMPLE_STEP);
    // posting scorer: the sample's items hold equal postings too (every 16th chunk of a PER-QUERY chunking: one window each,
    // one item per resident wavefront at most) -- with the same chunks for every query the launch lasted as long as the densest
    // query's items, 50 us
    const double per_item_s = std::max(384.0, (double)all_postings / BM25_SAMPLE_STEP / (cus * (double)(bpc * BP_WAVES)));
    // ... at least 64 chunks per query whatever its terms: the sample must stay 1 / 16 of the documents (a rare term whose one
    // chunk is the whole corpus would make the cut the m-th best of ALL documents: m < k pass, the query takes the fallback)
    const uint32_t spi_s_cap = std::max<uint32_t>(1, n_blocks / 64);
    auto spi_s_of = [&](size_t q) -> uint32_t {
        if (q_postings[q] == 0)
            return spi_s_cap;
        return (uint32_t)std::min<double>(spi_s_cap, std::max(1.0, std::floor(per_item_s * n_blocks / (double)q_postings[q])));
    };
    size_t n_items_s = 0;
    if (posting && emit)
    {
        uint32_t lists_s = 1;
        for (size_t q = 0; q < nq; q++)
        {
            const size_t ns = ceil_div(ceil_div((size_t)n_blocks, (size_t)spi_s_of(q)), (size_t)BM25_SAMPLE_STEP);
            n_items_s += ns;
            lists_s = std::max<uint32_t>(lists_s, (uint32_t)ns);
        }
        n_sb = lists_s; // lists per query in the sample buffer (unused ones stay KEY_NONE)
    }
    // m-th best of the sample as the cut: about STEP * m documents pass, 4 sigma (STEP * sqrt(m)) above k
    const double rs = 2.0 + std::sqrt(4.0 + (double)k / BM25_SAMPLE_STEP);
    const uint32_t cut_m = (uint32_t)std::min<double>(64.0, std::ceil(rs * rs));
    // one host blob -> one copy: [qoff u32][qterms u32][weight f32][cache f32][full u16][group u8][field u8][emit items 4 x u32][sample items 4 x u32]
    const size_t o_qoff = 0, o_terms = o_qoff + (nq + 1) * 4, o_w = o_terms + nf1 * 4, o_cache = o_w + nf1 * 4,
                 o_full = o_cache + nc * 4, o_group = o_full + round_up(nq * 2, (size_t)4), o_field = o_group + nf1,
                 o_items = round_up(o_field + nf1, (size_t)16), o_items_s = o_items + n_items_e * 16,
                 blob_bytes = round_up(o_items_s + n_items_s * 16, (size_t)16);
    PinnedRing & ring = pinned_ring(stream);
    int slot = 0;
    // (the blob is put together in ordinary memory and copied into the pinned slot in one go: stores into the pinned mapping cost
    // ~50 ns each -- 4.3 ms of host time per 4096-query batch went into writing 16 k items word by word)
    static thread_local std::vector<unsigned char> blob_heap;
    blob_heap.resize(blob_bytes);
    unsigned char * const blob = blob_heap.data();
    unsigned char * const blob_pinned = static_cast<unsigned char *>(ring.take(blob_bytes, slot, stream));
    uint32_t * h_qoff = reinterpret_cast<uint32_t *>(blob + o_qoff);
    uint32_t * h_terms = reinterpret_cast<uint32_t *>(blob + o_terms);
    float * weight = reinterpret_cast<float *>(blob + o_w);
    uint16_t * full = reinterpret_cast<uint16_t *>(blob + o_full);
    uint8_t * group = blob + o_group;
    uint8_t * field = blob + o_field;
    memcpy(blob + o_cache, cache, nc * 4);
    h_qoff[0] = 0;
    for (size_t q = 0; q < nq; q++)
    {
        if (qoff[q + 1] < qoff[q] || qoff[q + 1] - qoff[q] > BM25_MAX_TERMS)
            fail(MSVS_ERR_INVALID_ARGUMENT, "a query has more than %u terms (or offsets descend)", BM25_MAX_TERMS);
        h_qoff[q + 1] = (uint32_t)(qoff[q + 1] - f0);
        uint16_t m = 0;
        for (size_t j = qoff[q]; j < qoff[q + 1]; j++)
        {
            if (qterms[j] >= ps.num_terms)
                fail(MSVS_ERR_INVALID_ARGUMENT, "query term id %u out of range", qterms[j]);
            if (df[j] > total_docs)
                fail(MSVS_ERR_INVALID_ARGUMENT, "doc_freq exceeds total_docs");
            const uint32_t g = qgroups ? qgroups[j] : (uint32_t)(j - qoff[q]);
            if (g >= BM25_MAX_GROUPS && !operator_or)
                fail(MSVS_ERR_INVALID_ARGUMENT, "AND queries take at most %u tokens", BM25_MAX_GROUPS);
            group[j - f0] = (uint8_t)(g % BM25_MAX_GROUPS);
            m |= (uint16_t)(1u << group[j - f0]);
            h_terms[j - f0] = qterms[j];
            field[j - f0] = ps.h_term_field.empty() ? (uint8_t)0 : ps.h_term_field[qterms[j]];
            // tantivy Bm25Weight (bm25.rs): idf in f32
            volatile float x = ((float)(total_docs - df[j]) + 0.5f) / ((float)df[j] + 0.5f);
            volatile float idf = logf(1.0f + x);
            weight[j - f0] = idf * (1.0f + K1);
        }
        full[q] = m;
    }
    {
        uint32_t * it = reinterpret_cast<uint32_t *>(blob + o_items);
        if (n_items_e)
            for (size_t q = 0; q < nq; q++)
            {
                const uint32_t sq = spi_of(q);
                for (uint32_t b = 0; b < n_blocks; b += sq, it += 4)
                {
                    it[0] = (uint32_t)q;
                    it[1] = b;
                    it[2] = std::min<uint32_t>(n_blocks, b + sq);
                    it[3] = 0;
                }
            }
        it = reinterpret_cast<uint32_t *>(blob + o_items_s);
        if (n_items_s)
            for (size_t q = 0; q < nq; q++)
            {
                const uint32_t sq = spi_s_of(q);
                uint32_t li = 0;
                for (uint64_t b = 0; b < n_blocks; b += (uint64_t)sq * BM25_SAMPLE_STEP, it += 4, li++)
                {
                    it[0] = (uint32_t)q;
                    it[1] = (uint32_t)b;
                    it[2] = (uint32_t)std::min<uint64_t>(n_blocks, b + sq);
                    it[3] = li;
                }
            }
    }
    Scratch & scr = scratch_for(stream);
    scr.reserve(nq * (size_t)n_chunks * k * 8 + nf1 * (size_t)(n_blocks + 1) * 16 + blob_bytes
                    + (emit ? nq * ((size_t)(n_sb + 1) * cut_m * 8 + (size_t)BM25_CAND_CAP * 8 + 16) : 0) + nq * 8 + 256 + 65536,
                stream);
    Bm25Params a{};
    unsigned char * d_blob = scr.take<unsigned char>(blob_bytes);
    int64_t * d_bounds = scr.take<int64_t>(nf1 * (n_blocks + 1));
    int64_t * d_bounds_hi = scr.take<int64_t>(nf1 * (n_blocks + 1));
    uint64_t * partial = scr.take<uint64_t>(nq * (size_t)n_chunks * k);
    memcpy(blob_pinned, blob, blob_bytes);
    // the tables go to the device with the bounds launch (bm25_bounds8_kernel) when there is one, else with a copy kernel of their own;
    // the first scorer launch behind them tells the host that the pinned slot is free again (Bm25Params::slot_done)
    const bool tables_ride = n_flat != 0 && options().bm25_bounds8 != 0 && options().bm25_tables_ride != 0;
    if (!tables_ride)
        fetch_from_pinned(d_blob, blob_pinned, blob_bytes, stream);
    a.slot_done = ring.done + slot;
    a.slot_seq = ring.arm(slot);
    a.post_off = ps.post_off.p;
    a.doc_ids = ps.doc_ids.p;
    a.tfs = ps.tfs.p;
    a.fieldnorm_ids = ps.fieldnorm_ids.p;
    a.term_field = ps.term_field.p;
    a.alive = d_alive;
    a.nbits = (uint32_t)std::min<size_t>(nbits, 0xffffffffu);
    a.num_docs = (uint32_t)ps.num_docs;
    a.num_fields = (uint32_t)ps.num_fields;
    a.n_blocks = n_blocks;
    a.last_posting = ps.num_postings ? ps.num_postings - 1 : 0;
    a.nq = (uint32_t)nq;
    a.n_flat = (uint32_t)nf1;
    a.operator_or = operator_or;
    a.qoff = reinterpret_cast<uint32_t *>(d_blob + o_qoff);
    a.qterms = reinterpret_cast<uint32_t *>(d_blob + o_terms);
    a.weight = reinterpret_cast<float *>(d_blob + o_w);
    a.norm_cache = reinterpret_cast<float *>(d_blob + o_cache);
    a.qfull = reinterpret_cast<uint16_t *>(d_blob + o_full);
    a.qgroup = d_blob + o_group;
    a.qfield = d_blob + o_field;
    a.bounds = d_bounds;
    a.bounds_hi = d_bounds_hi;
    // TOPK over (a sample of) the chunks: lists of p.kk keys into p.partial, `lists` per slot
    auto launch_topk = [&](Bm25Params p, uint32_t lists, uint32_t step, size_t slots_bound, uint32_t item_spi, uint32_t item_chunks,
                           const uint32_t * items = nullptr, size_t n_items_tab = 0) {
        if (wave)
        {
            Bm25WParams w{};
            w.p = p;
            w.spi = item_spi;
            w.n_chunks = item_chunks;
            w.cstep = step;
            w.n_items_c = lists;
            w.lists = lists;
            w.sub_docs = docs_per_block;
            w.dbg = (uint32_t)options().bm25_dbg;
            const bool nf1k = ps.num_fields == 1;
            const int r = r_for_k(p.kk);
            if (posting)
            {
                w.items = items;
                w.n_items_tab = (uint32_t)n_items_tab;
                const size_t n_it = items ? n_items_tab : (size_t)lists * slots_bound;
                const unsigned pgrid = (unsigned)std::max<size_t>(1, std::min<size_t>(ceil_div(n_it, (size_t)BP_WAVES), (size_t)cus * bpc));
                if (recs)
                {
                    Bm25RParams rp{w, d_rec};
#define MSVS_BM25R(RR) \
    do \
    { \
        if (lean) \
            hipLaunchKernelGGL((bm25l_kernel<BM25_TOPK, RR, 8192>), dim3(pgrid), dim3(64 * BP_WAVES), 0, stream, rp); \
        else if (big_slots) \
            hipLaunchKernelGGL((bm25r_kernel<BM25_TOPK, RR, 16384>), dim3(pgrid), dim3(64 * BP_WAVES), 0, stream, rp); \
        else \
            hipLaunchKernelGGL((bm25r_kernel<BM25_TOPK, RR, 8192>), dim3(pgrid), dim3(64 * BP_WAVES), 0, stream, rp); \
    } while (0)
                    if (r == 1) MSVS_BM25R(1); else if (r == 2) MSVS_BM25R(2); else MSVS_BM25R(4);
#undef MSVS_BM25R
                    MSVS_HIP(hipGetLastError());
                    return;
                }
                if (r == 1)
                    hipLaunchKernelGGL((bm25p_kernel<BM25_TOPK, 1>), dim3(pgrid), dim3(64 * BP_WAVES), 0, stream, w);
                else if (r == 2)
                    hipLaunchKernelGGL((bm25p_kernel<BM25_TOPK, 2>), dim3(pgrid), dim3(64 * BP_WAVES), 0, stream, w);
                else
                    hipLaunchKernelGGL((bm25p_kernel<BM25_TOPK, 4>), dim3(pgrid), dim3(64 * BP_WAVES), 0, stream, w);
                MSVS_HIP(hipGetLastError());
                return;
            }
            const unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>(ceil_div((size_t)lists * slots_bound, (size_t)BW_WAVES), (size_t)cus * bw_blocks_per_cu(nf1k)));
#define MSVS_BM25W(RR, NF) hipLaunchKernelGGL((bm25w_kernel<BM25_TOPK, RR, NF>), dim3(grid), dim3(64 * BW_WAVES), 0, stream, w)
            if (nf1k)
            {
                if (r == 1) MSVS_BM25W(1, 1); else if (r == 2) MSVS_BM25W(2, 1); else MSVS_BM25W(4, 1);
            }
            else
            {
                if (r == 1) MSVS_BM25W(1, 4); else if (r == 2) MSVS_BM25W(2, 4); else MSVS_BM25W(4, 4);
            }
#undef MSVS_BM25W
        }
        else
        {
            p.n_pad = lists;
            p.bstep = step;
            const size_t lds = (size_t)5 * p.kk * 8;
            const dim3 grid(lists, (uint32_t)std::min<size_t>(slots_bound, std::max<size_t>(1, 2048 / lists)));
            switch (r_for_k(p.kk))
            {
                case 1:
                    hipLaunchKernelGGL((bm25_score_kernel<BM25_TOPK, 1>), grid, dim3(BLOCK), lds, stream, p);
                    break;
                case 2:
                    hipLaunchKernelGGL((bm25_score_kernel<BM25_TOPK, 2>), grid, dim3(BLOCK), lds, stream, p);
                    break;
                default:
                    hipLaunchKernelGGL((bm25_score_kernel<BM25_TOPK, 4>), grid, dim3(BLOCK), lds, stream, p);
                    break;
            }
        }
        MSVS_HIP(hipGetLastError());
    };
    ProfileScope prof("bm25_score", stream);
    // the buffers of the sample / emit path, taken here so that the bounds launch can fill them on its way
    uint64_t * sample = emit ? scr.take<uint64_t>(nq * (size_t)n_sb * cut_m) : nullptr;
    uint64_t * cut_keys = emit ? scr.take<uint64_t>(nq * (size_t)cut_m) : nullptr;
    uint64_t * cand = emit ? scr.take<uint64_t>(nq * (size_t)BM25_CAND_CAP) : nullptr;
    uint32_t * counters = emit ? scr.take<uint32_t>(2 * nq + 4) : nullptr; // ccnt[nq] | nfail | failq[nq]
    const bool fills_ride = emit && n_flat != 0;
    if (n_flat)
    {
        if ((uint32_t)options().bm25_dbg & 32u) // experiment: does the first launch of a batch wait for the blob copy?
            hipLaunchKernelGGL(bm25_nop_kernel, dim3(1), dim3(64), 0, stream);
        const dim3 bgrid((unsigned)ceil_div(n_flat * (size_t)(n_blocks + 1), (size_t)256));
        Bm25Params ab = a;
        if (tables_ride) // (the launch that copies the tables reads its own input, the flat terms, from the pinned slot)
            ab.qterms = reinterpret_cast<const uint32_t *>(blob_pinned + o_terms);
        if (options().bm25_bounds8 != 0)
            hipLaunchKernelGGL(bm25_bounds8_kernel,
                               dim3((unsigned)ceil_div((size_t)n_blocks + 1, (size_t)256), (unsigned)std::min<size_t>(n_flat, 65535)), dim3(256), 0, stream, ab, d_bounds,
                               recs ? (int64_t *)nullptr : d_bounds_hi /* only bm25p_kernel reads the shifted copy */, (uint32_t)n_flat, docs_per_block,
                               counters, fills_ride ? nq + 1 : (size_t)0, sample, fills_ride && n_items_s ? nq * (size_t)n_sb * cut_m : (size_t)0,
                               ps.skip_n ? ps.skip_row.p : (const int32_t *)nullptr, ps.skip_tab.p, ps.skip_n,
                               tables_ride ? reinterpret_cast<bm25_u32x4 *>(d_blob) : (bm25_u32x4 *)nullptr,
                               reinterpret_cast<const bm25_u32x4 *>(blob_pinned), tables_ride ? blob_bytes / 16 : (size_t)0);
        else
            hipLaunchKernelGGL(bm25_bounds_kernel, bgrid, dim3(256), 0, stream, a, d_bounds, d_bounds_hi, (uint32_t)n_flat, docs_per_block,
                               counters, fills_ride ? nq + 1 : (size_t)0, sample, fills_ride && n_items_s ? nq * (size_t)n_sb * cut_m : (size_t)0);
    }
    if (!emit)
    {
        a.partial = partial;
        a.kk = (uint32_t)k;
        launch_topk(a, n_chunks, 1, nq, spi, n_chunks);
        MergeParams m{};
        m.partial = partial;
        m.n_lists = n_chunks;
        m.k = (uint32_t)k;
        m.out_ids = d_ids;
        m.out_dis = d_scores;
        launch_merge(M_IP, m, (uint32_t)nq, stream);
        return;
    }
    if (!fills_ride)
        MSVS_HIP(hipMemsetAsync(counters, 0, (nq + 1) * 4, stream));
    uint32_t * ccnt = counters, * nfail = counters + nq, * failq = counters + nq + 1;
    // 1. the sample: every 16th chunk, short lists
    Bm25Params sp = a;
    sp.partial = sample;
    sp.kk = cut_m;
    if (n_items_s)
    {
        if (!fills_ride)
            MSVS_HIP(hipMemsetAsync(sample, 0xFF, nq * (size_t)n_sb * cut_m * 8, stream)); // a query with fewer items leaves lists unused
        launch_topk(sp, n_sb, BM25_SAMPLE_STEP, nq, spi_s, n_chunks_s, reinterpret_cast<const uint32_t *>(d_blob + o_items_s), n_items_s);
    }
    else
        launch_topk(sp, n_sb, BM25_SAMPLE_STEP, nq, spi_s, n_chunks_s);
    const size_t n_sample_keys = (size_t)n_sb * cut_m;
    if (options().bm25_cutk != 0 && (n_sample_keys <= 4096 || (nq <= 512 && n_sample_keys <= 8192 && cut_m <= 64)))
    {
        // only the score of the m-th best sample key is ever read (entry cut_m - 1): one wavefront per query selects it
        const dim3 cgrid((unsigned)ceil_div(nq, (size_t)(BLOCK / 64)));
        if (nq <= 512 && n_sample_keys > 256 && cut_m <= 64) // few queries: a workgroup each
        {
            if (n_sample_keys <= 1024)
                hipLaunchKernelGGL((bm25_cut_block_kernel<4>), dim3((unsigned)nq), dim3(BLOCK), 0, stream, sample, (uint32_t)n_sample_keys, (uint32_t)nq, cut_m, cut_keys);
            else if (n_sample_keys <= 2048)
                hipLaunchKernelGGL((bm25_cut_block_kernel<8>), dim3((unsigned)nq), dim3(BLOCK), 0, stream, sample, (uint32_t)n_sample_keys, (uint32_t)nq, cut_m, cut_keys);
            else if (n_sample_keys <= 4096)
                hipLaunchKernelGGL((bm25_cut_block_kernel<16>), dim3((unsigned)nq), dim3(BLOCK), 0, stream, sample, (uint32_t)n_sample_keys, (uint32_t)nq, cut_m, cut_keys);
            else
                hipLaunchKernelGGL((bm25_cut_block_kernel<32>), dim3((unsigned)nq), dim3(BLOCK), 0, stream, sample, (uint32_t)n_sample_keys, (uint32_t)nq, cut_m, cut_keys);
        }
        else if (n_sample_keys <= 1024)
            hipLaunchKernelGGL((bm25_cut_kernel<16>), cgrid, dim3(BLOCK), 0, stream, sample, (uint32_t)n_sample_keys, (uint32_t)nq, cut_m, cut_keys);
        else if (n_sample_keys <= 2048)
            hipLaunchKernelGGL((bm25_cut_kernel<32>), cgrid, dim3(BLOCK), 0, stream, sample, (uint32_t)n_sample_keys, (uint32_t)nq, cut_m, cut_keys);
        else
            hipLaunchKernelGGL((bm25_cut_kernel<64>), cgrid, dim3(BLOCK), 0, stream, sample, (uint32_t)n_sample_keys, (uint32_t)nq, cut_m, cut_keys);
    }
    else
    {
        MergeParams m{};
        m.partial = sample;
        m.n_lists = n_sb;
        m.k = cut_m;
        m.mode = 2;
        m.out_keys = cut_keys;
        launch_merge(M_IP, m, (uint32_t)nq, stream);
    }
    // 2. every chunk: emit what passes the cut
    Bm25Params ep = a;
    ep.cut_keys = cut_keys;
    ep.cut_m = cut_m;
    ep.cand = cand;
    ep.ccnt = ccnt;
    ep.cand_cap = cand_cap;
    if (wave)
    {
        Bm25WParams w{};
        w.p = ep;
        w.spi = spi;
        w.n_chunks = n_chunks;
        w.cstep = 1;
        w.n_items_c = n_chunks;
        w.lists = n_chunks;
        w.sub_docs = docs_per_block;
        w.dbg = (uint32_t)options().bm25_dbg;
        w.items = posting ? reinterpret_cast<const uint32_t *>(d_blob + o_items) : nullptr;
        w.n_items_tab = (uint32_t)n_items_e;
        const bool nf1k = ps.num_fields == 1;
        const unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>(ceil_div((size_t)n_chunks * nq, (size_t)BW_WAVES), (size_t)cus * bw_blocks_per_cu(nf1k)));
        if (recs)
        {
            const dim3 egrid((unsigned)std::max<size_t>(1, std::min<size_t>(ceil_div(std::max<size_t>(n_items_e, 1), (size_t)BP_WAVES), (size_t)cus * bpc)));
            Bm25RParams rp{w, d_rec};
            if (lean)
                hipLaunchKernelGGL((bm25l_kernel<BM25_EMIT, 1, 8192>), egrid, dim3(64 * BP_WAVES), 0, stream, rp);
            else if (big_slots)
                hipLaunchKernelGGL((bm25r_kernel<BM25_EMIT, 1, 16384>), egrid, dim3(64 * BP_WAVES), 0, stream, rp);
            else
                hipLaunchKernelGGL((bm25r_kernel<BM25_EMIT, 1, 8192>), egrid, dim3(64 * BP_WAVES), 0, stream, rp);
        }
        else if (posting)
            hipLaunchKernelGGL((bm25p_kernel<BM25_EMIT, 1>), dim3((unsigned)std::max<size_t>(1, std::min<size_t>(ceil_div((size_t)n_chunks * nq, (size_t)BP_WAVES), (size_t)cus * BP_BLOCKS_PER_CU))),
                               dim3(64 * BP_WAVES), 0, stream, w);
        else if (nf1k)
            hipLaunchKernelGGL((bm25w_kernel<BM25_EMIT, 1, 1>), dim3(grid), dim3(64 * BW_WAVES), 0, stream, w);
        else
            hipLaunchKernelGGL((bm25w_kernel<BM25_EMIT, 1, 4>), dim3(grid), dim3(64 * BW_WAVES), 0, stream, w);
    }
    else
    {
        ep.bstep = 1;
        hipLaunchKernelGGL((bm25_score_kernel<BM25_EMIT, 1>), dim3(n_blocks, (uint32_t)std::min<size_t>(nq, ceil_div((size_t)4096, (size_t)n_blocks))),
                           dim3(BLOCK), 0, stream, ep);
    }
    // 3. select, or queue for the fallback
    const size_t lds_k = (size_t)5 * k * 8;
    if (options().bm25_select2 != 0 && k <= 256)
        hipLaunchKernelGGL(bm25_select2_kernel, dim3((unsigned)nq), dim3(BLOCK), 0, stream, cand, ccnt, cand_cap, cut_keys, cut_m, (uint32_t)k, d_ids,
                           d_scores, failq, nfail, stat_fail);
    else
        hipLaunchKernelGGL(bm25_select_kernel, dim3((unsigned)nq), dim3(BM25_SELECT_THREADS), 0, stream, cand, ccnt, cand_cap, cut_keys, cut_m,
                           (uint32_t)k, d_ids, d_scores, failq, nfail, stat_fail);
    // 4. the exact fallback over the queue (empty launches when nobody queued)
    Bm25Params fp = a;
    fp.partial = partial;
    fp.kk = (uint32_t)k;
    fp.qsel = failq;
    fp.nsel = nfail;
    launch_topk(fp, n_chunks, 1, std::min<size_t>(nq, wave ? nq : 4), spi, n_chunks);
    switch (r_for_k((uint32_t)k))
    {
        case 1:
            hipLaunchKernelGGL((bm25_fb_merge_kernel<1>), dim3((unsigned)nq), dim3(BLOCK), lds_k, stream, partial, n_chunks, n_chunks,
                               (uint32_t)k, failq, nfail, d_ids, d_scores);
            break;
        case 2:
            hipLaunchKernelGGL((bm25_fb_merge_kernel<2>), dim3((unsigned)nq), dim3(BLOCK), lds_k, stream, partial, n_chunks, n_chunks,
                               (uint32_t)k, failq, nfail, d_ids, d_scores);
            break;
        default:
            hipLaunchKernelGGL((bm25_fb_merge_kernel<4>), dim3((unsigned)nq), dim3(BLOCK), lds_k, stream, partial, n_chunks, n_chunks,
                               (uint32_t)k, failq, nfail, d_ids, d_scores);
            break;
    }
    MSVS_HIP(hipGetLastError());
    g_bm25_queries.fetch_add(nq, std::memory_order_relaxed);
}

/// The batched search: statistics -> fieldnorm caches, effective filter (resident bitmap of the part AND the per-call
/// one), then chunks of queries sized so that the per-block partial lists stay under 256 MB.
void bm25_batch_device(const msvs_postings & ps, size_t nq, const uint32_t * qoff, const uint32_t * qterms,
                       const uint32_t * qgroups, const uint64_t * df, uint64_t total_docs, const uint64_t * total_tokens,
                       int operator_or, const uint64_t * d_alive, size_t nbits, size_t k, int64_t * d_ids, float * d_scores,
                       hipStream_t stream)
{
    if (k > MSVS_MAX_K)
        fail(MSVS_ERR_UNSUPPORTED_K, "k = %zu exceeds the device top-k limit %d", k, MSVS_MAX_K);
    if (total_docs == 0)
        fail(MSVS_ERR_INVALID_ARGUMENT, "total_docs is zero");
    // tantivy Bm25Weight (bm25.rs): K1 = 1.2, B = 0.75, all f32
    const float K1 = 1.2f, B = 0.75f;
    std::vector<float> cache(ps.num_fields * 256);
    for (size_t f = 0; f < ps.num_fields; f++)
    {
        const float avg = (float)total_tokens[f] / (float)total_docs;
        for (uint32_t i = 0; i < 256; i++)
        {
            volatile float t0 = B * (float)fieldnorm_of_id(i);
            volatile float t1 = t0 / avg;
            volatile float t2 = (1.0f - B) + t1;
            cache[f * 256 + i] = K1 * t2;
        }
    }
    std::shared_ptr<DevBuf<uint64_t>> resident;
    size_t res_bits = 0;
    {
        std::lock_guard<std::mutex> lk(ps.mu);
        resident = ps.alive;
        res_bits = ps.alive_nbits;
    }
    const uint64_t * eff = d_alive;
    size_t eff_bits = nbits;
    if (resident)
    {
        if (d_alive)
        {
            const size_t fwords = ceil_div(ps.num_docs, (size_t)64) + 1;
            Scratch & aux = aux_for(stream);
            aux.reserve(fwords * 8 + 256, stream);
            uint64_t * both = aux.take<uint64_t>(fwords);
            hipLaunchKernelGGL(and_words_kernel, dim3((unsigned)ceil_div(fwords, (size_t)256)), dim3(256), 0, stream, d_alive,
                               ceil_div(nbits, (size_t)64), resident->p, ceil_div(res_bits, (size_t)64), both, fwords);
            eff = both;
            eff_bits = std::min(nbits, res_bits);
        }
        else
        {
            eff = resident->p;
            eff_bits = res_bits;
        }
    }
    // score-ready records of the posting scorer (bm25r_kernels.hpp): derived per fieldnorm cache, built on first use
    std::shared_ptr<msvs_postings::RecSet> recset;
    if (options().bm25_wave != 0 && options().bm25_posting != 0 && options().bm25_rec != 0)
        recset = records_for(ps, cache, stream);
    // an upper bound of the sub-ranges per query term: the posting scorer's sub-range size follows the densest query (a chunk's
    // densest query is no denser than the batch's, so its sub-ranges are no shorter than this one)
    size_t sub_ub = BW_DOCS;
    if (options().bm25_wave != 0 && options().bm25_posting != 0)
    {
        double rho = 0;
        for (size_t q = 0; q < nq; q++)
        {
            uint64_t sum = 0;
            for (size_t j = qoff[q]; j < qoff[q + 1] && qoff[q + 1] >= qoff[q]; j++)
                if (qterms[j] < ps.num_terms)
                    sum += (uint64_t)(ps.h_post_off[qterms[j] + 1] - ps.h_post_off[qterms[j]]);
            rho = std::max(rho, (double)sum / (double)std::max<size_t>(ps.num_docs, 1));
        }
        const size_t formula = bm25_sub_docs_for(rho);
        const size_t sub_p = options().bm25_sub_docs >= 16 ? (size_t)options().bm25_sub_docs : formula;
        // no chunk can be denser than the batch: all of them take the posting scorer, or some may keep the dense accumulator
        sub_ub = rho <= 0.125 || options().bm25_posting == 2 ? sub_p : std::min<size_t>(BW_DOCS, sub_p);
    }
    const size_t n_blocks = std::max<size_t>(1, ceil_div(ps.num_docs, sub_ub));
    // per query: candidate slots + ~4 terms of sub-range bounds (the per-chunk lists are ~8192 x k keys for the whole batch); chunks
    // of queries sized so that this stays under 1 GB (round 5: 256 MB cut a 1024-query batch over 10M documents into 6 chunks of
    // seven launches each)
    const size_t per_q = (size_t)BM25_CAND_CAP * 8 + 4 * 16 * (n_blocks + 1) + 64 * k * 8 + 64
        + (options().bm25_wave != 0 ? 0 : ceil_div(ps.num_docs, (size_t)BM25_DOCS) * k * 8); // the block scorer: a list per block
    const size_t chunk = std::max<size_t>(1, std::min<size_t>(nq, ((size_t)1024 << 20) / per_q));
    for (size_t q0 = 0; q0 < nq; q0 += chunk)
    {
        const size_t nqc = std::min(chunk, nq - q0);
        bm25_chunk_device(ps, nqc, qoff + q0, qterms, qgroups, df, total_docs, cache.data(), operator_or, eff, eff_bits, k,
                          d_ids + q0 * k, d_scores + q0 * k, recset ? recset->rec.p + 1 : nullptr, stream);
    }
    // `resident` / `recset` are held until every kernel reading them is enqueued; a swapped-out buffer is released by hipFree,
    // which waits for the device
}
}

extern "C" int msvs_bm25_search_batch(const msvs_postings_t * ps, size_t nq, const uint32_t * qoff, const uint32_t * qterms,
                                      const uint32_t * qgroups, const uint64_t * df, uint64_t total_docs,
                                      const uint64_t * total_tokens, int operator_or, const uint64_t * alive_bits,
                                      size_t nbits, size_t k, uint64_t * row_ids, float * scores, uint32_t * n_out)
{
    return guarded([&] {
        if (!ps || (nq && (!qoff || !n_out || !total_tokens)) || (nq && qoff[nq] && (!qterms || !df))
            || (nq && k && (!row_ids || !scores)))
            fail(MSVS_ERR_INVALID_ARGUMENT, "null buffer");
        for (size_t q = 0; q < nq; q++)
            n_out[q] = 0;
        DeviceGuard on_device(ps->device);
        if (nq == 0 || k == 0 || ps->num_docs == 0)
            return;
        hipStream_t stream = thread_stream();
        Scratch & stg = staging_for(stream);
        const size_t words = alive_bits ? std::max<size_t>(1, ceil_div(nbits, (size_t)64)) : 0;
        stg.reserve(nq * k * 12 + words * 8 + 4096, stream);
        int64_t * d_ids = stg.take<int64_t>(nq * k);
        float * d_sc = stg.take<float>(nq * k);
        uint64_t * d_alive = words ? stg.take<uint64_t>(words) : nullptr;
        if (words)
        {
            MSVS_HIP(hipMemsetAsync(d_alive, 0, words * 8, stream));
            if (nbits)
                MSVS_HIP(hipMemcpyAsync(d_alive, alive_bits, ceil_div(nbits, (size_t)64) * 8, hipMemcpyHostToDevice, stream));
        }
        bm25_batch_device(*ps, nq, qoff, qterms, qgroups, df, total_docs, total_tokens, operator_or, d_alive, nbits, k, d_ids,
                          d_sc, stream);
        std::vector<int64_t> h_ids(nq * k);
        MSVS_HIP(hipMemcpyAsync(h_ids.data(), d_ids, nq * k * 8, hipMemcpyDeviceToHost, stream));
        MSVS_HIP(hipMemcpyAsync(scores, d_sc, nq * k * 4, hipMemcpyDeviceToHost, stream));
        MSVS_HIP(hipStreamSynchronize(stream));
        for (size_t q = 0; q < nq; q++)
        {
            uint32_t cnt = 0;
            while (cnt < k && h_ids[q * k + cnt] >= 0)
            {
                row_ids[q * k + cnt] = (uint64_t)h_ids[q * k + cnt];
                cnt++;
            }
            n_out[q] = cnt;
        }
    });
}

extern "C" int msvs_bm25_search_batch_device(const msvs_postings_t * ps, size_t nq, const uint32_t * qoff,
                                             const uint32_t * qterms, const uint32_t * qgroups, const uint64_t * df,
                                             uint64_t total_docs, const uint64_t * total_tokens, int operator_or,
                                             const uint64_t * d_alive_bits, size_t nbits, size_t k, int64_t * d_row_ids,
                                             float * d_scores, void * hip_stream)
{
    return guarded([&] {
        if (!ps || (nq && (!qoff || !total_tokens)) || (nq && qoff[nq] && (!qterms || !df)) || (nq && k && (!d_row_ids || !d_scores)))
            fail(MSVS_ERR_INVALID_ARGUMENT, "null buffer");
        if (nq == 0 || k == 0)
            return;
        hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);
        if (ps->num_docs == 0)
        {
            MSVS_HIP(hipMemsetAsync(d_row_ids, 0xFF, nq * k * 8, stream));
            return;
        }
        bm25_batch_device(*ps, nq, qoff, qterms, qgroups, df, total_docs, total_tokens, operator_or, d_alive_bits, nbits, k,
                          d_row_ids, d_scores, stream);
    });
}

extern "C" int msvs_bm25_search(const msvs_postings_t * ps, const uint32_t * qterms, const uint64_t * df,
                                size_t num_qterms, uint64_t total_docs, uint64_t total_tokens,
                                const uint64_t * alive_bits, size_t nbits, size_t k, uint64_t * row_ids, float * scores,
                                size_t * n_out)
{
    if (!n_out)
        return guarded([] { fail(MSVS_ERR_INVALID_ARGUMENT, "null buffer"); });
    *n_out = 0;
    if (k == 0 || num_qterms == 0)
        return MSVS_OK;
    const uint32_t qoff[2] = {0, (uint32_t)num_qterms};
    uint32_t cnt = 0;
    const uint64_t tokens[BM25_MAX_FIELDS] = {total_tokens, total_tokens, total_tokens, total_tokens};
    const int rc = msvs_bm25_search_batch(ps, 1, qoff, qterms, nullptr, df, total_docs, tokens, 1, alive_bits, nbits, k, row_ids,
                                          scores, &cnt);
    *n_out = cnt;
    return rc;
}

extern "C" int msvs_bm25_stats(uint64_t * queries, uint64_t * fallbacks)
{
    return guarded([&] {
        unsigned long long f = 0;
        unsigned long long * p = bm25_fail_counter();
        MSVS_HIP(hipDeviceSynchronize());
        MSVS_HIP(hipMemcpy(&f, p, 8, hipMemcpyDeviceToHost));
        if (queries)
            *queries = g_bm25_queries.load();
        if (fallbacks)
            *fallbacks = f;
    });
}

