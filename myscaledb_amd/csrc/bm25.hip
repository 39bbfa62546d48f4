// bm25.hip -- seam B of include/msvs.h: BM25 scoring of exported posting lists on the GPU.
#include <cmath>
#include <memory>
#include <mutex>

#include "bm25_kernels.hpp"
#include "device_ops.hpp"

using namespace msvs;

struct msvs_postings
{
    DevBuf<int64_t> post_off;
    DevBuf<uint32_t> doc_ids, tfs;
    DevBuf<uint8_t> fieldnorm_ids; // [num_fields][num_docs]
    DevBuf<uint8_t> term_field;    // empty: every term belongs to field 0
    std::vector<int64_t> h_post_off;
    size_t num_terms = 0, num_docs = 0, num_postings = 0, num_fields = 1;
    // resident alive bitmap (lightweight deletes of the part): swapped under `mu`, a search keeps its own reference
    mutable std::mutex mu;
    std::shared_ptr<DevBuf<uint64_t>> alive;
    size_t alive_nbits = 0;
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
            if (post_off[t + 1] < post_off[t] || (term_field && term_field[t] >= num_fields))
                fail(MSVS_ERR_INVALID_ARGUMENT, "posting offsets must ascend and term fields must be < num_fields");
        std::unique_ptr<msvs_postings> p(new msvs_postings);
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
            p->term_field.alloc(num_terms);
            MSVS_HIP(hipMemcpy(p->term_field.p, term_field, num_terms, hipMemcpyHostToDevice));
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
/// One pass over a chunk of the batch.  Host inputs (term ids, df, statistics: a few numbers per query -- the weights
/// need libm's logf to match tantivy), device work on `stream`, results left in d_ids / d_scores ([nq][k], id -1 = no
/// hit).  d_alive: the effective filter, already on the device.
void bm25_chunk_device(const msvs_postings & ps, size_t nq, const uint32_t * qoff, const uint32_t * qterms,
                       const uint32_t * qgroups, const uint64_t * df, uint64_t total_docs, const float * cache,
                       int operator_or, const uint64_t * d_alive, size_t nbits, size_t k, int64_t * d_ids, float * d_scores,
                       hipStream_t stream)
{
    const size_t f0 = qoff[0], n_flat = qoff[nq] - f0, nf1 = std::max<size_t>(n_flat, 1), nc = ps.num_fields * 256;
    const float K1 = 1.2f;
    // one host blob -> one copy: [qoff u32][qterms u32][weight f32][cache f32][full u16][group u8]
    const size_t o_qoff = 0, o_terms = o_qoff + (nq + 1) * 4, o_w = o_terms + nf1 * 4, o_cache = o_w + nf1 * 4,
                 o_full = o_cache + nc * 4, o_group = o_full + round_up(nq * 2, (size_t)4), blob_bytes = round_up(o_group + nf1, (size_t)16);
    auto * blob = new std::vector<unsigned char>(blob_bytes);
    std::unique_ptr<std::vector<unsigned char>> blob_owner(blob);
    uint32_t * h_qoff = reinterpret_cast<uint32_t *>(blob->data() + o_qoff);
    uint32_t * h_terms = reinterpret_cast<uint32_t *>(blob->data() + o_terms);
    float * weight = reinterpret_cast<float *>(blob->data() + o_w);
    uint16_t * full = reinterpret_cast<uint16_t *>(blob->data() + o_full);
    uint8_t * group = blob->data() + o_group;
    memcpy(blob->data() + o_cache, cache, nc * 4);
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
            // tantivy Bm25Weight (bm25.rs): idf in f32
            volatile float x = ((float)(total_docs - df[j]) + 0.5f) / ((float)df[j] + 0.5f);
            volatile float idf = logf(1.0f + x);
            weight[j - f0] = idf * (1.0f + K1);
        }
        full[q] = m;
    }
    const uint32_t n_blocks = (uint32_t)std::max<size_t>(1, ceil_div(ps.num_docs, (size_t)BM25_DOCS));
    // many doc blocks: their top-k lists are merged in two levels (32 groups per query, then the group lists)
    const uint32_t groups = n_blocks > 64 ? 32 : 1;
    const uint32_t n_pad = (uint32_t)round_up((size_t)n_blocks, (size_t)groups);
    Scratch & scr = scratch_for(stream);
    scr.reserve(nq * (size_t)(n_pad + groups) * k * 8 + nf1 * (size_t)(n_blocks + 1) * 8 + blob_bytes + 65536, stream);
    Bm25Params a{};
    unsigned char * d_blob = scr.take<unsigned char>(blob_bytes);
    int64_t * d_bounds = scr.take<int64_t>(nf1 * (n_blocks + 1));
    uint64_t * partial = scr.take<uint64_t>(nq * (size_t)n_pad * k);
    MSVS_HIP(hipMemcpyAsync(d_blob, blob->data(), blob_bytes, hipMemcpyHostToDevice, stream));
    // the blob lives until the copy has run (no host synchronisation on this path)
    MSVS_HIP(hipLaunchHostFunc(stream, [](void * p) { delete static_cast<std::vector<unsigned char> *>(p); }, blob));
    blob_owner.release();
    a.alive = d_alive;
    a.nbits = (uint32_t)std::min<size_t>(nbits, 0xffffffffu);
    a.qoff = reinterpret_cast<uint32_t *>(d_blob + o_qoff);
    a.qterms = reinterpret_cast<uint32_t *>(d_blob + o_terms);
    a.weight = reinterpret_cast<float *>(d_blob + o_w);
    a.norm_cache = reinterpret_cast<float *>(d_blob + o_cache);
    a.qfull = reinterpret_cast<uint16_t *>(d_blob + o_full);
    a.qgroup = d_blob + o_group;
    a.bounds = d_bounds;
    a.partial = partial;
    a.post_off = ps.post_off.p;
    a.doc_ids = ps.doc_ids.p;
    a.tfs = ps.tfs.p;
    a.fieldnorm_ids = ps.fieldnorm_ids.p;
    a.term_field = ps.term_field.p;
    a.num_docs = (uint32_t)ps.num_docs;
    a.num_fields = (uint32_t)ps.num_fields;
    a.n_blocks = n_blocks;
    a.n_pad = n_pad;
    a.k = (uint32_t)k;
    a.nq = (uint32_t)nq;
    a.operator_or = operator_or;
    if (n_pad > n_blocks) // the padding lists are empty
        for (size_t q = 0; q < nq; q++)
            MSVS_HIP(hipMemsetAsync(partial + (q * n_pad + n_blocks) * k, 0xFF, (size_t)(n_pad - n_blocks) * k * 8, stream));
    {
        ProfileScope prof("bm25_score", stream);
        if (n_flat)
            hipLaunchKernelGGL(bm25_bounds_kernel, dim3((unsigned)ceil_div(n_flat * (size_t)(n_blocks + 1), (size_t)256)), dim3(256),
                               0, stream, a, (uint32_t)n_flat);
        // enough blocks for the chip: split the batch over grid.y when the corpus has few document blocks
        const uint32_t y = (uint32_t)std::min<size_t>(nq, std::max<size_t>(1, 2048 / n_blocks));
        const size_t lds = (size_t)5 * k * 8;
        const dim3 grid(n_blocks, y);
        switch (r_for_k((uint32_t)k))
        {
            case 1:
                hipLaunchKernelGGL((bm25_score_kernel<1>), grid, dim3(BLOCK), lds, stream, a);
                break;
            case 2:
                hipLaunchKernelGGL((bm25_score_kernel<2>), grid, dim3(BLOCK), lds, stream, a);
                break;
            default:
                hipLaunchKernelGGL((bm25_score_kernel<4>), grid, dim3(BLOCK), lds, stream, a);
                break;
        }
        MSVS_HIP(hipGetLastError());
    }
    MergeParams m{};
    m.partial = partial;
    m.n_lists = n_blocks;
    m.k = (uint32_t)k;
    if (groups > 1)
    {
        uint64_t * level1 = scr.take<uint64_t>(nq * (size_t)groups * k);
        m.n_lists = n_pad / groups; // "query" (q, g) merges lists [g * n_lists, (g + 1) * n_lists) of query q
        m.mode = 2;
        m.out_keys = level1;
        launch_merge(M_IP, m, (uint32_t)(nq * groups), stream);
        m.partial = level1;
        m.n_lists = groups;
        m.mode = 0;
        m.out_keys = nullptr;
    }
    m.out_ids = d_ids;
    m.out_dis = d_scores;
    launch_merge(M_IP, m, (uint32_t)nq, stream);
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
    const size_t n_blocks = std::max<size_t>(1, ceil_div(ps.num_docs, (size_t)BM25_DOCS));
    const size_t per_q = (n_blocks + 64) * k * 8;
    const size_t chunk = std::max<size_t>(1, std::min<size_t>(nq, ((size_t)256 << 20) / per_q));
    for (size_t q0 = 0; q0 < nq; q0 += chunk)
    {
        const size_t nqc = std::min(chunk, nq - q0);
        bm25_chunk_device(ps, nqc, qoff + q0, qterms, qgroups, df, total_docs, cache.data(), operator_or, eff, eff_bits, k,
                          d_ids + q0 * k, d_scores + q0 * k, stream);
    }
    // `resident` is held until every kernel reading it is enqueued; a swapped-out bitmap is released by hipFree, which
    // waits for the device
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
        if (nq == 0 || k == 0 || ps->num_docs == 0)
            return;
        hipStream_t stream = nullptr;
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
