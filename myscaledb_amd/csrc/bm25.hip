// bm25.hip -- seam B of include/msvs.h: BM25 scoring of exported posting lists on the GPU.
#include <cmath>
#include <memory>

#include "bm25_kernels.hpp"
#include "device_ops.hpp"

using namespace msvs;

struct msvs_postings
{
    DevBuf<int64_t> post_off;
    DevBuf<uint32_t> doc_ids, tfs;
    DevBuf<uint8_t> fieldnorm_ids;
    size_t num_terms = 0, num_docs = 0, num_postings = 0;
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
}

extern "C" int msvs_postings_create(const int64_t * post_off, size_t num_terms, const uint32_t * doc_ids,
                                    const uint32_t * tfs, const uint8_t * fieldnorm_ids, size_t num_docs,
                                    msvs_postings_t ** out)
{
    return guarded([&] {
        if (!out || !post_off || (num_docs && !fieldnorm_ids))
            fail(MSVS_ERR_INVALID_ARGUMENT, "null buffer");
        *out = nullptr;
        const size_t np = (size_t)post_off[num_terms];
        if (np && (!doc_ids || !tfs))
            fail(MSVS_ERR_INVALID_ARGUMENT, "null postings");
        if (num_docs > 0xfffffff0ull)
            fail(MSVS_ERR_ID_RANGE, "num_docs exceeds the u32 row id range");
        std::unique_ptr<msvs_postings> p(new msvs_postings);
        p->num_terms = num_terms;
        p->num_docs = num_docs;
        p->num_postings = np;
        p->post_off.alloc(num_terms + 1);
        p->doc_ids.alloc(std::max<size_t>(np, 1));
        p->tfs.alloc(std::max<size_t>(np, 1));
        p->fieldnorm_ids.alloc(std::max<size_t>(num_docs, 1));
        MSVS_HIP(hipMemcpy(p->post_off.p, post_off, (num_terms + 1) * 8, hipMemcpyHostToDevice));
        if (np)
        {
            MSVS_HIP(hipMemcpy(p->doc_ids.p, doc_ids, np * 4, hipMemcpyHostToDevice));
            MSVS_HIP(hipMemcpy(p->tfs.p, tfs, np * 4, hipMemcpyHostToDevice));
        }
        if (num_docs)
            MSVS_HIP(hipMemcpy(p->fieldnorm_ids.p, fieldnorm_ids, num_docs, hipMemcpyHostToDevice));
        *out = p.release();
    });
}

extern "C" void msvs_postings_free(msvs_postings_t * postings) { delete postings; }

extern "C" int msvs_bm25_search(const msvs_postings_t * ps, const uint32_t * qterms, const uint64_t * df,
                                size_t num_qterms, uint64_t total_docs, uint64_t total_tokens,
                                const uint64_t * alive_bits, size_t nbits, size_t k, uint64_t * row_ids, float * scores,
                                size_t * n_out)
{
    return guarded([&] {
        if (!ps || !n_out || (num_qterms && (!qterms || !df)) || (k && (!row_ids || !scores)))
            fail(MSVS_ERR_INVALID_ARGUMENT, "null buffer");
        *n_out = 0;
        if (k == 0 || num_qterms == 0 || ps->num_docs == 0)
            return;
        if (k > MSVS_MAX_K)
            fail(MSVS_ERR_UNSUPPORTED_K, "k = %zu exceeds the device top-k limit %d", k, MSVS_MAX_K);
        if (num_qterms > BM25_MAX_TERMS)
            fail(MSVS_ERR_INVALID_ARGUMENT, "more than %u query terms", BM25_MAX_TERMS);
        if (total_docs == 0)
            fail(MSVS_ERR_INVALID_ARGUMENT, "total_docs is zero");
        hipStream_t stream = nullptr;
        Bm25Params a{};
        // tantivy Bm25Weight (bm25.rs): K1 = 1.2, B = 0.75, all f32
        const float K1 = 1.2f, B = 0.75f;
        const float avg = (float)total_tokens / (float)total_docs;
        for (uint32_t i = 0; i < 256; i++)
        {
            volatile float t0 = B * (float)fieldnorm_of_id(i);
            volatile float t1 = t0 / avg;
            volatile float t2 = (1.0f - B) + t1;
            a.norm_cache[i] = K1 * t2;
        }
        for (size_t t = 0; t < num_qterms; t++)
        {
            if (qterms[t] >= ps->num_terms)
                fail(MSVS_ERR_INVALID_ARGUMENT, "query term id %u out of range", qterms[t]);
            if (df[t] > total_docs)
                fail(MSVS_ERR_INVALID_ARGUMENT, "doc_freq exceeds total_docs");
            volatile float x = ((float)(total_docs - df[t]) + 0.5f) / ((float)df[t] + 0.5f);
            volatile float idf = logf(1.0f + x);
            a.qterms[t] = qterms[t];
            a.weight[t] = idf * (1.0f + K1);
        }
        a.n_terms = (uint32_t)num_qterms;
        a.post_off = ps->post_off.p;
        a.doc_ids = ps->doc_ids.p;
        a.tfs = ps->tfs.p;
        a.fieldnorm_ids = ps->fieldnorm_ids.p;
        a.num_docs = (uint32_t)ps->num_docs;
        a.k = (uint32_t)k;
        const uint32_t n_blocks = (uint32_t)ceil_div(ps->num_docs, BM25_DOCS);
        const size_t words = alive_bits ? ceil_div(nbits, 64) : 0;
        // many doc blocks: their top-k lists are merged in two levels (32 groups, then the 32 group lists) -- one block
        // walking 1221 lists (10M documents) took 153 us against 52 us for the scoring itself
        const uint32_t groups = n_blocks > 64 ? 32 : 1;
        const uint32_t n_pad = (uint32_t)round_up((size_t)n_blocks, (size_t)groups);
        Scratch & scr = scratch_for(stream);
        scr.reserve((size_t)(n_pad + groups) * k * 8 + words * 8 + k * 12 + 8192, stream);
        uint64_t * partial = scr.take<uint64_t>((size_t)n_pad * k);
        if (n_pad > n_blocks) // the padding lists are empty
            MSVS_HIP(hipMemsetAsync(partial + (size_t)n_blocks * k, 0xFF, (size_t)(n_pad - n_blocks) * k * 8, stream));
        int64_t * d_ids = scr.take<int64_t>(k);
        float * d_sc = scr.take<float>(k);
        if (words)
        {
            uint64_t * d_alive = scr.take<uint64_t>(words);
            MSVS_HIP(hipMemcpyAsync(d_alive, alive_bits, words * 8, hipMemcpyHostToDevice, stream));
            a.alive = d_alive;
            a.nbits = (uint32_t)std::min<size_t>(nbits, 0xffffffffu);
        }
        a.partial = partial;
        const size_t lds = (size_t)5 * k * 8;
        ProfileScope prof("bm25_score", stream);
        switch (r_for_k((uint32_t)k))
        {
            case 1:
                hipLaunchKernelGGL((bm25_score_kernel<1>), dim3(n_blocks), dim3(BLOCK), lds, stream, a);
                break;
            case 2:
                hipLaunchKernelGGL((bm25_score_kernel<2>), dim3(n_blocks), dim3(BLOCK), lds, stream, a);
                break;
            default:
                hipLaunchKernelGGL((bm25_score_kernel<4>), dim3(n_blocks), dim3(BLOCK), lds, stream, a);
                break;
        }
        MSVS_HIP(hipGetLastError());
        MergeParams m{};
        m.partial = partial;
        m.n_lists = n_blocks;
        m.k = (uint32_t)k;
        if (groups > 1)
        {
            uint64_t * level1 = scr.take<uint64_t>((size_t)groups * k);
            m.n_lists = n_pad / groups; // group g merges lists [g * n_lists, (g + 1) * n_lists)
            m.mode = 2;
            m.out_keys = level1;
            launch_merge(M_IP, m, groups, stream);
            m.partial = level1;
            m.n_lists = groups;
            m.mode = 0;
            m.out_keys = nullptr;
        }
        m.out_ids = d_ids;
        m.out_dis = d_sc;
        launch_merge(M_IP, m, 1, stream);
        std::vector<int64_t> h_ids(k);
        MSVS_HIP(hipMemcpyAsync(h_ids.data(), d_ids, k * 8, hipMemcpyDeviceToHost, stream));
        MSVS_HIP(hipMemcpyAsync(scores, d_sc, k * 4, hipMemcpyDeviceToHost, stream));
        MSVS_HIP(hipStreamSynchronize(stream));
        size_t cnt = 0;
        while (cnt < k && h_ids[cnt] >= 0)
        {
            row_ids[cnt] = (uint64_t)h_ids[cnt];
            cnt++;
        }
        *n_out = cnt;
    });
}
