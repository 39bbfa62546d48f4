// text_store.cpp -- host side of seam B: what sits between TantivyIndexStore's search methods and the device scorer.
//
//   TantivyIndexStore::bm25Search / bm25SearchWithFilter      src/Storages/MergeTree/TantivyIndexStore.cpp:900-954
//   TantivyIndexStore::getDocFreq / getTotalNumDocs / getTotalNumTokens                       .cpp:957-992
//   TANTIVY::ffi_index_multi_column_docs / ffi_index_writer_commit (the index side)            .cpp:654-769
//
// The reference crosses into the Rust tantivy_search library with an index DIRECTORY, a sentence, column names, an alive
// bitmap and table-level statistics.  Here the same call lands on a part's POSTINGS EXPORT: the inverted index as flat
// arrays (one CSR posting list per (column, token) term, u32 doc ids ascending, u32 term frequencies, one fieldnorm byte
// per (column, document)), resident in HBM (msvs_postings_t), plus the term dictionary on the host.  The export is a
// file ("MSVSPOST", layout below) written once per part next to the tantivy files; a deployment produces it by walking
// the tantivy segment (TermDictionary::stream + SegmentPostings + FieldNormReader -- the Rust side is absent from this
// tree); the exporter in THIS file builds it from the documents with the tokenizer the goldens pin, which is also what
// the tests use.
//
// Tokenizer: tantivy "default" = SimpleTokenizer (maximal runs of alphanumerics) -> RemoveLongFilter(40) -> LowerCaser.
// ASCII letters are lower-cased; bytes >= 0x80 (UTF-8 multi-byte sequences) count as alphanumeric and are kept as they
// are.  The natural-language query parser (enable_nlq) stays with tantivy: a sentence is a bag of tokens here.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/msvs_host.h"
#include "msvs_host.hpp"
#include "unicode_tables.hpp"

namespace
{
thread_local std::string g_text_error;

[[noreturn]] void text_fail(int code, const std::string & m)
{
    g_text_error = m;
    throw VectorIndex::VIException(code, m);
}

template <typename F>
int text_guarded(F && f)
{
    try
    {
        g_text_error.clear();
        f();
        return MSVS_OK;
    }
    catch (const VectorIndex::VIException & e)
    {
        if (g_text_error.empty())
            g_text_error = e.what();
        return e.code;
    }
    catch (const std::exception & e)
    {
        g_text_error = e.what();
        return MSVS_ERR_DEVICE;
    }
}

/// The default tokenizer chain of a tantivy_search `fts` index: SimpleTokenizer (split on every code point that is not
/// alphanumeric), RemoveLongFilter(40 bytes), LowerCaser (ASCII fast path, char::to_lowercase otherwise).  The Unicode
/// properties come from unicode_tables.hpp (generated, tools/gen_unicode_tables.py; known difference: combining marks with
/// the Other_Alphabetic property split a token here).  Malformed UTF-8 bytes are separators.
static bool uni_alnum(uint32_t cp)
{
    size_t lo = 0, hi = sizeof(msvs_unicode::kAlnum) / sizeof(msvs_unicode::kAlnum[0]);
    while (lo < hi)
    {
        const size_t mid = (lo + hi) / 2;
        if (cp > msvs_unicode::kAlnum[mid].hi)
            lo = mid + 1;
        else if (cp < msvs_unicode::kAlnum[mid].lo)
            hi = mid;
        else
            return true;
    }
    return false;
}

static uint32_t uni_lower(uint32_t cp)
{
    size_t lo = 0, hi = sizeof(msvs_unicode::kLower) / sizeof(msvs_unicode::kLower[0]);
    while (lo < hi)
    {
        const size_t mid = (lo + hi) / 2;
        if (cp > msvs_unicode::kLower[mid].from)
            lo = mid + 1;
        else if (cp < msvs_unicode::kLower[mid].from)
            hi = mid;
        else
            return msvs_unicode::kLower[mid].to;
    }
    return cp;
}

static void put_utf8(std::string & s, uint32_t cp)
{
    if (cp < 0x80)
        s.push_back((char)cp);
    else if (cp < 0x800)
    {
        s.push_back((char)(0xC0 | (cp >> 6)));
        s.push_back((char)(0x80 | (cp & 0x3F)));
    }
    else if (cp < 0x10000)
    {
        s.push_back((char)(0xE0 | (cp >> 12)));
        s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
        s.push_back((char)(0x80 | (cp & 0x3F)));
    }
    else
    {
        s.push_back((char)(0xF0 | (cp >> 18)));
        s.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
        s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
        s.push_back((char)(0x80 | (cp & 0x3F)));
    }
}

void tokenize(const char * text, std::vector<std::string> & out)
{
    std::string cur;
    size_t raw = 0; // bytes of the token as SimpleTokenizer cut it: RemoveLongFilter(40) sits BEFORE LowerCaser in tantivy's default
                    // chain, and lower-casing changes byte lengths (U+212A KELVIN SIGN, 3 bytes -> 'k'; U+0130 -> "i" + U+0307)
    auto flush = [&] {
        if (!cur.empty() && raw < 40)
            out.push_back(cur);
        cur.clear();
        raw = 0;
    };
    const unsigned char * p = reinterpret_cast<const unsigned char *>(text);
    while (*p)
    {
        const unsigned char c = *p;
        if (c < 0x80)
        {
            p++;
            if ((c >= '0' && c <= '9') || (c >= 'a' && c <= 'z'))
                cur.push_back((char)c), raw++;
            else if (c >= 'A' && c <= 'Z')
                cur.push_back((char)(c - 'A' + 'a')), raw++;
            else
                flush();
            continue;
        }
        // one UTF-8 sequence (shortest form, no surrogates); anything else is a separator byte
        uint32_t cp = 0;
        int len = 0;
        if ((c & 0xE0) == 0xC0)
            cp = c & 0x1F, len = 2;
        else if ((c & 0xF0) == 0xE0)
            cp = c & 0x0F, len = 3;
        else if ((c & 0xF8) == 0xF0)
            cp = c & 0x07, len = 4;
        bool ok = len != 0;
        for (int i = 1; ok && i < len; i++)
        {
            if ((p[i] & 0xC0) != 0x80) // also stops at the terminating NUL
                ok = false;
            else
                cp = cp << 6 | (p[i] & 0x3F);
        }
        if (ok && ((len == 2 && cp < 0x80) || (len == 3 && cp < 0x800) || (len == 4 && (cp < 0x10000 || cp > 0x10FFFF)) || (cp >= 0xD800 && cp <= 0xDFFF)))
            ok = false;
        if (!ok)
        {
            p++;
            flush();
            continue;
        }
        p += len;
        if (uni_alnum(cp))
        {
            raw += (size_t)len;
            if (cp == 0x130) // LATIN CAPITAL LETTER I WITH DOT ABOVE: char::to_lowercase gives TWO code points, "i" + U+0307
            {
                cur.push_back('i');
                put_utf8(cur, 0x307);
            }
            else
                put_utf8(cur, uni_lower(cp));
        }
        else
            flush();
    }
    flush();
}

/// tantivy fieldnorm_to_id (tantivy/src/fieldnorm/code.rs): largest id whose table value is <= len.
uint32_t fieldnorm_value(uint32_t b)
{
    if (b < 24)
        return b;
    const uint32_t i = b - 24, bits = i & 7;
    const int shift = (int)(i >> 3) - 1;
    const uint64_t dec = shift < 0 ? bits : ((uint64_t)(bits | 8) << shift);
    const uint64_t v = 24 + dec;
    return v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v;
}

uint8_t fieldnorm_id(uint32_t len)
{
    int lo = 0, hi = 255;
    while (lo < hi)
    {
        const int mid = (lo + hi + 1) / 2;
        if (fieldnorm_value((uint32_t)mid) <= len)
            lo = mid;
        else
            hi = mid - 1;
    }
    return (uint8_t)lo;
}

struct ExportHeader // 64 bytes, little endian
{
    char magic[8]; // "MSVSPOST"
    uint32_t version, num_fields;
    uint64_t num_docs, num_terms, num_postings, dict_bytes, names_bytes;
    uint64_t reserved;
};
static_assert(sizeof(ExportHeader) == 64, "export header layout");
}

/// One part's text index: writer state until commit(), then the export + its device copy.
struct msvs_text_index
{
    std::vector<std::string> columns;
    // writer: (field, token) -> postings being built (doc ids ascend because documents arrive in row order)
    std::map<std::pair<uint32_t, std::string>, std::vector<std::pair<uint32_t, uint32_t>>> building;
    std::vector<std::vector<uint8_t>> fn_building; // [field][doc]
    // the export
    bool committed = false;
    uint64_t num_docs = 0;
    std::vector<uint64_t> total_tokens; // [field]
    std::vector<uint8_t> term_field;    // [term], terms sorted by (field, bytes) like tantivy's term dictionary
    std::vector<uint64_t> term_str_off; // [term + 1]
    std::string term_bytes;
    std::vector<int64_t> post_off;
    std::vector<uint32_t> doc_ids, tfs;
    std::vector<uint8_t> fieldnorm_ids; // [field][doc]
    msvs_postings_t * device = nullptr;

    ~msvs_text_index() { msvs_postings_free(device); }

    size_t num_terms() const { return term_field.size(); }

    int64_t find_term(uint32_t field, const std::string & tok) const
    {
        size_t lo = 0, hi = num_terms();
        while (lo < hi)
        {
            const size_t mid = (lo + hi) / 2;
            const size_t len = term_str_off[mid + 1] - term_str_off[mid];
            int c = (int)term_field[mid] - (int)field;
            if (c == 0)
            {
                c = memcmp(term_bytes.data() + term_str_off[mid], tok.data(), std::min(len, tok.size()));
                if (c == 0)
                    c = len < tok.size() ? -1 : (len > tok.size() ? 1 : 0);
            }
            if (c == 0)
                return (int64_t)mid;
            if (c < 0)
                lo = mid + 1;
            else
                hi = mid;
        }
        return -1;
    }

    int field_of(const char * name) const
    {
        for (size_t f = 0; f < columns.size(); f++)
            if (columns[f] == name)
                return (int)f;
        return -1;
    }

    void upload()
    {
        msvs_postings_free(device);
        device = nullptr;
        VectorIndex::throwIfError(msvs_postings_create_fields(post_off.data(), num_terms(), term_field.data(), doc_ids.data(),
                                                              tfs.data(), fieldnorm_ids.data(), columns.size(), num_docs, &device));
    }
};

extern "C" {

MSVS_HOST_API const char * msvs_text_last_error(void) { return g_text_error.empty() ? msvs_last_error() : g_text_error.c_str(); }

MSVS_HOST_API int msvs_text_index_create(const char * const * column_names, size_t ncols, msvs_text_index_t ** out)
{
    return text_guarded([&] {
        if (!out || !column_names || ncols == 0 || ncols > 4)
            text_fail(MSVS_ERR_INVALID_ARGUMENT, "a text index takes 1 .. 4 columns");
        auto ix = std::make_unique<msvs_text_index>();
        for (size_t c = 0; c < ncols; c++)
            ix->columns.emplace_back(column_names[c]);
        ix->fn_building.resize(ncols);
        ix->total_tokens.assign(ncols, 0);
        *out = ix.release();
    });
}

MSVS_HOST_API void msvs_text_index_free(msvs_text_index_t * ix) { delete ix; }

MSVS_HOST_API int msvs_text_index_add_doc(msvs_text_index_t * ix, uint64_t row_id, const char * const * column_names,
                                          const char * const * docs, size_t ncols)
{
    return text_guarded([&] {
        if (!ix || ix->committed)
            text_fail(MSVS_ERR_INVALID_ARGUMENT, "index writer is closed");
        if (row_id != ix->num_docs)
            text_fail(MSVS_ERR_INVALID_ARGUMENT, "documents arrive in row order: expected row " + std::to_string(ix->num_docs));
        if (row_id >= 0xfffffff0ull)
            text_fail(MSVS_ERR_ID_RANGE, "a part holds fewer than 2^32 rows");
        std::vector<std::string> toks;
        for (size_t f = 0; f < ix->columns.size(); f++)
        {
            toks.clear();
            for (size_t c = 0; c < ncols; c++) // an Array(String) column arrives as several docs of one column
                if (ix->columns[f] == column_names[c])
                    tokenize(docs[c], toks);
            ix->fn_building[f].push_back(fieldnorm_id((uint32_t)toks.size()));
            ix->total_tokens[f] += toks.size();
            for (const auto & t : toks)
            {
                auto & pl = ix->building[{(uint32_t)f, t}];
                if (!pl.empty() && pl.back().first == (uint32_t)row_id)
                    pl.back().second++;
                else
                    pl.emplace_back((uint32_t)row_id, 1u);
            }
        }
        ix->num_docs++;
    });
}

MSVS_HOST_API int msvs_text_index_commit(msvs_text_index_t * ix)
{
    return text_guarded([&] {
        if (!ix || ix->committed)
            text_fail(MSVS_ERR_INVALID_ARGUMENT, "index writer is closed");
        ix->term_str_off.assign(1, 0);
        ix->post_off.assign(1, 0);
        for (auto & kv : ix->building) // std::map order = (field, bytes)
        {
            ix->term_field.push_back((uint8_t)kv.first.first);
            ix->term_bytes += kv.first.second;
            ix->term_str_off.push_back(ix->term_bytes.size());
            for (auto & p : kv.second)
            {
                ix->doc_ids.push_back(p.first);
                ix->tfs.push_back(p.second);
            }
            ix->post_off.push_back((int64_t)ix->doc_ids.size());
        }
        for (auto & f : ix->fn_building)
            ix->fieldnorm_ids.insert(ix->fieldnorm_ids.end(), f.begin(), f.end());
        ix->building.clear();
        ix->fn_building.clear();
        ix->committed = true;
        ix->upload();
    });
}

/* The export file:  ExportHeader | total_tokens u64[fields] | column names ('\0'-separated, names_bytes) |
 * term_field u8[terms] | term_str_off u64[terms + 1] | term bytes | post_off i64[terms + 1] | doc_ids u32[postings] |
 * tfs u32[postings] | fieldnorm_ids u8[fields * docs].  No padding; everything little endian. */
MSVS_HOST_API int msvs_text_index_save(const msvs_text_index_t * ix, const char * path)
{
    return text_guarded([&] {
        if (!ix || !ix->committed || !path)
            text_fail(MSVS_ERR_INVALID_ARGUMENT, "commit the index before exporting it");
        std::string names;
        for (auto & c : ix->columns)
            names += c + '\0';
        ExportHeader h{};
        memcpy(h.magic, "MSVSPOST", 8);
        h.version = 1;
        h.num_fields = (uint32_t)ix->columns.size();
        h.num_docs = ix->num_docs;
        h.num_terms = ix->num_terms();
        h.num_postings = ix->doc_ids.size();
        h.dict_bytes = ix->term_bytes.size();
        h.names_bytes = names.size();
        FILE * f = fopen(path, "wb");
        if (!f)
            text_fail(MSVS_ERR_IO, std::string("cannot create ") + path);
        auto put = [&](const void * p, size_t n) { return n == 0 || fwrite(p, 1, n, f) == n; };
        const bool ok = put(&h, sizeof h) && put(ix->total_tokens.data(), ix->total_tokens.size() * 8) && put(names.data(), names.size())
            && put(ix->term_field.data(), ix->term_field.size()) && put(ix->term_str_off.data(), ix->term_str_off.size() * 8)
            && put(ix->term_bytes.data(), ix->term_bytes.size()) && put(ix->post_off.data(), ix->post_off.size() * 8)
            && put(ix->doc_ids.data(), ix->doc_ids.size() * 4) && put(ix->tfs.data(), ix->tfs.size() * 4)
            && put(ix->fieldnorm_ids.data(), ix->fieldnorm_ids.size());
        if (fclose(f) != 0 || !ok)
            text_fail(MSVS_ERR_IO, std::string("short write to ") + path);
    });
}

MSVS_HOST_API int msvs_text_index_load(const char * path, msvs_text_index_t ** out)
{
    return text_guarded([&] {
        if (!path || !out)
            text_fail(MSVS_ERR_INVALID_ARGUMENT, "null argument");
        *out = nullptr;
        FILE * f = fopen(path, "rb");
        if (!f)
            text_fail(MSVS_ERR_IO, std::string("cannot open ") + path);
        std::unique_ptr<FILE, int (*)(FILE *)> closer(f, fclose);
        auto get = [&](void * p, size_t n) {
            if (n && fread(p, 1, n, f) != n)
                text_fail(MSVS_ERR_IO, std::string("truncated postings export ") + path);
        };
        ExportHeader h{};
        get(&h, sizeof h);
        if (memcmp(h.magic, "MSVSPOST", 8) != 0 || h.version != 1 || h.num_fields < 1 || h.num_fields > 4
            || h.num_docs >= 0xfffffff0ull || h.num_terms > ((uint64_t)1 << 40) || h.num_postings > ((uint64_t)1 << 44)
            || h.dict_bytes > ((uint64_t)1 << 40) || h.names_bytes > 4096)
            text_fail(MSVS_ERR_IO, std::string("not a postings export (bad header): ") + path);
        auto ix = std::make_unique<msvs_text_index>();
        ix->num_docs = h.num_docs;
        ix->total_tokens.resize(h.num_fields);
        get(ix->total_tokens.data(), h.num_fields * 8);
        std::string names(h.names_bytes, '\0');
        get(&names[0], names.size());
        for (size_t p = 0; p < names.size();)
        {
            const size_t e = names.find('\0', p);
            if (e == std::string::npos)
                break;
            ix->columns.push_back(names.substr(p, e - p));
            p = e + 1;
        }
        if (ix->columns.size() != h.num_fields)
            text_fail(MSVS_ERR_IO, "postings export: column names do not match the field count");
        ix->term_field.resize(h.num_terms);
        ix->term_str_off.resize(h.num_terms + 1);
        ix->term_bytes.resize(h.dict_bytes);
        ix->post_off.resize(h.num_terms + 1);
        ix->doc_ids.resize(h.num_postings);
        ix->tfs.resize(h.num_postings);
        ix->fieldnorm_ids.resize(h.num_fields * h.num_docs);
        get(ix->term_field.data(), h.num_terms);
        get(ix->term_str_off.data(), (h.num_terms + 1) * 8);
        get(&ix->term_bytes[0], h.dict_bytes);
        get(ix->post_off.data(), (h.num_terms + 1) * 8);
        get(ix->doc_ids.data(), h.num_postings * 4);
        get(ix->tfs.data(), h.num_postings * 4);
        get(ix->fieldnorm_ids.data(), ix->fieldnorm_ids.size());
        // structure checks: a corrupt export must not reach the device
        if (ix->post_off[0] != 0 || (uint64_t)ix->post_off[h.num_terms] != h.num_postings || ix->term_str_off[0] != 0
            || ix->term_str_off[h.num_terms] != h.dict_bytes)
            text_fail(MSVS_ERR_IO, "postings export: offsets do not cover the arrays");
        for (uint64_t t = 0; t < h.num_terms; t++)
        {
            if (ix->post_off[t + 1] < ix->post_off[t] || ix->term_str_off[t + 1] < ix->term_str_off[t] || ix->term_field[t] >= h.num_fields)
                text_fail(MSVS_ERR_IO, "postings export: descending offsets / bad field id");
            for (int64_t p = ix->post_off[t]; p < ix->post_off[t + 1]; p++)
                if (ix->doc_ids[p] >= h.num_docs || (p > ix->post_off[t] && ix->doc_ids[p] <= ix->doc_ids[p - 1]))
                    text_fail(MSVS_ERR_IO, "postings export: doc ids must ascend inside a posting list and stay below num_docs");
        }
        ix->committed = true;
        ix->upload();
        *out = ix.release();
    });
}

MSVS_HOST_API uint64_t msvs_text_index_total_num_docs(const msvs_text_index_t * ix) { return ix ? ix->num_docs : 0; }

MSVS_HOST_API int msvs_text_index_total_num_tokens(const msvs_text_index_t * ix, msvs_field_tokens_t * out, size_t cap, size_t * n)
{
    return text_guarded([&] {
        if (!ix || !n)
            text_fail(MSVS_ERR_INVALID_ARGUMENT, "null argument");
        *n = ix->columns.size();
        for (size_t f = 0; f < ix->columns.size() && f < cap; f++)
        {
            out[f].field_id = (uint32_t)f;
            out[f].field_total_tokens = ix->total_tokens[f];
        }
    });
}

/* ffi_get_doc_freq: every (token of the sentence, column) with this part's document frequency.  `term` pointers stay
 * valid until the calling thread's next msvs_text_index_doc_freq call. */
MSVS_HOST_API int msvs_text_index_doc_freq(const msvs_text_index_t * ix, const char * sentence, msvs_doc_freq_t * out, size_t cap,
                                           size_t * n)
{
    return text_guarded([&] {
        if (!ix || !ix->committed || !sentence || !n)
            text_fail(MSVS_ERR_INVALID_ARGUMENT, "null argument / index not committed");
        static thread_local std::vector<std::string> held;
        held.clear();
        tokenize(sentence, held);
        std::sort(held.begin(), held.end());
        held.erase(std::unique(held.begin(), held.end()), held.end());
        size_t cnt = 0;
        for (const auto & t : held)
            for (size_t f = 0; f < ix->columns.size(); f++)
            {
                if (cnt < cap)
                {
                    const int64_t id = ix->find_term((uint32_t)f, t);
                    out[cnt].term = t.c_str();
                    out[cnt].field_id = (uint32_t)f;
                    out[cnt].doc_freq = id < 0 ? 0 : (uint64_t)(ix->post_off[id + 1] - ix->post_off[id]);
                }
                cnt++;
            }
        *n = cnt;
    });
}

/* The tokens of a text under the index's (default) tokenizer, '\n'-separated into buf; *n_needed = bytes needed including the NUL
 * (call again with a larger buffer when it exceeds cap).  Tests compare it with a Python restatement on non-ASCII text. */
MSVS_HOST_API int msvs_text_tokenize(const char * text, char * buf, size_t cap, size_t * n_needed)
{
    return text_guarded([&] {
        if (!text || !n_needed)
            text_fail(MSVS_ERR_INVALID_ARGUMENT, "null argument");
        std::vector<std::string> toks;
        tokenize(text, toks);
        std::string joined;
        for (size_t i = 0; i < toks.size(); i++)
        {
            if (i)
                joined.push_back('\n');
            joined += toks[i];
        }
        *n_needed = joined.size() + 1;
        if (buf && cap >= joined.size() + 1)
            memcpy(buf, joined.c_str(), joined.size() + 1);
    });
}

MSVS_HOST_API int msvs_text_index_set_alive(msvs_text_index_t * ix, const uint8_t * u8_alive_bitmap, size_t nbytes)
{
    return text_guarded([&] {
        if (!ix || !ix->committed)
            text_fail(MSVS_ERR_INVALID_ARGUMENT, "index not committed");
        if (!u8_alive_bitmap)
        {
            VectorIndex::throwIfError(msvs_postings_set_alive(ix->device, nullptr, 0));
            return;
        }
        std::vector<uint64_t> words((nbytes + 7) / 8, 0);
        memcpy(words.data(), u8_alive_bitmap, nbytes); // byte i bit j = row 8 i + j: the same bits as little-endian u64 words
        VectorIndex::throwIfError(msvs_postings_set_alive(ix->device, words.data(), nbytes * 8));
    });
}

/* ffi_bm25_search for a batch of sentences (nq = 1: exactly the reference's call).  column_names NULL / ncols 0 = every
 * column of the index.  stats NULL or empty = this part's own statistics (a one-part table). */
MSVS_HOST_API int msvs_text_index_bm25_search_batch(const msvs_text_index_t * ix, const char * const * sentences, size_t nq,
                                                    const char * const * column_names, size_t ncols, uint32_t topk,
                                                    const uint8_t * u8_alive_bitmap, size_t nbytes, int use_filter, int enable_nlq,
                                                    int operator_or, const msvs_bm25_stats_t * stats, uint64_t * row_ids,
                                                    float * scores, uint32_t * n_out)
{
    return text_guarded([&] {
        if (!ix || !ix->committed || (nq && (!sentences || !n_out)))
            text_fail(MSVS_ERR_INVALID_ARGUMENT, "null argument / index not committed");
        if (enable_nlq)
            text_fail(MSVS_ERR_NOT_IMPLEMENTED, "the natural language query parser stays with tantivy_search (enable_nlq = false here)");
        std::vector<uint32_t> fields;
        if (!column_names || ncols == 0)
            for (size_t f = 0; f < ix->columns.size(); f++)
                fields.push_back((uint32_t)f);
        for (size_t c = 0; column_names && c < ncols; c++)
        {
            const int f = ix->field_of(column_names[c]);
            if (f < 0)
                text_fail(MSVS_ERR_INVALID_ARGUMENT, std::string("column ") + column_names[c] + " is not in this text index");
            fields.push_back((uint32_t)f);
        }
        const bool own_stats = !stats || stats->total_num_docs == 0;
        const uint64_t total_docs = own_stats ? ix->num_docs : stats->total_num_docs;
        std::vector<uint64_t> tokens(ix->columns.size());
        for (size_t f = 0; f < tokens.size(); f++)
            tokens[f] = ix->total_tokens[f];
        if (!own_stats)
            for (size_t i = 0; i < stats->n_fields; i++)
                if (stats->total_num_tokens[i].field_id < tokens.size())
                    tokens[stats->total_num_tokens[i].field_id] = stats->total_num_tokens[i].field_total_tokens;
        std::map<std::pair<uint32_t, std::string>, uint64_t> table_df;
        if (!own_stats)
            for (size_t i = 0; i < stats->n_docs_freq; i++)
                table_df[{stats->docs_freq[i].field_id, stats->docs_freq[i].term}] = stats->docs_freq[i].doc_freq;
        std::vector<uint32_t> qoff(nq + 1, 0), qterms, qgroups;
        std::vector<uint64_t> df;
        std::vector<std::string> toks;
        for (size_t q = 0; q < nq; q++)
        {
            toks.clear();
            tokenize(sentences[q], toks);
            uint32_t group = 0;
            bool impossible = false; // AND with a token this part has never seen: no document can match
            std::vector<std::string> seen;
            for (const auto & t : toks)
            {
                if (std::find(seen.begin(), seen.end(), t) != seen.end())
                    continue; // a repeated token is one clause
                seen.push_back(t);
                bool any = false;
                for (uint32_t f : fields) // token outer, column inner: the order tantivy's query parser emits clauses in
                {
                    const int64_t id = ix->find_term(f, t);
                    if (id < 0)
                        continue;
                    uint64_t d = (uint64_t)(ix->post_off[id + 1] - ix->post_off[id]);
                    if (!own_stats)
                    {
                        auto it = table_df.find({f, t});
                        if (it != table_df.end())
                            d = it->second;
                    }
                    qterms.push_back((uint32_t)id);
                    qgroups.push_back(group);
                    df.push_back(d);
                    any = true;
                }
                if (any)
                    group++;
                else if (!operator_or)
                    impossible = true;
            }
            if (impossible)
            {
                qterms.resize(qoff[q]);
                qgroups.resize(qoff[q]);
                df.resize(qoff[q]);
            }
            qoff[q + 1] = (uint32_t)qterms.size();
        }
        std::vector<uint64_t> words;
        if (use_filter && (!u8_alive_bitmap || nbytes == 0))
        {
            // a filter was asked for and it is empty: no row passes (NOT "no filter")
            for (size_t q = 0; q < nq; q++)
                n_out[q] = 0;
            return;
        }
        const bool filter = use_filter && u8_alive_bitmap;
        if (filter)
        {
            words.assign((nbytes + 7) / 8, 0);
            memcpy(words.data(), u8_alive_bitmap, nbytes);
        }
        VectorIndex::throwIfError(msvs_bm25_search_batch(ix->device, nq, qoff.data(), qterms.data(), qgroups.data(), df.data(),
                                                         total_docs, tokens.data(), operator_or, filter ? words.data() : nullptr,
                                                         filter ? nbytes * 8 : 0, topk, row_ids, scores, n_out));
    });
}

MSVS_HOST_API int msvs_text_index_bm25_search(const msvs_text_index_t * ix, const char * sentence, const char * const * column_names,
                                              size_t ncols, uint32_t topk, const uint8_t * u8_alive_bitmap, size_t nbytes,
                                              int use_filter, int enable_nlq, int operator_or, const msvs_bm25_stats_t * stats,
                                              uint64_t * row_ids, float * scores, uint32_t * n_out)
{
    return msvs_text_index_bm25_search_batch(ix, &sentence, 1, column_names, ncols, topk, u8_alive_bitmap, nbytes, use_filter,
                                             enable_nlq, operator_or, stats, row_ids, scores, n_out);
}

/* ------------------------------------------------------------------------------- distributed BM25 statistics (DFS)
 * What a Distributed-table text / hybrid search does before any shard scores: the initiator runs
 * ftsIndex(db, table, column, query_text) on every shard, each shard answers with ONE row summed over its parts
 * (ReadFromFtsIndex::initializePipeline, src/VectorIndex/Storages/StorageFtsIndex.cpp:150-213), the initiator adds the rows up
 * (collectStatisticForBM25Calculation / parseBM25StaisiticsInfo, src/VectorIndex/Utils/CommonUtils.cpp:190-330) and ships the
 * result to the shards as the "_fts_statistic_info" scalar, which becomes the `statistics` argument of every ffi_bm25_search.
 * Both sums keep the reference's container order: fields by id, terms by (field_id, term bytes). */
struct msvs_fts_stats
{
    uint64_t total_docs = 0;
    std::map<uint32_t, uint64_t> tokens;
    std::map<std::pair<uint32_t, std::string>, uint64_t> terms;
    std::vector<msvs_field_tokens_t> tokens_v;
    std::vector<msvs_doc_freq_t> terms_v;
    msvs_bm25_stats_t view{};
    void finish()
    {
        for (const auto & kv : tokens)
            tokens_v.push_back(msvs_field_tokens_t{kv.first, kv.second});
        for (const auto & kv : terms)
            terms_v.push_back(msvs_doc_freq_t{kv.first.second.c_str(), kv.first.first, kv.second}); // map nodes do not move
        view.docs_freq = terms_v.data();
        view.n_docs_freq = terms_v.size();
        view.total_num_tokens = tokens_v.data();
        view.n_fields = tokens_v.size();
        view.total_num_docs = total_docs;
    }
};

MSVS_HOST_API int msvs_host_fts_index_statistics(const msvs_text_index_t * const * parts, size_t nparts, const char * query_text,
                                                 msvs_fts_stats_t ** out)
{
    return text_guarded([&] {
        if (!out || !query_text || (nparts && !parts))
            text_fail(MSVS_ERR_INVALID_ARGUMENT, "null argument");
        std::unique_ptr<msvs_fts_stats> st(new msvs_fts_stats);
        std::vector<std::string> toks;
        tokenize(query_text, toks);
        std::sort(toks.begin(), toks.end());
        toks.erase(std::unique(toks.begin(), toks.end()), toks.end());
        for (size_t p = 0; p < nparts; p++)
        {
            const msvs_text_index_t * ix = parts[p];
            if (!ix || !ix->committed) // the reference: "Fts index file ... does not exist" (NOT_IMPLEMENTED)
                text_fail(MSVS_ERR_NOT_IMPLEMENTED, "part " + std::to_string(p) + " has no committed text index");
            st->total_docs += ix->num_docs;
            for (size_t f = 0; f < ix->columns.size(); f++)
                st->tokens[(uint32_t)f] += ix->total_tokens[f];
            for (const auto & t : toks)
                for (size_t f = 0; f < ix->columns.size(); f++)
                {
                    const int64_t id = ix->find_term((uint32_t)f, t);
                    st->terms[std::make_pair((uint32_t)f, t)] += id < 0 ? 0 : (uint64_t)(ix->post_off[id + 1] - ix->post_off[id]);
                }
        }
        st->finish();
        *out = st.release();
    });
}

MSVS_HOST_API int msvs_host_fts_statistics_merge(const msvs_bm25_stats_t * const * rows, size_t nrows, msvs_fts_stats_t ** out)
{
    return text_guarded([&] {
        if (!out || (nrows && !rows))
            text_fail(MSVS_ERR_INVALID_ARGUMENT, "null argument");
        std::unique_ptr<msvs_fts_stats> st(new msvs_fts_stats);
        for (size_t r = 0; r < nrows; r++)
        {
            const msvs_bm25_stats_t * row = rows[r];
            if (!row || (row->n_fields && !row->total_num_tokens) || (row->n_docs_freq && !row->docs_freq))
                text_fail(MSVS_ERR_INVALID_ARGUMENT, "malformed statistics row " + std::to_string(r));
            st->total_docs += row->total_num_docs;
            for (size_t i = 0; i < row->n_fields; i++)
                st->tokens[row->total_num_tokens[i].field_id] += row->total_num_tokens[i].field_total_tokens;
            for (size_t i = 0; i < row->n_docs_freq; i++)
            {
                if (!row->docs_freq[i].term)
                    text_fail(MSVS_ERR_INVALID_ARGUMENT, "null term in statistics row " + std::to_string(r));
                st->terms[std::make_pair(row->docs_freq[i].field_id, std::string(row->docs_freq[i].term))] += row->docs_freq[i].doc_freq;
            }
        }
        st->finish();
        *out = st.release();
    });
}

MSVS_HOST_API const msvs_bm25_stats_t * msvs_fts_stats_view(const msvs_fts_stats_t * st) { return st ? &st->view : nullptr; }
MSVS_HOST_API void msvs_fts_stats_free(msvs_fts_stats_t * st) { delete st; }
}
