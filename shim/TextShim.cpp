// TextShim.cpp -- seam B of INTEGRATION.md as code: the TANTIVY::ffi_* functions the BM25 path of the host calls
// (TantivyIndexStore.cpp:853-998), implemented on the postings export + device scorer (libmsvs_host.so: text_store.cpp,
// libmsvs.so: bm25.hip) instead of the Rust library.  Compiled against shim/stubs/tantivy_search/tantivy_search.h, which
// restates the generated header from its call sites.  An index "directory" is looked up as <index_path>/postings.mspost
// (the export file written next to the tantivy files); readers are cached by path like the Rust side caches its
// IndexReaderBridge (ffi_load_index_reader / ffi_free_index_reader).  Errors travel by value in .error, never as
// exceptions -- the host turns them into TANTIVY_SEARCH_INTERNAL_ERROR (TantivyIndexStore.cpp:919-923).
//
// The writer side is a TEE (TantivyIndexStore.cpp:713 create, :742 index, :824 commit, :792 free): the same four calls
// the part writer already makes build the postings export next to whatever else the directory holds, so no host code
// changes to get <index_path>/postings.mspost written.  Built with -DMSVS_TANTIVY_FORWARD_NS=<ns> every writer call is
// ALSO forwarded to <ns>::ffi_* (the Rust crate's bridge regenerated under another namespace, for a build that keeps
// tantivy's own files for the boolean FTS functions); without it this library stands alone.
#include <tantivy_search/tantivy_search.h>

#include <algorithm>
#include <map>
#include <memory>
#include <mutex>

#include "../include/msvs_host.h"

namespace
{
struct Reader
{
    msvs_text_index_t * ix = nullptr;
    ~Reader() { msvs_text_index_free(ix); }
};
std::mutex g_mu;
std::map<std::string, std::shared_ptr<Reader>> g_readers;

struct Writer
{
    msvs_text_index_t * ix = nullptr;
    std::mutex mu; // the part writer is single-threaded per index; this only keeps a stray concurrent commit honest
    ~Writer() { msvs_text_index_free(ix); }
};
std::map<std::string, std::shared_ptr<Writer>> g_writers;

std::shared_ptr<Writer> writer_of(const std::string & path)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_writers.find(path);
    return it == g_writers.end() ? nullptr : it->second;
}

std::shared_ptr<Reader> reader_of(const std::string & path, std::string & err)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_readers.find(path);
    if (it != g_readers.end())
        return it->second;
    auto r = std::make_shared<Reader>();
    if (msvs_text_index_load((path + "/postings.mspost").c_str(), &r->ix) != 0)
    {
        err = msvs_text_last_error();
        return nullptr;
    }
    g_readers[path] = r;
    return r;
}

template <typename R>
R failed(const std::string & m)
{
    R r;
    r.error.is_error = true;
    r.error.message = m;
    return r;
}
}

#ifdef MSVS_TANTIVY_FORWARD_NS
namespace MSVS_TANTIVY_FORWARD_NS
{
TANTIVY::FFIBoolResult ffi_create_index_with_parameter(const std::string &, const std::vector<std::string> &, const std::string &);
TANTIVY::FFIBoolResult ffi_index_multi_column_docs(const std::string &, uint64_t, const std::vector<std::string> &,
                                                   const std::vector<std::string> &);
TANTIVY::FFIBoolResult ffi_index_writer_commit(const std::string &);
TANTIVY::FFIBoolResult ffi_free_index_writer(const std::string &);
}
#define MSVS_TEE(call) \
    do \
    { \
        TANTIVY::FFIBoolResult fw = MSVS_TANTIVY_FORWARD_NS::call; \
        if (fw.error.is_error || !fw.result) \
            return fw; \
    } while (0)
#else
#define MSVS_TEE(call) \
    do \
    { \
    } while (0)
#endif

namespace TANTIVY
{
// ---- the writer side: TantivyIndexStore::getTantivyIndexWriter / indexMultiColumnDoc / commitTantivyIndex / freeTantivyIndexWriter
FFIBoolResult ffi_create_index_with_parameter(const std::string & index_path, const std::vector<std::string> & column_names,
                                              const std::string & index_json_parameter)
{
    MSVS_TEE(ffi_create_index_with_parameter(index_path, column_names, index_json_parameter));
    // index_json_parameter ({"<column>": {"tokenizer": {"type": "<name>", ...}}}, MergeTreeIndexTantivy.cpp:795) picks a
    // tokenizer per column; the export implements tantivy's default chain only, and a column on another tokenizer must
    // fail here rather than score differently later
    for (size_t at = index_json_parameter.find("\"type\""); at != std::string::npos; at = index_json_parameter.find("\"type\"", at + 6))
    {
        const size_t q0 = index_json_parameter.find('"', index_json_parameter.find(':', at + 6));
        const size_t q1 = q0 == std::string::npos ? q0 : index_json_parameter.find('"', q0 + 1);
        if (q1 == std::string::npos || index_json_parameter.substr(q0 + 1, q1 - q0 - 1) != "default")
            return failed<FFIBoolResult>("msvs text export: only tantivy's default tokenizer chain is implemented, got " + index_json_parameter);
    }
    // ... and the default chain WITH options (tantivy_search documents case_sensitive, stop_word_filters, stem_languages,
    // length_limit for it) is another chain: anything but their defaults fails here too
    {
        const std::string & j = index_json_parameter;
        auto value_after = [&](const char * key, size_t from) -> std::string {
            const size_t at = j.find(key, from);
            if (at == std::string::npos)
                return "";
            size_t b = j.find(':', at);
            if (b == std::string::npos)
                return "";
            b++;
            while (b < j.size() && (j[b] == ' ' || j[b] == '\t' || j[b] == '\n'))
                b++;
            size_t e = b;
            if (e < j.size() && j[e] == '[')
                e = j.find(']', e) == std::string::npos ? j.size() : j.find(']', e) + 1;
            else
                while (e < j.size() && j[e] != ',' && j[e] != '}')
                    e++;
            std::string v = j.substr(b, e - b);
            v.erase(std::remove_if(v.begin(), v.end(), [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '"'; }), v.end());
            return v;
        };
        for (size_t from = 0; j.find("\"case_sensitive\"", from) != std::string::npos; from = j.find("\"case_sensitive\"", from) + 1)
            if (value_after("\"case_sensitive\"", from) == "true")
                return failed<FFIBoolResult>("msvs text export: tokenizer option case_sensitive = true is not implemented (" + j + ")");
        for (const char * key : {"\"stop_word_filters\"", "\"stem_languages\""})
            for (size_t from = 0; j.find(key, from) != std::string::npos; from = j.find(key, from) + 1)
            {
                const std::string v = value_after(key, from);
                if (!v.empty() && v != "[]")
                    return failed<FFIBoolResult>(std::string("msvs text export: tokenizer option ") + key + " is not implemented (" + j + ")");
            }
        for (size_t from = 0; j.find("\"length_limit\"", from) != std::string::npos; from = j.find("\"length_limit\"", from) + 1)
            if (value_after("\"length_limit\"", from) != "40")
                return failed<FFIBoolResult>("msvs text export: tokenizer option length_limit other than 40 is not implemented (" + j + ")");
    }
    std::vector<const char *> cols;
    for (const auto & c : column_names)
        cols.push_back(c.c_str());
    auto w = std::make_shared<Writer>();
    if (msvs_text_index_create(cols.empty() ? nullptr : cols.data(), cols.size(), &w->ix) != 0)
        return failed<FFIBoolResult>(msvs_text_last_error());
    std::lock_guard<std::mutex> lk(g_mu);
    g_writers[index_path] = w; // like the Rust side: creating again over a live writer replaces it
    g_readers.erase(index_path); // a reader of the previous contents would be stale
    FFIBoolResult r;
    r.result = true;
    return r;
}

FFIBoolResult ffi_index_multi_column_docs(const std::string & index_path, uint64_t row_id, const std::vector<std::string> & column_names,
                                          const std::vector<std::string> & docs)
{
    MSVS_TEE(ffi_index_multi_column_docs(index_path, row_id, column_names, docs));
    auto w = writer_of(index_path);
    if (!w)
        return failed<FFIBoolResult>("msvs text export: no index writer for " + index_path);
    if (column_names.size() != docs.size())
        return failed<FFIBoolResult>("msvs text export: column_names and docs differ in length");
    std::vector<const char *> cols, texts;
    for (size_t i = 0; i < docs.size(); i++)
    {
        cols.push_back(column_names[i].c_str());
        texts.push_back(docs[i].c_str());
    }
    std::lock_guard<std::mutex> lk(w->mu);
    if (msvs_text_index_add_doc(w->ix, row_id, cols.data(), texts.data(), docs.size()) != 0)
        return failed<FFIBoolResult>(msvs_text_last_error());
    FFIBoolResult r;
    r.result = true;
    return r;
}

FFIBoolResult ffi_index_writer_commit(const std::string & index_path)
{
    MSVS_TEE(ffi_index_writer_commit(index_path));
    auto w = writer_of(index_path);
    if (!w)
        return failed<FFIBoolResult>("msvs text export: no index writer for " + index_path);
    std::lock_guard<std::mutex> lk(w->mu);
    if (msvs_text_index_commit(w->ix) != 0 || msvs_text_index_save(w->ix, (index_path + "/postings.mspost").c_str()) != 0)
        return failed<FFIBoolResult>(msvs_text_last_error());
    {
        std::lock_guard<std::mutex> lk2(g_mu);
        g_readers.erase(index_path); // the next ffi_load_index_reader sees what was just committed
    }
    FFIBoolResult r;
    r.result = true;
    return r;
}

FFIBoolResult ffi_free_index_writer(const std::string & index_path)
{
    MSVS_TEE(ffi_free_index_writer(index_path));
    std::lock_guard<std::mutex> lk(g_mu);
    g_writers.erase(index_path); // freeing a writer that is not there is not an error (freeTantivyIndexWriter is idempotent)
    FFIBoolResult r;
    r.result = true;
    return r;
}

// ---- the reader side
FFIBoolResult ffi_load_index_reader(const std::string & index_path)
{
    std::string err;
    if (!reader_of(index_path, err))
        return failed<FFIBoolResult>(err);
    FFIBoolResult r;
    r.result = true;
    return r;
}

FFIBoolResult ffi_free_index_reader(const std::string & index_path)
{
    std::lock_guard<std::mutex> lk(g_mu);
    FFIBoolResult r;
    r.result = g_readers.erase(index_path) > 0;
    return r;
}

FFIVecRowIdWithScoreResult ffi_bm25_search(const std::string & index_path, const std::string & sentence,
                                           const std::vector<std::string> & column_names, uint32_t topk,
                                           const std::vector<uint8_t> & u8_alive_bitmap, bool use_filter, bool enable_nlq,
                                           bool operator_or, const Statistics & statistics)
{
    std::string err;
    auto rd = reader_of(index_path, err);
    if (!rd)
        return failed<FFIVecRowIdWithScoreResult>(err);
    std::vector<msvs_doc_freq_t> df(statistics.docs_freq.size());
    for (size_t i = 0; i < df.size(); i++)
        df[i] = {statistics.docs_freq[i].term_str.c_str(), statistics.docs_freq[i].field_id, statistics.docs_freq[i].doc_freq};
    std::vector<msvs_field_tokens_t> tk(statistics.total_num_tokens.size());
    for (size_t i = 0; i < tk.size(); i++)
        tk[i] = {statistics.total_num_tokens[i].field_id, statistics.total_num_tokens[i].field_total_tokens};
    msvs_bm25_stats_t st{df.data(), df.size(), tk.data(), tk.size(), statistics.total_num_docs};
    std::vector<const char *> cols;
    for (const auto & c : column_names)
        cols.push_back(c.c_str());
    std::vector<uint64_t> rows(topk);
    std::vector<float> scores(topk);
    uint32_t n = 0;
    const int rc = msvs_text_index_bm25_search(rd->ix, sentence.c_str(), cols.empty() ? nullptr : cols.data(), cols.size(), topk,
                                               u8_alive_bitmap.data(), u8_alive_bitmap.size(), use_filter ? 1 : 0, enable_nlq ? 1 : 0,
                                               operator_or ? 1 : 0, &st, rows.data(), scores.data(), &n);
    if (rc != 0)
        return failed<FFIVecRowIdWithScoreResult>(msvs_text_last_error());
    FFIVecRowIdWithScoreResult r;
    r.result.resize(n);
    for (uint32_t i = 0; i < n; i++)
    {
        r.result[i].row_id = rows[i];
        r.result[i].score = scores[i];
        r.result[i].doc_id = (uint32_t)rows[i]; // one segment per part: the row offset is the doc id
    }
    return r;
}

FFIVecDocWithFreqResult ffi_get_doc_freq(const std::string & index_path, const std::string & sentence)
{
    std::string err;
    auto rd = reader_of(index_path, err);
    if (!rd)
        return failed<FFIVecDocWithFreqResult>(err);
    // every (token, column) pair of the sentence: the callee reports how many there are; a second call takes them all
    std::vector<msvs_doc_freq_t> out(256);
    size_t n = 0;
    if (msvs_text_index_doc_freq(rd->ix, sentence.c_str(), out.data(), out.size(), &n) != 0)
        return failed<FFIVecDocWithFreqResult>(msvs_text_last_error());
    if (n > out.size())
    {
        out.resize(n);
        if (msvs_text_index_doc_freq(rd->ix, sentence.c_str(), out.data(), out.size(), &n) != 0)
            return failed<FFIVecDocWithFreqResult>(msvs_text_last_error());
    }
    FFIVecDocWithFreqResult r;
    for (size_t i = 0; i < n && i < out.size(); i++)
    {
        DocWithFreq d;
        d.term_str = std::string(out[i].term);
        d.field_id = out[i].field_id;
        d.doc_freq = out[i].doc_freq;
        r.result.push_back(d);
    }
    return r;
}

FFIU64Result ffi_get_total_num_docs(const std::string & index_path)
{
    std::string err;
    auto rd = reader_of(index_path, err);
    if (!rd)
        return failed<FFIU64Result>(err);
    FFIU64Result r;
    r.result = msvs_text_index_total_num_docs(rd->ix);
    return r;
}

FFIU64Result ffi_get_indexed_doc_counts(const std::string & index_path) { return ffi_get_total_num_docs(index_path); }

FFIFieldTokenNumsResult ffi_get_total_num_tokens(const std::string & index_path)
{
    std::string err;
    auto rd = reader_of(index_path, err);
    if (!rd)
        return failed<FFIFieldTokenNumsResult>(err);
    std::vector<msvs_field_tokens_t> out(8);
    size_t n = 0;
    if (msvs_text_index_total_num_tokens(rd->ix, out.data(), out.size(), &n) != 0)
        return failed<FFIFieldTokenNumsResult>(msvs_text_last_error());
    if (n > out.size()) // more than eight text columns: take them all
    {
        out.resize(n);
        if (msvs_text_index_total_num_tokens(rd->ix, out.data(), out.size(), &n) != 0)
            return failed<FFIFieldTokenNumsResult>(msvs_text_last_error());
    }
    FFIFieldTokenNumsResult r;
    for (size_t i = 0; i < n && i < out.size(); i++)
    {
        FieldTokenNums f;
        f.field_id = out[i].field_id;
        f.field_total_tokens = out[i].field_total_tokens;
        r.result.push_back(f);
    }
    return r;
}
}
