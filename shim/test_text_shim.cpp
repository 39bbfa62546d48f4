// test_text_shim.cpp -- drives shim/TextShim.cpp the way TantivyIndexStore / MergeTreeTextSearchManager /
// BM25InfoInDataParts drive tantivy_search: two parts are exported, the table-level statistics are summed over the
// parts (BM25InfoInDataParts.cpp:40-93), every part is searched with them and the hits are printed.
//   usage: test_text_shim <dir>   reads <dir>/docs.txt ("<part>\t<text>" per line, rows in order), <dir>/query.txt
//                                 writes <dir>/part<i>/postings.mspost, prints "part row score" lines + statistics
#include <tantivy_search/tantivy_search.h>

#include <cstdio>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <sys/stat.h>

#include "../include/msvs_host.h"

int main(int argc, char ** argv)
{
    if (argc < 2)
        return 2;
    const std::string dir = argv[1];
    std::ifstream in(dir + "/docs.txt");
    std::string line;
    std::map<int, std::vector<std::string>> parts;
    while (std::getline(in, line))
    {
        const size_t tab = line.find('\t');
        parts[std::stoi(line.substr(0, tab))].push_back(line.substr(tab + 1));
    }
    std::ifstream qf(dir + "/query.txt");
    std::string query, op;
    std::getline(qf, query);
    std::getline(qf, op); // "or" / "and"
    const std::string col = "doc";
    // the part writer's side, through the calls TantivyIndexStore makes (getTantivyIndexWriter :713, indexMultiColumnDoc
    // :742, commitTantivyIndex :824, freeTantivyIndexWriter :792); the shim tees them into <dir>/postings.mspost
    for (auto & kv : parts)
    {
        const std::string pdir = dir + "/part" + std::to_string(kv.first);
        mkdir(pdir.c_str(), 0755);
        auto made = TANTIVY::ffi_create_index_with_parameter(pdir, {col}, "{}");
        if (made.error.is_error || !made.result)
        {
            std::cerr << std::string(made.error.message) << "\n";
            return 3;
        }
        for (size_t r = 0; r < kv.second.size(); r++)
        {
            auto st = TANTIVY::ffi_index_multi_column_docs(pdir, r, {col}, {kv.second[r]});
            if (st.error.is_error || !st.result)
                return 4;
        }
        auto done = TANTIVY::ffi_index_writer_commit(pdir);
        if (done.error.is_error || !done.result)
        {
            std::cerr << std::string(done.error.message) << "\n";
            return 5;
        }
        if (!TANTIVY::ffi_free_index_writer(pdir).result)
            return 5;
    }
    // misuse is an error VALUE: indexing into a writer that was freed, a tokenizer the export does not implement
    std::printf("freed_writer_is_error %d\n", TANTIVY::ffi_index_multi_column_docs(dir + "/part0", 0, {col}, {"x"}).error.is_error ? 1 : 0);
    std::printf("other_tokenizer_is_error %d\n",
                TANTIVY::ffi_create_index_with_parameter(dir + "/nope", {col}, "{\"doc\": {\"tokenizer\": {\"type\": \"ngram\"}}}").error.is_error ? 1 : 0);
    std::printf("default_tokenizer_is_fine %d\n",
                TANTIVY::ffi_create_index_with_parameter(dir + "/nope", {col}, "{\"doc\": {\"tokenizer\": {\"type\": \"default\"}}}").result ? 1 : 0);
    TANTIVY::ffi_free_index_writer(dir + "/nope");
    // the default chain WITH non-default options is another chain: rejected; spelled-out defaults are fine
    std::printf("tokenizer_options_are_errors %d\n",
                (TANTIVY::ffi_create_index_with_parameter(dir + "/nope", {col}, "{\"doc\": {\"tokenizer\": {\"type\": \"default\", \"case_sensitive\": true}}}").error.is_error
                 && TANTIVY::ffi_create_index_with_parameter(dir + "/nope", {col}, "{\"doc\": {\"tokenizer\": {\"type\": \"default\", \"stop_word_filters\": [\"english\"]}}}").error.is_error
                 && TANTIVY::ffi_create_index_with_parameter(dir + "/nope", {col}, "{\"doc\": {\"tokenizer\": {\"type\": \"default\", \"stem_languages\": [\"english\"]}}}").error.is_error
                 && TANTIVY::ffi_create_index_with_parameter(dir + "/nope", {col}, "{\"doc\": {\"tokenizer\": {\"type\": \"default\", \"length_limit\": 60}}}").error.is_error)
                    ? 1 : 0);
    std::printf("default_options_spelled_out_are_fine %d\n",
                TANTIVY::ffi_create_index_with_parameter(dir + "/nope", {col}, "{\"doc\": {\"tokenizer\": {\"type\": \"default\", \"case_sensitive\": false, \"stop_word_filters\": [], \"stem_languages\": [ ], \"length_limit\": 40}}}").result ? 1 : 0);
    TANTIVY::ffi_free_index_writer(dir + "/nope");
    // the host side: statistics over the parts, then one search per part
    TANTIVY::Statistics stats;
    std::map<std::pair<uint32_t, std::string>, uint64_t> df;
    std::map<uint32_t, uint64_t> tokens;
    for (auto & kv : parts)
    {
        const std::string pdir = dir + "/part" + std::to_string(kv.first);
        auto ld = TANTIVY::ffi_load_index_reader(pdir);
        if (ld.error.is_error)
        {
            std::cerr << std::string(ld.error.message) << "\n";
            return 6;
        }
        stats.total_num_docs += TANTIVY::ffi_get_total_num_docs(pdir).result;
        for (auto & t : TANTIVY::ffi_get_total_num_tokens(pdir).result)
            tokens[t.field_id] += t.field_total_tokens;
        for (auto & d : TANTIVY::ffi_get_doc_freq(pdir, query).result)
            df[{d.field_id, std::string(d.term_str)}] += d.doc_freq;
    }
    for (auto & kv : tokens)
        stats.total_num_tokens.push_back({kv.first, kv.second});
    for (auto & kv : df)
    {
        TANTIVY::DocWithFreq d;
        d.term_str = kv.first.second;
        d.field_id = kv.first.first;
        d.doc_freq = kv.second;
        stats.docs_freq.push_back(d);
    }
    std::printf("stats %llu %llu\n", (unsigned long long)stats.total_num_docs, (unsigned long long)tokens[0]);
    for (auto & kv : parts)
    {
        const std::string pdir = dir + "/part" + std::to_string(kv.first);
        auto res = TANTIVY::ffi_bm25_search(pdir, query, {"doc"}, 5, {}, false, false, op != "and", stats);
        if (res.error.is_error)
        {
            std::cerr << std::string(res.error.message) << "\n";
            return 7;
        }
        for (auto & h : res.result)
            std::printf("hit %d %llu %.9g\n", kv.first, (unsigned long long)h.row_id, (double)h.score);
        // with a filter that only lets even rows through
        std::vector<uint8_t> even((kv.second.size() + 7) / 8, 0x55);
        auto fr = TANTIVY::ffi_bm25_search(pdir, query, {"doc"}, 5, even, true, false, op != "and", stats);
        for (auto & h : fr.result)
            std::printf("even %d %llu %.9g\n", kv.first, (unsigned long long)h.row_id, (double)h.score);
        TANTIVY::ffi_free_index_reader(pdir);
    }
    auto bad = TANTIVY::ffi_bm25_search(dir + "/nowhere", query, {"doc"}, 5, {}, false, false, true, stats);
    std::printf("missing_index_is_error %d\n", bad.error.is_error ? 1 : 0);
    return 0;
}
