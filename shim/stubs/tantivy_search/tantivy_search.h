// Stub of the C++ header the tantivy_search crate generates (contrib/tantivy-search is an un-vendored submodule):
// only what the BM25 path of the host uses, reconstructed from the call sites
//   src/Storages/MergeTree/TantivyIndexStore.cpp:853-998   (ffi_* calls and the {result, error{is_error, message}} wrappers)
//   src/Storages/MergeTree/TantivyIndexStore.cpp:713,742,792,824   (the writer: create / index docs / free / commit)
//   src/VectorIndex/Storages/MergeTreeTextSearchManager.cpp:183-267   (RowIdWithScore::row_id / score)
//   src/VectorIndex/Common/BM25InfoInDataParts.cpp:40-93    (DocWithFreq{term_str, field_id, doc_freq}, FieldTokenNums)
//   src/VectorIndex/Utils/ReadWithHybridSearch.cpp:204-206,279-291   (Statistics{docs_freq, total_num_tokens, total_num_docs})
#pragma once
#include <rust/cxx.h>

#include <cstdint>
#include <string>
#include <vector>

namespace TANTIVY
{
struct FFIError
{
    bool is_error = false;
    rust::String message;
};
struct RowIdWithScore
{
    uint64_t row_id = 0;
    float score = 0.f;
    uint32_t seg_id = 0;
    uint32_t doc_id = 0;
    rust::Vec<rust::String> docs;
};
struct DocWithFreq
{
    rust::String term_str;
    uint32_t field_id = 0;
    uint64_t doc_freq = 0;
};
struct FieldTokenNums
{
    uint32_t field_id = 0;
    uint64_t field_total_tokens = 0;
};
struct Statistics
{
    rust::Vec<DocWithFreq> docs_freq;
    rust::Vec<FieldTokenNums> total_num_tokens;
    uint64_t total_num_docs = 0;
};
struct FFIBoolResult
{
    bool result = false;
    FFIError error;
};
struct FFIU64Result
{
    uint64_t result = 0;
    FFIError error;
};
struct FFIVecRowIdWithScoreResult
{
    rust::Vec<RowIdWithScore> result;
    FFIError error;
};
struct FFIVecDocWithFreqResult
{
    rust::Vec<DocWithFreq> result;
    FFIError error;
};
struct FFIFieldTokenNumsResult
{
    rust::Vec<FieldTokenNums> result;
    FFIError error;
};

FFIBoolResult ffi_create_index_with_parameter(const std::string & index_path, const std::vector<std::string> & column_names,
                                              const std::string & index_json_parameter);
FFIBoolResult ffi_index_multi_column_docs(const std::string & index_path, uint64_t row_id, const std::vector<std::string> & column_names,
                                          const std::vector<std::string> & docs);
FFIBoolResult ffi_index_writer_commit(const std::string & index_path);
FFIBoolResult ffi_free_index_writer(const std::string & index_path);
FFIBoolResult ffi_load_index_reader(const std::string & index_path);
FFIBoolResult ffi_free_index_reader(const std::string & index_path);
FFIVecRowIdWithScoreResult ffi_bm25_search(const std::string & index_path, const std::string & sentence,
                                           const std::vector<std::string> & column_names, uint32_t topk,
                                           const std::vector<uint8_t> & u8_alive_bitmap, bool use_filter, bool enable_nlq,
                                           bool operator_or, const Statistics & statistics);
FFIVecDocWithFreqResult ffi_get_doc_freq(const std::string & index_path, const std::string & sentence);
FFIU64Result ffi_get_total_num_docs(const std::string & index_path);
FFIFieldTokenNumsResult ffi_get_total_num_tokens(const std::string & index_path);
FFIU64Result ffi_get_indexed_doc_counts(const std::string & index_path);
}
