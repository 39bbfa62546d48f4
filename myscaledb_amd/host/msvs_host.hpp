// msvs_host.hpp -- host-side mirror of the reference's operator interface for the hot path, written against
// libmsvs.so.  Names, argument meaning and error behaviour follow the reference so that a maintainer can map each
// piece onto the file it replaces (citations are paths in the MyScaleDB tree); the implementation is new.
//
//   VectorIndex::tryBruteForceSearch<FloatVector>   src/VectorIndex/Common/BruteForceSearch.h:63-92
//   VIWithColumnInPart::searchWithoutIndex          src/VectorIndex/Common/VIWithDataPart.h:341-382
//   MergeTreeVSManager::searchWrapper               src/VectorIndex/Storages/MergeTreeVSManager.cpp:1537-1679
//   MergeTreeBaseSearchManager::getTotalTopSearchResultImpl   ...MergeTreeBaseSearchManager.cpp:207-299
//   RankFusion / RelativeScoreFusion / hybridSearch src/VectorIndex/Utils/HybridSearchUtils.cpp:164-314,
//                                                   src/VectorIndex/Storages/MergeTreeHybridSearchManager.cpp:108-171
#pragma once

#include <cstdint>
#include <limits>
#include <map>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/msvs.h"

namespace VectorIndex
{

enum class VIMetric
{
    L2 = MSVS_METRIC_L2,
    IP = MSVS_METRIC_IP,
    Cosine = MSVS_METRIC_COSINE
};

/// DB::Exception stand-in: carries the msvs status code (NOT_IMPLEMENTED, ...) and message.
struct VIException : std::runtime_error
{
    int code;
    VIException(int c, const std::string & m) : std::runtime_error(m), code(c) {}
};

inline void throwIfError(int rc)
{
    if (rc != MSVS_OK)
        throw VIException(rc, msvs_last_error());
}

/// Same signature as the reference (raw caller-owned buffers of nx*k results, -1 = unfilled).
inline void tryBruteForceSearch(const float * x, const float * y, size_t d, size_t k, size_t nx, size_t ny,
                                int64_t * result_id, float * distance, const VIMetric & metric_type)
{
    if (metric_type != VIMetric::IP && metric_type != VIMetric::L2)
        throw VIException(MSVS_ERR_NOT_IMPLEMENTED, "Metric not implemented in brute force search for Float32 Vector");
    throwIfError(msvs_knn_f32(x, y, d, k, nx, ny, static_cast<int>(metric_type), result_id, distance));
}

/// Owning/non-owning row-major float dataset with the reference's normalize() semantics executed on the GPU.
struct VectorDataset
{
    float * data;
    int64_t total_vectors;
    int64_t dimension;
    void normalize() { throwIfError(msvs_normalize_f32(data, static_cast<size_t>(total_vectors), static_cast<size_t>(dimension))); }
};

struct VIWithColumnInPart
{
    static void searchWithoutIndex(VectorDataset & query_data, VectorDataset & base_data, int32_t k, float * distances,
                                   int64_t * labels, const VIMetric & metric);
};

}

namespace DB
{

using VIMetric = VectorIndex::VIMetric;

/// LSB-first bitmap view (1 = row exists / passes the filter).
struct VIBitmapView
{
    const uint64_t * words = nullptr;
    bool is_member(size_t i) const { return !words || ((words[i >> 6] >> (i & 63)) & 1); }
};

/// ColumnArray(Float32) of one data part as the MergeTree reader hands it over: offsets[i] = end of row i in `data`
/// (ClickHouse ColumnArray layout; an empty Array has offsets[i] == offsets[i-1]).
struct ColumnArrayView
{
    const uint64_t * offsets = nullptr;
    const float * data = nullptr;
    size_t rows = 0;
};

/// (label, [query id,] distance) result columns of one part: CommonSearchResult::result_columns
/// (ColumnUInt32 label, ColumnUInt32 vector_id for batch_distance, ColumnFloat32 distance).
struct VectorScanResult
{
    std::vector<uint32_t> labels;
    std::vector<uint32_t> query_ids; // only for batch_distance
    std::vector<float> distances;
    bool computed = false;
};

struct MergeTreeVSManager
{
    static void searchWrapper(bool prewhere, VectorIndex::VectorDataset & query_vector,
                              VectorIndex::VectorDataset & base_data, int k, int dim, int nq, int num_rows_read,
                              std::vector<int64_t> & final_id, std::vector<float> & final_distance,
                              const std::vector<size_t> & actual_id_in_range, const VIMetric & metric,
                              const VIBitmapView & row_exists, int delete_id_num);

    /// The block-vs-running two-way merge at the end of searchWrapper (MergeTreeVSManager.cpp:1652-1678).
    static void mergeBlockResult(const std::vector<int64_t> & per_id, const std::vector<float> & per_distance, int k, int nq,
                                 int num_rows_read, std::vector<int64_t> & final_id, std::vector<float> & final_distance,
                                 const VIMetric & metric);

    /// getQueryVector / getFloatQueryVectorInBatch (MergeTreeVSManager.cpp:59-181): Array(Float32|Float64) query
    /// constants -> row-major f32; a row whose length differs from `dim` is an error like in the reference.
    static std::vector<float> generateVectorDataset(const void * values, bool is_float64, const uint64_t * offsets,
                                                    size_t nq, size_t dim);

    /// vectorScanWithoutIndex<FloatVector> (MergeTreeVSManager.cpp:959-1535): brute force over one part, mark by
    /// mark.  filter == nullptr: dense blocks (empty rows padded with FLT_MAX), lightweight-deleted rows handled by
    /// row_exists; filter != nullptr: only the passing, non-empty rows of each mark are compacted and searched
    /// (the filter already includes the lightweight deletes).  Returns the part's result columns.
    static VectorScanResult vectorScanWithoutIndex(const ColumnArrayView & column, size_t dim, size_t index_granularity,
                                                   const std::vector<float> & queries, size_t nq, int k,
                                                   const VIMetric & metric, bool is_batch, const VIBitmapView * filter,
                                                   const VIBitmapView * row_exists);

    /// The same scan over RESIDENT blocks (msvs_cache_t): the dense block of a mark is uploaded the
    /// first time any query touches the part (keyed by part_key / mark) and searched in HBM afterwards; lightweight
    /// deletes and PREWHERE filters become the search's row bitmap over the resident block instead of a per-query
    /// compaction + upload.  Same result columns as vectorScanWithoutIndex.
    static VectorScanResult vectorScanWithoutIndexResident(msvs_cache_t * cache, const std::string & part_key,
                                                           const ColumnArrayView & column, size_t dim,
                                                           size_t index_granularity, const std::vector<float> & queries,
                                                           size_t nq, int k, const VIMetric & metric, bool is_batch,
                                                           const VIBitmapView * filter, const VIBitmapView * row_exists);
};

/// MergeTreeBaseSearchManager::mergeSearchResultImpl (MergeTreeBaseSearchManager.cpp:23-164), reduced to its join:
/// for every read row (identified by its _part_offset) the position of its label in the part's result, or -1.
std::vector<int64_t> mergeSearchResult(const std::vector<uint64_t> & part_offsets, const std::vector<uint32_t> & labels);

/// Search::intersectDenseBitmaps as used in VIWithDataPart.cpp:903-908: filter AND delete bitmap (word-wise).
std::vector<uint64_t> intersectDenseBitmaps(const std::vector<uint64_t> & a, const std::vector<uint64_t> & b);

struct ScoreWithPartIndexAndLabel
{
    float score = 0;
    uint64_t part_index = 0;
    uint64_t label_id = 0;
    uint32_t shard_num = 0;
};
using ScoreWithPartIndexAndLabels = std::vector<ScoreWithPartIndexAndLabel>;

struct MergeTreeBaseSearchManager
{
    /// `all` = every part's (score, part_index, label) in part order then rank order.
    static ScoreWithPartIndexAndLabels getTotalTopSearchResultImpl(const ScoreWithPartIndexAndLabels & all,
                                                                   uint64_t top_k, bool desc_direction);
};

void RankFusion(std::map<std::tuple<uint32_t, uint64_t, uint64_t>, float> & fusion_id_with_score,
                const ScoreWithPartIndexAndLabels & vec_scan_result_dataset,
                const ScoreWithPartIndexAndLabels & text_search_result_dataset, uint64_t fusion_k);

void RelativeScoreFusion(std::map<std::tuple<uint32_t, uint64_t, uint64_t>, float> & fusion_id_with_score,
                         const ScoreWithPartIndexAndLabels & vec_scan_result_dataset,
                         const ScoreWithPartIndexAndLabels & text_search_result_dataset, float fusion_weight,
                         int8_t vector_scan_direction);

void computeNormalizedScore(const ScoreWithPartIndexAndLabels & search_result_dataset, std::vector<float> & norm_score);

struct HybridSearchInfo
{
    std::string fusion_type = "rsf"; // "rsf" | "rrf"
    int fusion_k = 60;
    float fusion_weight = 0.5f;
    int topk = 0;
    int vector_scan_direction = 1;
};

/// HybridSearchFusionTransform::generate (src/VectorIndex/Processors/HybridSearchFusionTransform.cpp:22-182): the fusion step
/// of a Distributed-table hybrid search on the initiator.  `rows` are the merged shard results in pipeline order -- the
/// distance rows (score_type 0) first, then the bm25 rows (score_type 1).  At most num_candidates rows of each kind take part
/// (direction 1: the LAST num_candidates distance rows, read backwards, i.e. best first).  Returns (index into rows, fused
/// score): every bm25 row in order, then the distance rows that are not among them.
struct FusionRow
{
    float score = 0;
    uint8_t score_type = 0;
    uint32_t shard_num = 0;
    uint64_t part_index = 0, part_offset = 0;
};
std::vector<std::pair<size_t, float>> hybridSearchFusionTransform(const std::vector<FusionRow> & rows, uint64_t num_candidates,
                                                                  const HybridSearchInfo & info);

struct MergeTreeHybridSearchManager
{
    static ScoreWithPartIndexAndLabels hybridSearch(const ScoreWithPartIndexAndLabels & vec_scan_result_with_part_index,
                                                    const ScoreWithPartIndexAndLabels & text_search_result_with_part_index,
                                                    const HybridSearchInfo & hybrid_info);
};

}
