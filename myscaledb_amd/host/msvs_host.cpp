// msvs_host.cpp -- implementation of the host mirror (see msvs_host.hpp) + its C entry points (include/msvs_host.h).
#include "msvs_host.hpp"

#include <algorithm>
#include <cstring>
#include <functional>

#include "../../include/msvs_host.h"

namespace VectorIndex
{

void VIWithColumnInPart::searchWithoutIndex(VectorDataset & query_data, VectorDataset & base_data, int32_t k,
                                            float * distances, int64_t * labels, const VIMetric & metric)
{
    VIMetric new_metric = metric;
    if (metric == VIMetric::Cosine)
    {
        // cosine = normalise both sides, search by inner product, report 1 - ip
        new_metric = VIMetric::IP;
        query_data.normalize();
        base_data.normalize();
    }
    tryBruteForceSearch(query_data.data, base_data.data, static_cast<size_t>(query_data.dimension), static_cast<size_t>(k),
                        static_cast<size_t>(query_data.total_vectors), static_cast<size_t>(base_data.total_vectors), labels,
                        distances, new_metric);
    if (metric == VIMetric::Cosine)
        for (int64_t i = 0; i < static_cast<int64_t>(k) * query_data.total_vectors; i++)
            distances[i] = 1 - distances[i];
}

}

namespace DB
{

void MergeTreeVSManager::searchWrapper(bool prewhere, VectorIndex::VectorDataset & query_vector,
                                       VectorIndex::VectorDataset & base_data, int k, int /*dim*/, int nq,
                                       int num_rows_read, std::vector<int64_t> & final_id,
                                       std::vector<float> & final_distance,
                                       const std::vector<size_t> & actual_id_in_range, const VIMetric & metric,
                                       const VIBitmapView & row_exists, int delete_id_num)
{
    // "worst" sentinels of the block result.  IP deliberately uses numeric_limits<float>::min() (smallest positive
    // normal), reproducing the reference: inner products <= 1.18e-38 never displace an empty slot in the merge.
    const float worst = metric == VIMetric::IP ? std::numeric_limits<float>::min() : std::numeric_limits<float>::max();
    const size_t slots = static_cast<size_t>(k) * nq;
    std::vector<float> per_distance(slots, worst);
    std::vector<int64_t> per_id(slots, -1);

    if (delete_id_num > 0 && static_cast<size_t>(k) + delete_id_num > MSVS_MAX_K && row_exists.words)
    {
        // Many deleted rows in the block: instead of over-fetching k + delete_id_num results and dropping the dead
        // ones afterwards, let the scan skip them (msvs_knn_f32_filtered) -- the same top-k of the alive rows,
        // without a top-(k + thousands) selection.
        VIMetric m = metric;
        if (metric == VIMetric::Cosine)
        {
            m = VIMetric::IP;
            query_vector.normalize();
            base_data.normalize();
        }
        VectorIndex::throwIfError(msvs_knn_f32_filtered(
            query_vector.data, base_data.data, static_cast<size_t>(base_data.dimension), static_cast<size_t>(k),
            static_cast<size_t>(nq), static_cast<size_t>(base_data.total_vectors), static_cast<int>(m), row_exists.words,
            per_id.data(), per_distance.data()));
        for (size_t i = 0; i < slots; i++)
        {
            if (metric == VIMetric::Cosine)
                per_distance[i] = 1 - per_distance[i];
            if (per_id[i] < 0)
                per_distance[i] = worst; // unfilled slots keep the wrapper's own sentinel, like the reference
        }
    }
    else if (delete_id_num > 0)
    {
        // over-fetch by the number of lightweight-deleted rows of the block, then drop the deleted ones
        const size_t kk = static_cast<size_t>(k) + delete_id_num;
        std::vector<float> wide_distance(kk * nq, worst);
        std::vector<int64_t> wide_id(kk * nq, -1);
        VectorIndex::VIWithColumnInPart::searchWithoutIndex(query_vector, base_data, static_cast<int32_t>(kk),
                                                            wide_distance.data(), wide_id.data(), metric);
        for (int q = 0; q < nq; q++)
        {
            size_t kept = 0;
            for (size_t j = 0; j < kk && kept < static_cast<size_t>(k); j++)
            {
                const int64_t id = wide_id[q * kk + j];
                if (id >= 0 && row_exists.is_member(static_cast<size_t>(id)))
                {
                    per_id[q * k + kept] = id;
                    per_distance[q * k + kept] = wide_distance[q * kk + j];
                    kept++;
                }
            }
        }
    }
    else
    {
        VectorIndex::VIWithColumnInPart::searchWithoutIndex(query_vector, base_data, k, per_distance.data(), per_id.data(),
                                                            metric);
    }

    if (prewhere)
        for (auto & id : per_id)
            if (id > -1)
                id = static_cast<int64_t>(actual_id_in_range[static_cast<size_t>(id)]);
    mergeBlockResult(per_id, per_distance, k, nq, num_rows_read, final_id, final_distance, metric);
}

void MergeTreeVSManager::mergeBlockResult(const std::vector<int64_t> & per_id, const std::vector<float> & per_distance, int k,
                                          int nq, int num_rows_read, std::vector<int64_t> & final_id,
                                          std::vector<float> & final_distance, const VIMetric & metric)
{
    // two-way merge of the sorted block result into the sorted running result; strict comparison, so on equal
    // distances the running (earlier block) entry is kept first
    const size_t slots = static_cast<size_t>(k) * nq;
    std::vector<float> merged_distance(slots);
    std::vector<int64_t> merged_id(slots);
    const bool larger_is_better = metric == VIMetric::IP;
    for (int q = 0; q < nq; q++)
    {
        size_t run = static_cast<size_t>(q) * k, blk = run, dst = run;
        for (int i = 0; i < k; i++, dst++)
        {
            const bool take_block = larger_is_better ? final_distance[run] < per_distance[blk]
                                                     : final_distance[run] > per_distance[blk];
            if (take_block)
            {
                merged_distance[dst] = per_distance[blk];
                merged_id[dst] = per_id[blk] + num_rows_read;
                blk++;
            }
            else
            {
                merged_distance[dst] = final_distance[run];
                merged_id[dst] = final_id[run];
                run++;
            }
        }
    }
    final_distance.swap(merged_distance);
    final_id.swap(merged_id);
}

std::vector<float> MergeTreeVSManager::generateVectorDataset(const void * values, bool is_float64,
                                                             const uint64_t * offsets, size_t nq, size_t dim)
{
    std::vector<float> out(nq * dim);
    uint64_t begin = 0;
    for (size_t q = 0; q < nq; q++)
    {
        const uint64_t end = offsets ? offsets[q] : (q + 1) * dim;
        if (end - begin != dim)
            throw VectorIndex::VIException(MSVS_ERR_INVALID_ARGUMENT,
                                           "Dimension is not equal: query: " + std::to_string(end - begin) + " vs search column: "
                                               + std::to_string(dim));
        for (size_t j = 0; j < dim; j++)
            out[q * dim + j] = is_float64 ? static_cast<float>(static_cast<const double *>(values)[begin + j])
                                          : static_cast<const float *>(values)[begin + j];
        begin = end;
    }
    return out;
}

VectorScanResult MergeTreeVSManager::vectorScanWithoutIndex(const ColumnArrayView & column, size_t dim,
                                                            size_t index_granularity, const std::vector<float> & queries,
                                                            size_t nq, int k, const VIMetric & metric, bool is_batch,
                                                            const VIBitmapView * filter, const VIBitmapView * row_exists)
{
    const float worst = metric == VIMetric::IP ? std::numeric_limits<float>::min() : std::numeric_limits<float>::max();
    std::vector<float> final_distance(static_cast<size_t>(k) * nq, worst);
    std::vector<int64_t> final_id(static_cast<size_t>(k) * nq, -1);
    std::vector<float> query_copy;   // searchWithoutIndex normalises its inputs in place (cosine)
    std::vector<float> block;
    std::vector<size_t> actual_id_in_range;
    const size_t total_rows = column.rows;

    auto row_begin = [&](size_t r) { return r == 0 ? 0 : column.offsets[r - 1]; };

    for (size_t mark_start = 0; mark_start < total_rows; mark_start += index_granularity)
    {
        const size_t mark_end = std::min(total_rows, mark_start + index_granularity);
        block.clear();
        actual_id_in_range.clear();
        std::vector<uint64_t> exists_words;
        int deleted_row_num = 0;
        size_t block_rows = 0;
        if (filter)
        {
            // only rows that pass the filter AND carry a vector are searched; their part offsets are remembered
            for (size_t r = mark_start; r < mark_end; r++)
            {
                if (!filter->is_member(r))
                    continue;
                const uint64_t b = row_begin(r), e = column.offsets[r];
                if (b == e)
                    continue;
                block.insert(block.end(), column.data + b, column.data + e);
                actual_id_in_range.push_back(r);
                block_rows++;
            }
        }
        else
        {
            // dense block: a row without a vector is padded with FLT_MAX (its distance overflows and never wins)
            block.assign((mark_end - mark_start) * dim, std::numeric_limits<float>::max());
            block_rows = mark_end - mark_start;
            exists_words.assign((block_rows + 63) / 64, ~0ull);
            for (size_t r = mark_start; r < mark_end; r++)
            {
                const uint64_t b = row_begin(r), e = column.offsets[r];
                if (e - b == dim)
                    std::copy(column.data + b, column.data + e, block.begin() + (r - mark_start) * dim);
                if (row_exists && !row_exists->is_member(r))
                {
                    exists_words[(r - mark_start) >> 6] &= ~(1ull << ((r - mark_start) & 63));
                    deleted_row_num++;
                }
            }
        }
        query_copy = queries;
        VectorIndex::VectorDataset q{query_copy.data(), static_cast<int64_t>(nq), static_cast<int64_t>(dim)};
        VectorIndex::VectorDataset base{block.data(), static_cast<int64_t>(block_rows), static_cast<int64_t>(dim)};
        VIBitmapView exists_view{exists_words.empty() ? nullptr : exists_words.data()};
        searchWrapper(filter != nullptr, q, base, k, static_cast<int>(dim), static_cast<int>(nq),
                      filter ? 0 : static_cast<int>(mark_start), final_id, final_distance, actual_id_in_range, metric,
                      exists_view, deleted_row_num);
    }

    // result columns: only filled slots (id > -1); batch_distance adds the query id = slot / k
    VectorScanResult res;
    for (size_t slot = 0; slot < static_cast<size_t>(k) * nq; slot++)
    {
        if (final_id[slot] <= -1)
            continue;
        res.labels.push_back(static_cast<uint32_t>(final_id[slot]));
        if (is_batch)
            res.query_ids.push_back(static_cast<uint32_t>(slot / static_cast<size_t>(k)));
        res.distances.push_back(final_distance[slot]);
    }
    res.computed = true;
    return res;
}

VectorScanResult MergeTreeVSManager::vectorScanWithoutIndexResident(msvs_cache_t * cache, const std::string & part_key,
                                                                    const ColumnArrayView & column, size_t dim,
                                                                    size_t index_granularity, const std::vector<float> & queries,
                                                                    size_t nq, int k, const VIMetric & metric, bool is_batch,
                                                                    const VIBitmapView * filter, const VIBitmapView * row_exists)
{
    const float worst = metric == VIMetric::IP ? std::numeric_limits<float>::min() : std::numeric_limits<float>::max();
    const size_t slots = static_cast<size_t>(k) * nq;
    std::vector<float> final_distance(slots, worst);
    std::vector<int64_t> final_id(slots, -1);
    const size_t total_rows = column.rows;
    auto row_begin = [&](size_t r) { return r == 0 ? 0 : column.offsets[r - 1]; };
    // cosine: the block is stored normalised (once, at upload); the queries are normalised per call like searchWithoutIndex does
    std::vector<float> q = queries;
    VIMetric m = metric;
    if (metric == VIMetric::Cosine)
    {
        m = VIMetric::IP;
        VectorIndex::VectorDataset qd{q.data(), static_cast<int64_t>(nq), static_cast<int64_t>(dim)};
        qd.normalize();
    }
    std::vector<float> block;
    // normalised (Cosine) and raw (L2 / IP) copies of a part are different blocks; msvs_cache_evict(part_key) drops both
    const std::string form_key = part_key + (metric == VIMetric::Cosine ? "/cos" : "/raw");
    for (size_t mark_start = 0, mark = 0; mark_start < total_rows; mark_start += index_granularity, mark++)
    {
        const size_t mark_end = std::min(total_rows, mark_start + index_granularity), block_rows = mark_end - mark_start;
        msvs_block_t * blk = nullptr;
        VectorIndex::throwIfError(msvs_block_lookup(cache, form_key.c_str(), mark, &blk));
        if (blk)
        {
            // a hit must be the block this search would have uploaded: same rows, same dimension, same stored form (a per-query
            // metric setting may search one part under Cosine and under L2: the two forms live under different keys, and a block
            // whose shape no longer matches the mark means the part changed without an eviction)
            size_t bn = 0, bd = 0;
            int bnorm = 0;
            const int irc = msvs_block_info(blk, &bn, &bd, &bnorm);
            if (irc != 0 || bn != block_rows || bd != dim || bnorm != (metric == VIMetric::Cosine ? 1 : 0))
            {
                msvs_block_release(blk);
                VectorIndex::throwIfError(irc);
                throw VectorIndex::VIException(MSVS_ERR_INVALID_ARGUMENT, "resident block of `" + part_key + "` mark " + std::to_string(mark)
                                                                          + " does not match the part (rows / dimension / stored form): evict the part first");
            }
        }
        if (!blk)
        {
            // the dense block of the mark exactly as the scan builds it: rows without a vector padded with FLT_MAX
            block.assign(block_rows * dim, std::numeric_limits<float>::max());
            for (size_t r = mark_start; r < mark_end; r++)
            {
                const uint64_t b = row_begin(r), e = column.offsets[r];
                if (e - b == dim)
                    std::copy(column.data + b, column.data + e, block.begin() + (r - mark_start) * dim);
            }
            VectorIndex::throwIfError(msvs_block_upload(cache, form_key.c_str(), mark, block.data(), block_rows, dim,
                                                        metric == VIMetric::Cosine ? 1 : 0, &blk));
        }
        // rows the search may return: not lightweight-deleted; with a filter only the passing rows that carry a vector
        // (the filter path of the reference compacts exactly those rows, :1042-1330)
        std::vector<uint64_t> alive;
        if (filter || row_exists)
        {
            alive.assign((block_rows + 63) / 64, 0);
            for (size_t r = mark_start; r < mark_end; r++)
            {
                bool ok = !row_exists || row_exists->is_member(r);
                if (ok && filter)
                    ok = filter->is_member(r) && column.offsets[r] != row_begin(r);
                if (ok)
                    alive[(r - mark_start) >> 6] |= 1ull << ((r - mark_start) & 63);
            }
        }
        std::vector<float> per_distance(slots, worst);
        std::vector<int64_t> per_id(slots, -1);
        const int rc = msvs_knn_resident(blk, q.data(), static_cast<size_t>(k), nq, static_cast<int>(m),
                                         alive.empty() ? nullptr : alive.data(), per_id.data(), per_distance.data());
        msvs_block_release(blk);
        VectorIndex::throwIfError(rc);
        for (size_t i = 0; i < slots; i++)
        {
            if (metric == VIMetric::Cosine)
                per_distance[i] = 1 - per_distance[i];
            if (per_id[i] < 0)
                per_distance[i] = worst;
        }
        mergeBlockResult(per_id, per_distance, k, static_cast<int>(nq), static_cast<int>(mark_start), final_id, final_distance,
                         metric);
    }
    VectorScanResult res;
    for (size_t slot = 0; slot < slots; slot++)
    {
        if (final_id[slot] <= -1)
            continue;
        res.labels.push_back(static_cast<uint32_t>(final_id[slot]));
        if (is_batch)
            res.query_ids.push_back(static_cast<uint32_t>(slot / static_cast<size_t>(k)));
        res.distances.push_back(final_distance[slot]);
    }
    res.computed = true;
    return res;
}

std::vector<int64_t> mergeSearchResult(const std::vector<uint64_t> & part_offsets, const std::vector<uint32_t> & labels)
{
    // labels sorted together with their original positions, then one binary search per read row
    std::vector<std::pair<uint32_t, int64_t>> sorted(labels.size());
    for (size_t i = 0; i < labels.size(); i++)
        sorted[i] = {labels[i], static_cast<int64_t>(i)};
    std::sort(sorted.begin(), sorted.end());
    std::vector<int64_t> pos(part_offsets.size(), -1);
    for (size_t r = 0; r < part_offsets.size(); r++)
    {
        auto it = std::lower_bound(sorted.begin(), sorted.end(), std::make_pair(static_cast<uint32_t>(part_offsets[r]), int64_t(-1)));
        if (it != sorted.end() && it->first == part_offsets[r])
            pos[r] = it->second;
    }
    return pos;
}

std::vector<uint64_t> intersectDenseBitmaps(const std::vector<uint64_t> & a, const std::vector<uint64_t> & b)
{
    std::vector<uint64_t> out(std::min(a.size(), b.size()));
    for (size_t i = 0; i < out.size(); i++)
        out[i] = a[i] & b[i];
    return out;
}

ScoreWithPartIndexAndLabels MergeTreeBaseSearchManager::getTotalTopSearchResultImpl(
    const ScoreWithPartIndexAndLabels & all, uint64_t top_k, bool desc_direction)
{
    // a multimap keeps equal scores in insertion order; reverse iteration (descending) reverses that order too
    std::multimap<float, ScoreWithPartIndexAndLabel> sorted;
    for (const auto & e : all)
        sorted.emplace(e.score, e);
    ScoreWithPartIndexAndLabels result;
    result.reserve(top_k);
    if (desc_direction)
    {
        for (auto it = sorted.rbegin(); it != sorted.rend() && result.size() < top_k; ++it)
            result.push_back(it->second);
    }
    else
    {
        for (auto it = sorted.begin(); it != sorted.end() && result.size() < top_k; ++it)
            result.push_back(it->second);
    }
    return result;
}

void RankFusion(std::map<std::tuple<uint32_t, uint64_t, uint64_t>, float> & fusion_id_with_score,
                const ScoreWithPartIndexAndLabels & vec_scan_result_dataset,
                const ScoreWithPartIndexAndLabels & text_search_result_dataset, uint64_t fusion_k)
{
    // RRF: score += 1 / (fusion_k + rank), rank 1-based inside each list
    for (const auto * list : {&vec_scan_result_dataset, &text_search_result_dataset})
    {
        size_t rank = 1;
        for (const auto & e : *list)
        {
            fusion_id_with_score[std::make_tuple(e.shard_num, e.part_index, e.label_id)] += 1.0f / (fusion_k + rank);
            rank++;
        }
    }
}

void computeNormalizedScore(const ScoreWithPartIndexAndLabels & search_result_dataset, std::vector<float> & norm_score)
{
    const size_t n = search_result_dataset.size();
    if (n == 0)
        return;
    norm_score.reserve(n);
    float min_score = search_result_dataset[n - 1].score, max_score = search_result_dataset[0].score;
    if (min_score == max_score)
    {
        norm_score.assign(n, 1.0f);
        return;
    }
    if (min_score > max_score)
        std::swap(min_score, max_score);
    const float scale = max_score - min_score;
    for (const auto & e : search_result_dataset)
        norm_score.push_back((e.score - min_score) / scale);
}

void RelativeScoreFusion(std::map<std::tuple<uint32_t, uint64_t, uint64_t>, float> & fusion_id_with_score,
                         const ScoreWithPartIndexAndLabels & vec_scan_result_dataset,
                         const ScoreWithPartIndexAndLabels & text_search_result_dataset, float fusion_weight,
                         int8_t vector_scan_direction)
{
    std::vector<float> norm;
    computeNormalizedScore(text_search_result_dataset, norm);
    for (size_t i = 0; i < text_search_result_dataset.size(); i++)
    {
        const auto & e = text_search_result_dataset[i];
        fusion_id_with_score[std::make_tuple(e.shard_num, e.part_index, e.label_id)] = norm[i] * fusion_weight;
    }
    norm.clear();
    computeNormalizedScore(vec_scan_result_dataset, norm);
    for (size_t i = 0; i < vec_scan_result_dataset.size(); i++)
    {
        const auto & e = vec_scan_result_dataset[i];
        // direction -1 (IP): larger is better; otherwise a smaller distance is better
        const float s = vector_scan_direction == -1 ? norm[i] * (1 - fusion_weight) : (1 - norm[i]) * (1 - fusion_weight);
        fusion_id_with_score[std::make_tuple(e.shard_num, e.part_index, e.label_id)] += s;
    }
}

ScoreWithPartIndexAndLabels MergeTreeHybridSearchManager::hybridSearch(
    const ScoreWithPartIndexAndLabels & vec_scan_result_with_part_index,
    const ScoreWithPartIndexAndLabels & text_search_result_with_part_index, const HybridSearchInfo & hybrid_info)
{
    std::map<std::tuple<uint32_t, uint64_t, uint64_t>, float> fused;
    if (hybrid_info.fusion_type == "rsf" || hybrid_info.fusion_type == "RSF")
        RelativeScoreFusion(fused, vec_scan_result_with_part_index, text_search_result_with_part_index,
                            hybrid_info.fusion_weight, static_cast<int8_t>(hybrid_info.vector_scan_direction));
    else
        RankFusion(fused, vec_scan_result_with_part_index, text_search_result_with_part_index,
                   hybrid_info.fusion_k <= 0 ? 60 : static_cast<uint64_t>(hybrid_info.fusion_k));
    std::multimap<float, std::pair<uint64_t, uint64_t>, std::greater<float>> by_score;
    for (const auto & [id, score] : fused)
        by_score.emplace(score, std::make_pair(std::get<1>(id), std::get<2>(id)));
    ScoreWithPartIndexAndLabels result;
    for (const auto & [score, id] : by_score)
    {
        if (static_cast<int>(result.size()) == hybrid_info.topk)
            break;
        ScoreWithPartIndexAndLabel e;
        e.score = score;
        e.part_index = id.first;
        e.label_id = id.second;
        result.push_back(e);
    }
    return result;
}

std::vector<std::pair<size_t, float>> hybridSearchFusionTransform(const std::vector<FusionRow> & rows, uint64_t num_candidates,
                                                                  const HybridSearchInfo & info)
{
    const size_t total_rows = rows.size();
    // row ranges {start index, length}
    std::pair<size_t, size_t> distance_row_range{0, 0}, bm25_row_range{0, 0};
    for (size_t row = 0; row < total_rows; ++row)
    {
        if (rows[row].score_type == 0)
            bm25_row_range.first++;
        else
            break;
    }
    bm25_row_range.second = std::min<size_t>(total_rows - bm25_row_range.first, num_candidates);
    const int direction = info.vector_scan_direction;
    if (direction == -1)
    {
        distance_row_range.first = 0;
        distance_row_range.second = std::min<size_t>(bm25_row_range.first, num_candidates);
    }
    else if (direction == 1)
    {
        if (bm25_row_range.first >= num_candidates)
        {
            distance_row_range.first = bm25_row_range.first - num_candidates;
            distance_row_range.second = num_candidates;
        }
        else if (bm25_row_range.first > 0)
        {
            distance_row_range.first = 0;
            distance_row_range.second = bm25_row_range.first;
        }
    }
    auto entry = [&](size_t row) {
        ScoreWithPartIndexAndLabel e;
        e.score = rows[row].score;
        e.part_index = rows[row].part_index;
        e.label_id = rows[row].part_offset;
        e.shard_num = rows[row].shard_num;
        return e;
    };
    ScoreWithPartIndexAndLabels bm25_score_dataset, distance_score_dataset;
    for (size_t offset = 0; offset < distance_row_range.second; ++offset)
        distance_score_dataset.push_back(entry(direction == -1 ? distance_row_range.first + offset
                                                               : distance_row_range.first + distance_row_range.second - 1 - offset));
    for (size_t offset = 0; offset < bm25_row_range.second; ++offset)
        bm25_score_dataset.push_back(entry(bm25_row_range.first + offset));
    std::map<std::tuple<uint32_t, uint64_t, uint64_t>, float> fusion_id_with_score;
    if (info.fusion_type == "rsf")
        RelativeScoreFusion(fusion_id_with_score, distance_score_dataset, bm25_score_dataset, info.fusion_weight, (int8_t)direction);
    else
        RankFusion(fusion_id_with_score, distance_score_dataset, bm25_score_dataset, (uint64_t)info.fusion_k);
    std::vector<std::pair<size_t, float>> out;
    for (size_t i = 0; i < bm25_row_range.second; ++i) // the bm25 rows keep their place, their score becomes the fused one
    {
        const size_t row = bm25_row_range.first + i;
        const auto id = std::make_tuple(rows[row].shard_num, rows[row].part_index, rows[row].part_offset);
        out.emplace_back(row, fusion_id_with_score[id]);
        fusion_id_with_score.erase(id);
    }
    for (size_t offset = 0; offset < distance_row_range.second; ++offset) // then the distance rows that were not among them
    {
        const size_t row = distance_row_range.first + offset;
        const auto id = std::make_tuple(rows[row].shard_num, rows[row].part_index, rows[row].part_offset);
        auto it = fusion_id_with_score.find(id);
        if (it != fusion_id_with_score.end())
            out.emplace_back(row, it->second);
    }
    return out;
}


}

// ================================================================================================ C entry points

namespace
{
template <typename F>
int guarded(F && f)
{
    try
    {
        f();
        return MSVS_OK;
    }
    catch (const VectorIndex::VIException & e)
    {
        return e.code;
    }
    catch (...)
    {
        return MSVS_ERR_DEVICE;
    }
}

DB::ScoreWithPartIndexAndLabels make_list(const float * s, const uint64_t * p, const uint64_t * l, size_t n)
{
    DB::ScoreWithPartIndexAndLabels v(n);
    for (size_t i = 0; i < n; i++)
    {
        v[i].score = s[i];
        v[i].part_index = p[i];
        v[i].label_id = l[i];
    }
    return v;
}
}

extern "C" int msvs_host_search_without_index(float * query, float * base, size_t dim, size_t k, size_t nq, size_t nbase,
                                              int metric, int64_t * labels, float * distances)
{
    return guarded([&] {
        VectorIndex::VectorDataset q{query, static_cast<int64_t>(nq), static_cast<int64_t>(dim)};
        VectorIndex::VectorDataset b{base, static_cast<int64_t>(nbase), static_cast<int64_t>(dim)};
        VectorIndex::VIWithColumnInPart::searchWithoutIndex(q, b, static_cast<int32_t>(k), distances, labels,
                                                            static_cast<VectorIndex::VIMetric>(metric));
    });
}

extern "C" int msvs_host_search_wrapper(int prewhere, float * query, float * base, size_t nbase, int k, int dim, int nq,
                                        int num_rows_read, int64_t * final_id, float * final_distance,
                                        const uint64_t * actual_id_in_range, int metric, const uint64_t * row_exists,
                                        int delete_id_num)
{
    return guarded([&] {
        VectorIndex::VectorDataset q{query, nq, dim};
        VectorIndex::VectorDataset b{base, static_cast<int64_t>(nbase), dim};
        std::vector<int64_t> fid(final_id, final_id + static_cast<size_t>(k) * nq);
        std::vector<float> fdist(final_distance, final_distance + static_cast<size_t>(k) * nq);
        std::vector<size_t> actual;
        if (prewhere)
            actual.assign(actual_id_in_range, actual_id_in_range + nbase);
        DB::VIBitmapView bits{row_exists};
        DB::MergeTreeVSManager::searchWrapper(prewhere != 0, q, b, k, dim, nq, num_rows_read, fid, fdist, actual,
                                              static_cast<VectorIndex::VIMetric>(metric), bits, delete_id_num);
        std::memcpy(final_id, fid.data(), fid.size() * sizeof(int64_t));
        std::memcpy(final_distance, fdist.data(), fdist.size() * sizeof(float));
    });
}

extern "C" int msvs_host_vector_scan_without_index(const uint64_t * offsets, const float * data, size_t rows, size_t dim,
                                                   size_t index_granularity, const float * queries, size_t nq, int k,
                                                   int metric, int is_batch, const uint64_t * filter_bits,
                                                   const uint64_t * row_exists_bits, uint32_t * out_labels,
                                                   uint32_t * out_query_ids, float * out_distances, size_t * n_out)
{
    return guarded([&] {
        DB::ColumnArrayView col{offsets, data, rows};
        std::vector<float> q(queries, queries + nq * dim);
        DB::VIBitmapView f{filter_bits}, e{row_exists_bits};
        auto res = DB::MergeTreeVSManager::vectorScanWithoutIndex(col, dim, index_granularity, q, nq, k,
                                                                  static_cast<VectorIndex::VIMetric>(metric), is_batch != 0,
                                                                  filter_bits ? &f : nullptr, row_exists_bits ? &e : nullptr);
        for (size_t i = 0; i < res.labels.size(); i++)
        {
            out_labels[i] = res.labels[i];
            out_distances[i] = res.distances[i];
            if (is_batch)
                out_query_ids[i] = res.query_ids[i];
        }
        *n_out = res.labels.size();
    });
}

extern "C" int msvs_host_vector_scan_resident(msvs_cache_t * cache, const char * part_key, const uint64_t * offsets,
                                              const float * data, size_t rows, size_t dim, size_t index_granularity,
                                              const float * queries, size_t nq, int k, int metric, int is_batch,
                                              const uint64_t * filter_bits, const uint64_t * row_exists_bits,
                                              uint32_t * out_labels, uint32_t * out_query_ids, float * out_distances,
                                              size_t * n_out)
{
    return guarded([&] {
        DB::ColumnArrayView col{offsets, data, rows};
        std::vector<float> q(queries, queries + nq * dim);
        DB::VIBitmapView f{filter_bits}, e{row_exists_bits};
        auto res = DB::MergeTreeVSManager::vectorScanWithoutIndexResident(
            cache, part_key ? part_key : "", col, dim, index_granularity, q, nq, k, static_cast<VectorIndex::VIMetric>(metric),
            is_batch != 0, filter_bits ? &f : nullptr, row_exists_bits ? &e : nullptr);
        for (size_t i = 0; i < res.labels.size(); i++)
        {
            out_labels[i] = res.labels[i];
            out_distances[i] = res.distances[i];
            if (is_batch)
                out_query_ids[i] = res.query_ids[i];
        }
        *n_out = res.labels.size();
    });
}

extern "C" void msvs_host_merge_search_result(const uint64_t * part_offsets, size_t n_rows, const uint32_t * labels,
                                              size_t n_labels, int64_t * out_pos)
{
    auto pos = DB::mergeSearchResult(std::vector<uint64_t>(part_offsets, part_offsets + n_rows),
                                     std::vector<uint32_t>(labels, labels + n_labels));
    std::copy(pos.begin(), pos.end(), out_pos);
}

extern "C" size_t msvs_host_total_topk(const float * scores, const uint64_t * part_index, const uint64_t * labels,
                                       size_t n, size_t top_k, int desc_direction, float * out_scores,
                                       uint64_t * out_part_index, uint64_t * out_labels)
{
    auto r = DB::MergeTreeBaseSearchManager::getTotalTopSearchResultImpl(make_list(scores, part_index, labels, n), top_k,
                                                                         desc_direction != 0);
    for (size_t i = 0; i < r.size(); i++)
    {
        out_scores[i] = r[i].score;
        out_part_index[i] = r[i].part_index;
        out_labels[i] = r[i].label_id;
    }
    return r.size();
}

extern "C" size_t msvs_host_hybrid_search(int fusion_type, const float * vec_scores, const uint64_t * vec_parts,
                                          const uint64_t * vec_labels, size_t nvec, const float * txt_scores,
                                          const uint64_t * txt_parts, const uint64_t * txt_labels, size_t ntxt,
                                          uint64_t fusion_k, float fusion_weight, int vector_scan_direction, size_t topk,
                                          float * out_scores, uint64_t * out_parts, uint64_t * out_labels)
{
    DB::HybridSearchInfo info;
    info.fusion_type = fusion_type == 1 ? "rsf" : "rrf";
    info.fusion_k = static_cast<int>(fusion_k);
    info.fusion_weight = fusion_weight;
    info.topk = static_cast<int>(topk);
    info.vector_scan_direction = vector_scan_direction;
    auto r = DB::MergeTreeHybridSearchManager::hybridSearch(make_list(vec_scores, vec_parts, vec_labels, nvec),
                                                            make_list(txt_scores, txt_parts, txt_labels, ntxt), info);
    for (size_t i = 0; i < r.size(); i++)
    {
        out_scores[i] = r[i].score;
        out_parts[i] = r[i].part_index;
        out_labels[i] = r[i].label_id;
    }
    return r.size();
}

/* The same fusion for a BATCH of queries straight from the device searches' output arrays: query q's vector rows are
 * vec_dis / vec_ids[q * kv ...] (id < 0 ends the list, like the host's `> -1` unpack), its text rows txt_scores / txt_ids[q * kt ...];
 * one part (part index 0).  out_*[q * topk ...], n_out[q] rows each. */
extern "C" int msvs_host_hybrid_search_batch(int fusion_type, const float * vec_dis, const int64_t * vec_ids, size_t kv,
                                             const float * txt_scores, const int64_t * txt_ids, size_t kt, size_t nq, uint64_t fusion_k,
                                             float fusion_weight, int vector_scan_direction, size_t topk, float * out_scores,
                                             uint64_t * out_labels, uint32_t * n_out)
{
    if ((nq && (!vec_dis || !vec_ids || !txt_scores || !txt_ids || !out_scores || !out_labels || !n_out)) || topk == 0)
        return MSVS_ERR_INVALID_ARGUMENT;
    // The fusion of hybridSearch() over flat arrays instead of a std::map + std::multimap per query (~400 node allocations:
    // 31 us per query, the largest item of a hybrid batch).  Same arithmetic in the same order: a label's contributions are
    // applied in list order (RRF: vector list then text list, every one added to 0; RSF: the text list ASSIGNS weight * norm,
    // the vector list adds), and the output is the stable sort by descending score of the label-ascending sequence -- what
    // iterating the map into the multimap yields.  tests/test_host_mirror.py holds it against hybrid_search().
    struct Contribution
    {
        uint64_t label;
        uint32_t seq;
        float value;
        bool assign;
    };
    struct Fused
    {
        uint64_t label;
        float score;
    };
    const bool rsf = fusion_type == 1;
    const uint64_t fk = fusion_k == 0 ? 60 : fusion_k;
    std::vector<Contribution> c;
    std::vector<Fused> f;
    std::vector<float> norm;
    auto normalized = [&](const float * sc, size_t n) { // computeNormalizedScore
        norm.clear();
        if (n == 0)
            return;
        float mn = sc[n - 1], mx = sc[0];
        if (mn == mx)
        {
            norm.assign(n, 1.0f);
            return;
        }
        if (mn > mx)
            std::swap(mn, mx);
        const float scale = mx - mn;
        for (size_t i = 0; i < n; i++)
            norm.push_back((sc[i] - mn) / scale);
    };
    for (size_t q = 0; q < nq; q++)
    {
        size_t nv = 0, nt = 0;
        while (nv < kv && vec_ids[q * kv + nv] > -1)
            nv++;
        while (nt < kt && txt_ids[q * kt + nt] > -1)
            nt++;
        const float * vs = vec_dis + q * kv, * ts = txt_scores + q * kt;
        const int64_t * vi = vec_ids + q * kv, * ti = txt_ids + q * kt;
        c.clear();
        uint32_t seq = 0;
        if (rsf)
        {
            normalized(ts, nt);
            for (size_t i = 0; i < nt; i++)
                c.push_back(Contribution{(uint64_t)ti[i], seq++, norm[i] * fusion_weight, true});
            normalized(vs, nv);
            for (size_t i = 0; i < nv; i++)
                c.push_back(Contribution{(uint64_t)vi[i], seq++,
                                         vector_scan_direction == -1 ? norm[i] * (1 - fusion_weight) : (1 - norm[i]) * (1 - fusion_weight),
                                         false});
        }
        else
        {
            for (size_t i = 0; i < nv; i++)
                c.push_back(Contribution{(uint64_t)vi[i], seq++, 1.0f / (fk + (i + 1)), false});
            for (size_t i = 0; i < nt; i++)
                c.push_back(Contribution{(uint64_t)ti[i], seq++, 1.0f / (fk + (i + 1)), false});
        }
        std::sort(c.begin(), c.end(), [](const Contribution & a, const Contribution & b) {
            return a.label != b.label ? a.label < b.label : a.seq < b.seq;
        });
        f.clear();
        for (size_t i = 0; i < c.size();)
        {
            float sc = 0.0f;
            size_t j = i;
            for (; j < c.size() && c[j].label == c[i].label; j++)
                sc = c[j].assign ? c[j].value : sc + c[j].value;
            f.push_back(Fused{c[i].label, sc});
            i = j;
        }
        std::stable_sort(f.begin(), f.end(), [](const Fused & a, const Fused & b) { return a.score > b.score; });
        const size_t n = std::min(f.size(), topk);
        for (size_t i = 0; i < n; i++)
        {
            out_scores[q * topk + i] = f[i].score;
            out_labels[q * topk + i] = f[i].label;
        }
        n_out[q] = (uint32_t)n;
    }
    return MSVS_OK;
}

/* HybridSearchFusionTransform::generate over flat arrays; returns the number of output rows. */
extern "C" size_t msvs_host_fusion_transform(int fusion_type, const float * score, const uint8_t * score_type, const uint32_t * shard_num,
                                             const uint64_t * part_index, const uint64_t * part_offset, size_t n_rows,
                                             uint64_t num_candidates, uint64_t fusion_k, float fusion_weight, int vector_scan_direction,
                                             uint64_t * out_rows, float * out_scores)
{
    std::vector<DB::FusionRow> rows(n_rows);
    for (size_t i = 0; i < n_rows; i++)
    {
        rows[i].score = score[i];
        rows[i].score_type = score_type[i];
        rows[i].shard_num = shard_num[i];
        rows[i].part_index = part_index[i];
        rows[i].part_offset = part_offset[i];
    }
    DB::HybridSearchInfo info;
    info.fusion_type = fusion_type == 1 ? "rsf" : "rrf";
    info.fusion_k = static_cast<int>(fusion_k);
    info.fusion_weight = fusion_weight;
    info.vector_scan_direction = vector_scan_direction;
    const auto r = DB::hybridSearchFusionTransform(rows, num_candidates, info);
    for (size_t i = 0; i < r.size(); i++)
    {
        out_rows[i] = r[i].first;
        out_scores[i] = r[i].second;
    }
    return r.size();
}

extern "C" int msvs_host_merge_topk(const int64_t * ids, const float * dis, size_t nparts, size_t nq, size_t k, int metric,
                                    int64_t * out_ids, float * out_dis)
{
    if (metric != MSVS_METRIC_L2 && metric != MSVS_METRIC_IP)
        return MSVS_ERR_NOT_IMPLEMENTED;
    const bool desc = metric == MSVS_METRIC_IP;
    std::vector<std::pair<float, int64_t>> all;
    for (size_t q = 0; q < nq; q++)
    {
        all.clear();
        for (size_t p = 0; p < nparts; p++)
            for (size_t j = 0; j < k; j++)
            {
                const size_t o = (p * nq + q) * k + j;
                if (ids[o] >= 0)
                    all.emplace_back(dis[o], ids[o]);
            }
        std::sort(all.begin(), all.end(), [desc](const auto & a, const auto & b) {
            if (a.first != b.first)
                return desc ? a.first > b.first : a.first < b.first;
            return a.second < b.second;
        });
        for (size_t j = 0; j < k; j++)
        {
            out_ids[q * k + j] = j < all.size() ? all[j].second : -1;
            out_dis[q * k + j] = j < all.size() ? all[j].first
                                                : (desc ? -std::numeric_limits<float>::max() : std::numeric_limits<float>::max());
        }
    }
    return MSVS_OK;
}

extern "C" int msvs_host_generate_vector_dataset(const void * values, int is_float64, const uint64_t * offsets, size_t nq,
                                                 size_t dim, float * out)
{
    return guarded([&] {
        const std::vector<float> v = DB::MergeTreeVSManager::generateVectorDataset(values, is_float64 != 0, offsets, nq, dim);
        std::copy(v.begin(), v.end(), out);
    });
}

extern "C" void msvs_host_sum_bm25_stats(const uint64_t * per_part, size_t nparts, size_t n_terms, uint64_t * out)
{
    const size_t w = 2 + n_terms;
    for (size_t j = 0; j < w; j++)
        out[j] = 0;
    for (size_t p = 0; p < nparts; p++)
        for (size_t j = 0; j < w; j++)
            out[j] += per_part[p * w + j];
}

extern "C" int msvs_host_all_reduce_bm25_stats(const struct msvs_comm * comm, uint64_t * stats, size_t n_terms, void * hip_stream)
{
    if (!comm || !stats)
        return MSVS_ERR_INVALID_ARGUMENT;
    return msvs_comm_all_reduce_u64(comm, stats, 2 + n_terms, hip_stream);
}

/* Measurement / test driver for the reference's calling pattern (MergeTreeVSManager.cpp:973: up to ScanThreadLimiter-many host
 * threads, ONE query per VectorIndex::search call): `threads` native threads, thread t searching queries t, t + threads, ...
 * (wrapping) `calls_per_thread` times through msvs_index_search.  seconds = wall time of the whole run; lat_us (nullable,
 * [threads * calls_per_thread]) every call's latency; ids / dis (nullable, [n_queries][k]) the rows of each query's last call. */
#include <chrono>
#include <thread>
extern "C" int msvs_host_concurrent_search(const msvs_index_t * ix, const float * queries, size_t n_queries, size_t dim, int threads,
                                           size_t calls_per_thread, int k, const char * params, double * seconds, float * lat_us,
                                           int64_t * ids, float * dis)
{
    if (!ix || !queries || n_queries == 0 || threads < 1 || k < 1 || !seconds)
        return MSVS_ERR_INVALID_ARGUMENT;
    std::vector<int> rc((size_t)threads, 0);
    std::vector<std::thread> pool;
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < threads; t++)
        pool.emplace_back([&, t] {
            std::vector<int64_t> li((size_t)k);
            std::vector<float> ld((size_t)k);
            for (size_t c = 0; c < calls_per_thread; c++)
            {
                const size_t qi = ((size_t)t + c * (size_t)threads) % n_queries;
                const auto a = std::chrono::steady_clock::now();
                const int r = msvs_index_search(ix, queries + qi * dim, 1, k, params, nullptr, 0, ids ? ids + qi * (size_t)k : li.data(),
                                                dis ? dis + qi * (size_t)k : ld.data());
                const auto b = std::chrono::steady_clock::now();
                if (lat_us)
                    lat_us[(size_t)t * calls_per_thread + c] = std::chrono::duration<float, std::micro>(b - a).count();
                if (r)
                {
                    rc[(size_t)t] = r;
                    return;
                }
            }
        });
    for (auto & th : pool)
        th.join();
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int r : rc)
        if (r)
            return r;
    return MSVS_OK;
}

