// STUB (see SearchIndexCommon.h): Search::VectorIndex<IS, OS, Bitmap, DataType> as the host drives it --
// VIWithDataPart.h:332-337 (setTrainDataChunkSize / setAddDataChunkSize / getResourceUsage / build),
// VIWithDataPart.cpp:415-446 (createVectorIndex), :472-479 (serialize / saveDataID / getVersion), :698-700 (load /
// loadDataID / numData), :853 (computeTopDistanceSubset), :878 (ready), :926 (search), :131 (supportTwoStageSearch).
#pragma once

#include <functional>

#include "Common/IndexDataIO.h"
#include "SearchIndexCommon.h"

namespace Search
{

template <DataType T>
struct DataTypeTraits;
template <>
struct DataTypeTraits<DataType::FloatVector>
{
    using Element = float;
};
template <>
struct DataTypeTraits<DataType::BinaryVector>
{
    using Element = bool; // one byte carries 8 bits (VIPartReader.h:148-150)
};

template <typename IS, typename OS, typename Bitmap, DataType T>
class VectorIndex
{
public:
    using Element = typename DataTypeTraits<T>::Element;
    virtual ~VectorIndex() = default;

    virtual void setTrainDataChunkSize(size_t bytes) = 0;
    virtual void setAddDataChunkSize(size_t bytes) = 0;
    virtual void build(IndexSourceDataReader<Element> * reader, int num_threads, std::function<bool()> check_cancelled) = 0;

    virtual std::shared_ptr<SearchResult> search(std::shared_ptr<DataSet<Element>> queries, int32_t k, Parameters & params,
                                                 bool first_stage_only, Bitmap * filter) = 0;
    virtual std::shared_ptr<SearchResult> computeTopDistanceSubset(std::shared_ptr<DataSet<Element>> queries,
                                                                   std::shared_ptr<SearchResult> first_stage, int32_t top_k)
        = 0;
    virtual bool supportTwoStageSearch() const = 0;

    virtual void serialize(IndexDataFileWriter<OS> * writer) = 0;
    virtual void saveDataID(IndexDataFileWriter<OS> * writer) = 0;
    virtual void load(IndexDataFileReader<IS> * reader, std::function<bool()> check_expired) = 0;
    virtual void loadDataID(IndexDataFileReader<IS> * reader) = 0;

    virtual bool ready() const = 0;
    virtual size_t numData() const = 0;
    virtual IndexResourceUsage getResourceUsage() const = 0;
    virtual IndexVersion getVersion() const = 0;
};

/// VIWithDataPart.cpp:415-430: (name, type, metric, dimension, total_vec, params, max_threads, cache_prefix, cancel callback)
template <typename IS, typename OS, typename Bitmap, DataType T>
std::shared_ptr<VectorIndex<IS, OS, Bitmap, T>> createVectorIndex(const std::string & name, IndexType type, Metric metric,
                                                                  size_t dimension, size_t total_vec, const Parameters & params,
                                                                  size_t max_threads, const std::string & cache_prefix,
                                                                  std::function<bool()> check_cancelled);

}
