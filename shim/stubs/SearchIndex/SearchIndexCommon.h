// STUB of the absent contrib/search-index headers, reconstructed from the reference's CALL SITES only (SURVEY.md
// Appendix A lists the evidence, file:line in /root/reference): just enough of namespace Search for the host code of
// the hot path -- and shim/HostShim.cpp -- to compile against.  Nothing here is copied: the real headers are not in
// the tree (.gitmodules:338-340, empty submodule).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace Search
{

using idx_t = int64_t; // VIPartReader.h:124,199; result ids int64 with -1 = empty (MergeTreeVSManager.h:212)

enum class DataType // VSDescription.h:28-31, VICommon.h:155,163
{
    FloatVector,
    BinaryVector
};

enum class Metric // MergeTreeVSManager.cpp:1560-1583, VIWithDataPart.h:193
{
    L2,
    IP,
    Cosine,
    Hamming,
    Jaccard
};

enum class IndexType // VICommon.h:173-184, type names of tests/queries/2_vector_search
{
    FLAT,
    BinaryFLAT,
    IVFFLAT,
    IVFPQ,
    IVFSQ,
    HNSWFLAT,
    HNSWSQ,
    HNSWPQ,
    SCANN,
    MSTG,
    BinaryMSTG
};

std::string enumToString(IndexType t);
std::string enumToString(Metric m);
/// case-insensitive (helpers/00000_prepare_index_cosine.sh:6 writes "cosine")
Metric getMetricType(const std::string & name, DataType type);
IndexType getVectorIndexType(const std::string & name, DataType type);
/// VIDescriptions.cpp:41,133 (`TYPE DEFAULT`), declared there with exactly this signature
std::string getDefaultIndexType(const DataType & search_type);
/// parseVSParameters.cpp:78, VIDescriptions.cpp:172: a JSON object  INDEX TYPE (upper case) -> parameter name ->
/// {"type": "int" | "float" | "string", "case_sensitive": bool, "range": [lo, hi] | [], "candidates": [...]}  driving the DDL
/// check of `TYPE X('k=v', ...)` (VIDescriptions.cpp:248-330) and the check of search arguments (parseVSParameters.cpp:76-222)
extern const std::string MYSCALE_VALID_INDEX_PARAMETER;

/// string -> string map (VICommon.h:127,186-212; MergeTreeVSManager.cpp:361-366; VIWithDataPart.cpp:645,910-911)
class Parameters : public std::map<std::string, std::string>
{
public:
    void setParam(const std::string & key, const std::string & value) { (*this)[key] = value; }
    template <typename T>
    void setParam(const std::string & key, const T & value)
    {
        (*this)[key] = std::to_string(value);
    }
    std::string toString() const
    {
        std::string s = "{";
        for (const auto & kv : *this)
            s += (s.size() > 1 ? "," : "") + ("\"" + kv.first + "\":\"" + kv.second + "\"");
        return s + "}";
    }
};

class SearchIndexException : public std::exception // VICommon.h:86-91, VIWithDataPart.cpp:948-952
{
public:
    SearchIndexException(int code_, std::string msg_) : code(code_), msg(std::move(msg_)) {}
    int getCode() const { return code; }
    const char * what() const noexcept override { return msg.c_str(); }

private:
    int code;
    std::string msg;
};

/// non-owning view (VIWithDataPart.cpp:851-852,922-924)
template <typename T>
class DataSet
{
public:
    DataSet(T * data_, int64_t n_, int64_t dim_) : data(data_), n(n_), dim(dim_) {}
    T * getData() const { return data; }
    int64_t numData() const { return n; }
    int64_t dimension() const { return dim; }

private:
    T * data;
    int64_t n, dim;
};

/// flat nq x k result (MergeTreeVSManager.cpp:456-461,565-567,604; VIWithDataPart.cpp:61-66,90-95,114-117)
class SearchResult
{
public:
    static std::shared_ptr<SearchResult> createTopKHolder(int64_t nq, int64_t k)
    {
        return std::shared_ptr<SearchResult>(new SearchResult(nq, k));
    }
    idx_t * getResultIndices() { return ids.data(); }
    float * getResultDistances() { return dis.data(); }
    /// the k slots of query q (iterable, mutable)
    struct Slice
    {
        idx_t * b;
        idx_t * e;
        idx_t * begin() const { return b; }
        idx_t * end() const { return e; }
    };
    Slice getResultIndices(int64_t q) { return Slice{ids.data() + q * k, ids.data() + (q + 1) * k}; }
    int64_t getNumCandidates() const { return k; }
    int64_t numQueries() const { return nq; }

private:
    SearchResult(int64_t nq_, int64_t k_) : nq(nq_), k(k_), ids((size_t)(nq_ * k_), -1), dis((size_t)(nq_ * k_), 0.f) {}
    int64_t nq, k;
    std::vector<idx_t> ids;
    std::vector<float> dis;
};

/// 1 = the row passes (WHERE and not deleted); (.*Processor.cpp:917-927, MergeTreeVSManager.cpp:1058-1062,1436-1456,
/// VIWithDataPart.cpp:905-908, MergeTreeTextSearchManager.cpp:191-194,224-232).  Bit order: LSB first inside each byte
/// (the reference hands get_bitmap() verbatim to Tantivy; the order itself is "parity unpinned").
class DenseBitmap
{
public:
    explicit DenseBitmap(size_t n_, bool fill = false) : n(n_), bytes((n_ + 7) / 8, fill ? 0xFF : 0x00)
    {
        if (fill && (n & 7))
            bytes.back() = (uint8_t)((1u << (n & 7)) - 1);
    }
    void set(size_t i) { bytes[i >> 3] |= (uint8_t)(1u << (i & 7)); }
    void unset(size_t i) { bytes[i >> 3] &= (uint8_t)~(1u << (i & 7)); }
    bool is_member(size_t i) const { return i < n && unsafe_test(i); }
    bool unsafe_test(size_t i) const { return (bytes[i >> 3] >> (i & 7)) & 1; }
    size_t get_size() const { return n; }
    size_t byte_size() const { return bytes.size(); }
    size_t count() const
    {
        size_t c = 0;
        for (uint8_t b : bytes)
            c += (size_t)__builtin_popcount(b);
        return c;
    }
    bool any() const { return count() != 0; }
    bool all() const { return count() == n; }
    std::vector<size_t> to_vector() const
    {
        std::vector<size_t> v;
        for (size_t i = 0; i < n; i++)
            if (unsafe_test(i))
                v.push_back(i);
        return v;
    }
    uint8_t * get_bitmap() { return bytes.data(); }
    const uint8_t * get_bitmap() const { return bytes.data(); }

private:
    size_t n;
    std::vector<uint8_t> bytes;
};

inline std::shared_ptr<DenseBitmap> intersectDenseBitmaps(const std::shared_ptr<DenseBitmap> & a,
                                                          const std::shared_ptr<DenseBitmap> & b)
{
    if (!a)
        return b;
    if (!b)
        return a;
    auto r = std::make_shared<DenseBitmap>(std::min(a->get_size(), b->get_size()));
    for (size_t i = 0; i < r->byte_size(); i++)
        r->get_bitmap()[i] = a->get_bitmap()[i] & b->get_bitmap()[i];
    return r;
}

struct IndexResourceUsage // VIWithDataPart.cpp:486-488, VIWithDataPart.h:333
{
    size_t memory_usage_bytes = 0;
    size_t disk_usage_bytes = 0;
    size_t build_memory_usage_bytes = 0;
};

struct IndexVersion // getVersion().toString(), VIWithDataPart.cpp:485
{
    std::string v;
    std::string toString() const { return v; }
};

}
using Search::SearchIndexException; // VICommon.h:86 uses the unqualified name
