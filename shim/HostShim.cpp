// HostShim.cpp -- what a MyScaleDB maintainer links INSTEAD of contrib/search-index for the hot path: the library's
// C++ surface (namespace Search, the four faiss:: brute-force calls) implemented on the C-ABI of libmsvs.so.
// Compiled against shim/stubs/ (reconstructions of the absent headers from the reference's call sites, SURVEY.md
// Appendix A) so that the boundary code has been through a compiler; with the real headers only the include path
// changes.  Everything computes on the GPU through libmsvs; errors come back as Search::SearchIndexException, which
// the host already converts to VIException (VICommon.h:75-104, VIWithDataPart.cpp:948-956).
//
//   seam A1  Search::VectorIndex<IS, OS, Bitmap, FloatVector>   -> MsvsVectorIndex            (FLAT, IVFFLAT, MSTG*)
//            Search::VectorIndex<IS, OS, Bitmap, BinaryVector>  -> MsvsBinaryIndex            (BinaryFLAT, BinaryMSTG*)
//   seam A2  faiss::knn_L2sqr / knn_inner_product / hammings_knn_mc / jaccard_knn -> msvs_knn_f32 / msvs_knn_bin
//   (* MSTG is proprietary and absent: its partition scan is served by IVFFLAT with the same metric)
//
// FORWARDING (-DMSVS_SEARCH_FORWARD_SUFFIX=<suffix>): a build that keeps contrib/search-index for the index types libmsvs does
// not serve -- HNSWFLAT / HNSWSQ / HNSWPQ, IVFPQ, IVFSQ, SCANN: the types the open-source README recommends and the functional
// tests name ~90 times -- compiles that library with its four entry points renamed by the preprocessor
//     -DcreateVectorIndex=createVectorIndex<suffix> -DgetDefaultIndexType=getDefaultIndexType<suffix>
//     -DMYSCALE_VALID_INDEX_PARAMETER=MYSCALE_VALID_INDEX_PARAMETER<suffix>
//     -Dknn_L2sqr=knn_L2sqr<suffix> -Dknn_inner_product=knn_inner_product<suffix> -Dhammings_knn_mc=... -Djaccard_knn=...
// (a macro renames definition AND internal callers, so the library stays self-consistent; its CLASSES keep their names and are the
// very types of the stub headers, which is why the rename is by symbol and not by namespace) and links both.  This file then
// forwards createVectorIndex for every type it does not serve to createVectorIndex<suffix>, takes getDefaultIndexType and the
// parameter table from the original, and keeps serving FLAT / IVFFLAT / MSTG / Binary* itself.  Same device as TextShim.cpp's
// MSVS_TANTIVY_FORWARD_NS tee.  Without the macro this library stands alone: unserved types throw NOT_IMPLEMENTED, which the
// host turns into its CPU brute-force fallback (VIWithDataPart.cpp:948-956).
#include <SearchIndex/VectorIndex.h>
#include <faiss/utils/distances.h>

#include <cctype>
#include <cstdlib>

#include "../include/msvs.h"

#ifdef MSVS_SEARCH_FORWARD_SUFFIX
#define MSVS_CAT2(a, b) a##b
#define MSVS_CAT(a, b) MSVS_CAT2(a, b)
#define MSVS_FWD(name) MSVS_CAT(name, MSVS_SEARCH_FORWARD_SUFFIX)
namespace Search
{
// the original library's entry points under their build-time names (see the header of this file)
template <typename IS, typename OS, typename Bitmap, DataType T>
std::shared_ptr<VectorIndex<IS, OS, Bitmap, T>> MSVS_FWD(createVectorIndex)(const std::string & name, IndexType type, Metric metric,
                                                                            size_t dimension, size_t total_vec, const Parameters & params,
                                                                            size_t max_threads, const std::string & cache_prefix,
                                                                            std::function<bool()> check_cancelled);
std::string MSVS_FWD(getDefaultIndexType)(const DataType & search_type);
extern const std::string MSVS_FWD(MYSCALE_VALID_INDEX_PARAMETER);
}
#endif

namespace
{

/// The index types libmsvs serves itself (everything else: forwarded, or NOT_IMPLEMENTED).
[[maybe_unused]] bool served_by_msvs(Search::IndexType t)
{
    return t == Search::IndexType::FLAT || t == Search::IndexType::IVFFLAT || t == Search::IndexType::MSTG
        || t == Search::IndexType::BinaryFLAT || t == Search::IndexType::BinaryMSTG;
}

/// msvs_index_set_cancel's C callback over the factory's std::function (VIWithDataPart.cpp:425-430).
int cancel_trampoline(void * ctx)
{
    auto * f = static_cast<std::function<bool()> *>(ctx);
    return (*f && (*f)()) ? 1 : 0;
}

[[noreturn]] void raise(int code)
{
    throw Search::SearchIndexException(code, msvs_last_error());
}

inline void check(int rc)
{
    if (rc != MSVS_OK)
        raise(rc);
}

std::string lower(std::string s)
{
    for (auto & c : s)
        c = (char)tolower((unsigned char)c);
    return s;
}

int msvs_metric_of(Search::Metric m)
{
    switch (m)
    {
        case Search::Metric::L2:
            return MSVS_METRIC_L2;
        case Search::Metric::IP:
            return MSVS_METRIC_IP;
        case Search::Metric::Cosine:
            return MSVS_METRIC_COSINE;
        case Search::Metric::Hamming:
            return MSVS_METRIC_HAMMING;
        default:
            return MSVS_METRIC_JACCARD;
    }
}

/// msvs_io_t over the library's file-set objects: NAME -> writer->open(NAME) / reader->open(NAME), i.e. the host's
/// VectorIndexWriter / VectorIndexReader over IDisk (VectorIndexIO.h:25-166).
template <typename IS, typename OS>
struct StreamIO
{
    Search::IndexDataFileWriter<OS> * writer = nullptr;
    Search::IndexDataFileReader<IS> * reader = nullptr;
    std::vector<std::shared_ptr<OS>> outs;
    std::vector<std::shared_ptr<IS>> ins;

    static void * open(void * ctx, const char * name, int write)
    {
        auto * self = static_cast<StreamIO *>(ctx);
        if (write)
        {
            auto s = self->writer ? self->writer->open(name) : nullptr;
            if (!s)
                return nullptr;
            self->outs.push_back(s);
            return s.get();
        }
        auto s = self->reader ? self->reader->open(name) : nullptr;
        if (!s || !s->is_open())
            return nullptr;
        self->ins.push_back(s);
        return s.get();
    }
    static int64_t write(void *, void * stream, const void * buf, size_t n)
    {
        static_cast<OS *>(stream)->write(static_cast<const char *>(buf), (std::streamsize)n);
        return (int64_t)n; // VectorIndexWriter::write has no failure channel of its own; close() finalises
    }
    static int64_t read(void *, void * stream, void * buf, size_t n)
    {
        auto * s = static_cast<IS *>(stream);
        s->read(static_cast<char *>(buf), (std::streamsize)n);
        return (int64_t)s->gcount();
    }
    static int close(void *, void * stream)
    {
        // output streams: AbstractOStream::close(); input streams have nothing to close (the holder drops them)
        return stream ? 0 : 1;
    }
    msvs_io_t io() { return msvs_io_t{this, &open, &write, &read, &close}; }
};

template <typename IS, typename OS, typename Bitmap>
class MsvsVectorIndex final : public Search::VectorIndex<IS, OS, Bitmap, Search::DataType::FloatVector>
{
public:
    using Reader = Search::IndexSourceDataReader<float>;

    MsvsVectorIndex(Search::IndexType type_, Search::Metric metric_, size_t dim_, size_t total_vec_,
                    const Search::Parameters & params_, std::function<bool()> factory_cancel_ = {})
        : type(type_), metric(metric_), dim(dim_), total_vec(total_vec_), params(params_), factory_cancel(std::move(factory_cancel_))
    {
        create();
    }
    ~MsvsVectorIndex() override { msvs_index_free(ix); }

    void setTrainDataChunkSize(size_t bytes) override { train_chunk = bytes; }
    void setAddDataChunkSize(size_t bytes) override { add_chunk = bytes; }

    /// train on a sample, then add the part chunk by chunk (VIPartReader::readDataImpl feeds dense float[n x dim] +
    /// idx_t[n]; rows of empty arrays arrive as zero vectors with id 0 and are listed in the reader's emptyIds(): the
    /// host excludes them through the filter bitmap, as it does for every index type)
    /// check_cancelled (and the factory's callback, VIWithDataPart.cpp:425-430) are polled between the chunks here AND inside the
    /// library between k-means iterations / at every add / in build (msvs_index_set_cancel): a cancelled build leaves with the
    /// host's own error, code DB::ErrorCodes::ABORTED (236) "Cancelled building vector index" (VIPartReader.h:175-176).
    void build(Reader * reader, int /*num_threads*/, std::function<bool()> check_cancelled) override
    {
        std::function<bool()> either = [this, check_cancelled] {
            return (check_cancelled && check_cancelled()) || (factory_cancel && factory_cancel());
        };
        struct Uninstall
        {
            msvs_index_t * ix;
            ~Uninstall() { msvs_index_set_cancel(ix, nullptr, nullptr); }
        } uninstall{ix};
        check(msvs_index_set_cancel(ix, &cancel_trampoline, &either));
        const size_t row_bytes = dim * sizeof(float);
        if (msvs_kind() == MSVS_INDEX_IVFFLAT)
        {
            const size_t want = std::max<size_t>(ncentroids() * 64, 4096);
            auto sample = reader->sampleData(std::min(want, std::max<size_t>(total_vec, 1)));
            check(msvs_index_train(ix, sample->getData(), sample->numData(), MSVS_MEM_HOST));
        }
        const size_t rows_per = std::max<size_t>(1, (add_chunk ? add_chunk : ((size_t)64 << 20)) / row_bytes);
        while (!reader->eof())
        {
            if (either())
                throw Search::SearchIndexException(MSVS_ERR_ABORTED, "Cancelled building vector index");
            auto chunk = reader->readData(rows_per);
            if (!chunk)
                break;
            check(msvs_index_add(ix, chunk->getData(), chunk->getDataID(), chunk->numData(), MSVS_MEM_HOST));
        }
        check(msvs_index_build(ix));
    }

    std::shared_ptr<Search::SearchResult> search(std::shared_ptr<Search::DataSet<float>> queries, int32_t k,
                                                 Search::Parameters & search_params, bool /*first_stage_only*/,
                                                 Bitmap * filter) override
    {
        auto res = Search::SearchResult::createTopKHolder(queries->numData(), k);
        std::string p;
        for (const auto & kv : search_params)
            if (kv.first != "load_index_version" && kv.first != "metric_type")
                p += (p.empty() ? "" : ",") + kv.first + "=" + kv.second;
        // DenseBitmap bytes are LSB-first: a byte array IS the little-endian u64 word array libmsvs takes (padded copy
        // so that the last word is whole)
        std::vector<uint64_t> words;
        size_t nbits = 0;
        if (filter)
        {
            nbits = filter->get_size();
            words.assign((nbits + 63) / 64 + 1, 0);
            memcpy(words.data(), filter->get_bitmap(), filter->byte_size());
        }
        check(msvs_index_search(ix, queries->getData(), (size_t)queries->numData(), k, p.c_str(),
                                filter ? words.data() : nullptr, nbits, res->getResultIndices(), res->getResultDistances()));
        return res;
    }

    /// dormant in the reference (no caller sets first_stage_only: SURVEY.md Appendix C); every search here is exact
    std::shared_ptr<Search::SearchResult> computeTopDistanceSubset(std::shared_ptr<Search::DataSet<float>>,
                                                                   std::shared_ptr<Search::SearchResult> first_stage,
                                                                   int32_t) override
    {
        return first_stage;
    }
    bool supportTwoStageSearch() const override { return false; }

    void serialize(Search::IndexDataFileWriter<OS> * writer) override
    {
        StreamIO<IS, OS> s;
        s.writer = writer;
        const msvs_io_t io = s.io();
        check(msvs_index_serialize_io(ix, &io)); // writes data_bin AND id_list
        for (auto & o : s.outs)
            o->close();
    }
    void saveDataID(Search::IndexDataFileWriter<OS> *) override {} // id_list travels with serialize()
    /// check_expired (VIWithDataPart.cpp:577-600, :698: the cache entry was dropped while the load was queued) is asked before the
    /// files are read and again before the loaded index replaces the old one: an expired load leaves the object untouched
    void load(Search::IndexDataFileReader<IS> * reader, std::function<bool()> check_expired) override
    {
        if (check_expired && check_expired())
            throw Search::SearchIndexException(MSVS_ERR_ABORTED, "vector index expired before load");
        StreamIO<IS, OS> s;
        s.reader = reader;
        const msvs_io_t io = s.io();
        msvs_index_t * loaded = nullptr;
        check(msvs_index_load_io(&io, &loaded));
        if (check_expired && check_expired())
        {
            msvs_index_free(loaded);
            throw Search::SearchIndexException(MSVS_ERR_ABORTED, "vector index expired during load");
        }
        msvs_index_free(ix);
        ix = loaded;
    }
    void loadDataID(Search::IndexDataFileReader<IS> *) override {}

    bool ready() const override { return msvs_index_ready(ix) != 0; }
    size_t numData() const override { return msvs_index_num_data(ix); }
    Search::IndexResourceUsage getResourceUsage() const override
    {
        Search::IndexResourceUsage u;
        check(msvs_index_resource_usage(ix, &u.memory_usage_bytes, &u.disk_usage_bytes, &u.build_memory_usage_bytes));
        if (!ready()) // asked before build (VIWithDataPart.h:333): estimate from total_vec
            u.build_memory_usage_bytes = std::max(u.build_memory_usage_bytes, 3 * total_vec * dim * sizeof(float));
        return u;
    }
    Search::IndexVersion getVersion() const override { return Search::IndexVersion{msvs_index_version()}; }

private:
    int msvs_kind() const { return type == Search::IndexType::FLAT ? MSVS_INDEX_FLAT : MSVS_INDEX_IVFFLAT; }
    size_t ncentroids() const
    {
        auto it = params.find("ncentroids");
        if (it != params.end())
            return (size_t)std::max(1l, atol(it->second.c_str()));
        // MSTG / default: ~sqrt-ish partitions of ~1000 rows
        return std::max<size_t>(1, std::min<size_t>(65536, total_vec / 1000 + 1));
    }
    void create()
    {
        if (type != Search::IndexType::FLAT && type != Search::IndexType::IVFFLAT && type != Search::IndexType::MSTG)
            throw Search::SearchIndexException(MSVS_ERR_NOT_IMPLEMENTED,
                                               "index type " + Search::enumToString(type) + " is not served by libmsvs");
        std::string p = "ncentroids=" + std::to_string(ncentroids());
        check(msvs_index_create(msvs_kind(), msvs_metric_of(metric), dim, p.c_str(), &ix));
    }

    Search::IndexType type;
    Search::Metric metric;
    size_t dim, total_vec;
    Search::Parameters params;
    std::function<bool()> factory_cancel;
    size_t train_chunk = 0, add_chunk = 0;
    msvs_index_t * ix = nullptr;
};

/// Search::VectorIndex<IS, OS, Bitmap, BinaryVector> (VICommon.h:142-143 BinaryVI; created at VIWithDataPart.cpp:431-446,
/// searched at :928-935): BinaryFLAT and the scan stage of BinaryMSTG as an exhaustive Hamming / Jaccard scan over rows kept
/// resident on the device (msvs_bin_index_*).  Element = bool with one BYTE carrying 8 bits (VIPartReader.h:143-150): a dataset
/// of `dimension` bits is dimension / 8 bytes per row.
template <typename IS, typename OS, typename Bitmap>
class MsvsBinaryIndex final : public Search::VectorIndex<IS, OS, Bitmap, Search::DataType::BinaryVector>
{
public:
    using Reader = Search::IndexSourceDataReader<bool>;

    MsvsBinaryIndex(Search::IndexType type_, Search::Metric metric_, size_t dim_bits, size_t total_vec_,
                    std::function<bool()> factory_cancel_ = {})
        : type(type_), metric(metric_), dim(dim_bits), total_vec(total_vec_), factory_cancel(std::move(factory_cancel_))
    {
        if (type != Search::IndexType::BinaryFLAT && type != Search::IndexType::BinaryMSTG)
            throw Search::SearchIndexException(MSVS_ERR_NOT_IMPLEMENTED,
                                               "index type " + Search::enumToString(type) + " is not a binary index served by libmsvs");
        if (dim == 0 || dim % 8 != 0)
            throw Search::SearchIndexException(MSVS_ERR_INVALID_ARGUMENT, "BinaryVector dimension must be a multiple of 8");
        check(msvs_bin_index_create(dim / 8, msvs_metric_of(metric), &ix));
    }
    ~MsvsBinaryIndex() override { msvs_bin_index_free(ix); }

    void setTrainDataChunkSize(size_t) override {}
    void setAddDataChunkSize(size_t bytes) override { add_chunk = bytes; }

    void build(Reader * reader, int /*num_threads*/, std::function<bool()> check_cancelled) override
    {
        const size_t row_bytes = dim / 8;
        const size_t rows_per = std::max<size_t>(1, (add_chunk ? add_chunk : ((size_t)64 << 20)) / row_bytes);
        while (!reader->eof())
        {
            if ((check_cancelled && check_cancelled()) || (factory_cancel && factory_cancel()))
                throw Search::SearchIndexException(MSVS_ERR_ABORTED, "Cancelled building vector index");
            auto chunk = reader->readData(rows_per);
            if (!chunk)
                break;
            check(msvs_bin_index_add(ix, reinterpret_cast<const uint8_t *>(chunk->getData()), chunk->getDataID(), chunk->numData()));
        }
        built = true;
    }

    std::shared_ptr<Search::SearchResult> search(std::shared_ptr<Search::DataSet<bool>> queries, int32_t k, Search::Parameters &,
                                                 bool /*first_stage_only*/, Bitmap * filter) override
    {
        auto res = Search::SearchResult::createTopKHolder(queries->numData(), k);
        std::vector<uint64_t> words;
        size_t nbits = 0;
        if (filter)
        {
            nbits = filter->get_size();
            words.assign((nbits + 63) / 64 + 1, 0);
            memcpy(words.data(), filter->get_bitmap(), filter->byte_size());
        }
        check(msvs_bin_index_search(ix, reinterpret_cast<const uint8_t *>(queries->getData()), (size_t)queries->numData(), (size_t)k,
                                    filter ? words.data() : nullptr, nbits, res->getResultIndices(), res->getResultDistances()));
        return res;
    }
    std::shared_ptr<Search::SearchResult> computeTopDistanceSubset(std::shared_ptr<Search::DataSet<bool>>,
                                                                   std::shared_ptr<Search::SearchResult> first_stage, int32_t) override
    {
        return first_stage;
    }
    bool supportTwoStageSearch() const override { return false; }

    void serialize(Search::IndexDataFileWriter<OS> * writer) override
    {
        StreamIO<IS, OS> s;
        s.writer = writer;
        const msvs_io_t io = s.io();
        check(msvs_bin_index_serialize_io(ix, &io));
        for (auto & o : s.outs)
            o->close();
    }
    void saveDataID(Search::IndexDataFileWriter<OS> *) override {}
    void load(Search::IndexDataFileReader<IS> * reader, std::function<bool()> check_expired) override
    {
        if (check_expired && check_expired())
            throw Search::SearchIndexException(MSVS_ERR_IO, "index files expired before load");
        StreamIO<IS, OS> s;
        s.reader = reader;
        const msvs_io_t io = s.io();
        msvs_bin_index_t * loaded = nullptr;
        check(msvs_bin_index_load_io(&io, &loaded));
        msvs_bin_index_free(ix);
        ix = loaded;
        built = true;
    }
    void loadDataID(Search::IndexDataFileReader<IS> *) override {}

    bool ready() const override { return built; }
    size_t numData() const override { return msvs_bin_index_num_data(ix); }
    Search::IndexResourceUsage getResourceUsage() const override
    {
        Search::IndexResourceUsage u;
        const size_t n = built ? numData() : total_vec, row = dim / 8;
        u.memory_usage_bytes = n * ((row + 15) / 16 * 16 + 4);
        u.disk_usage_bytes = built ? 48 + n * row + 8 + n * 8 : 0;
        u.build_memory_usage_bytes = 2 * n * row + n * 8;
        return u;
    }
    Search::IndexVersion getVersion() const override { return Search::IndexVersion{msvs_index_version()}; }

private:
    Search::IndexType type;
    Search::Metric metric;
    size_t dim, total_vec;
    std::function<bool()> factory_cancel;
    size_t add_chunk = 0;
    bool built = false;
    msvs_bin_index_t * ix = nullptr;
};

}

namespace Search
{

std::string enumToString(IndexType t)
{
    static const char * names[] = {"FLAT", "BinaryFLAT", "IVFFLAT", "IVFPQ", "IVFSQ", "HNSWFLAT", "HNSWSQ", "HNSWPQ", "SCANN", "MSTG",
                                   "BinaryMSTG"};
    return names[(int)t];
}

std::string enumToString(Metric m)
{
    static const char * names[] = {"L2", "IP", "Cosine", "Hamming", "Jaccard"};
    return names[(int)m];
}

Metric getMetricType(const std::string & name, DataType type)
{
    const std::string n = lower(name);
    if (type == DataType::FloatVector)
    {
        if (n == "l2")
            return Metric::L2;
        if (n == "ip")
            return Metric::IP;
        if (n == "cosine")
            return Metric::Cosine;
    }
    else
    {
        if (n == "hamming")
            return Metric::Hamming;
        if (n == "jaccard")
            return Metric::Jaccard;
    }
    throw SearchIndexException(MSVS_ERR_INVALID_ARGUMENT, "unknown metric type `" + name + "`");
}

IndexType getVectorIndexType(const std::string & name, DataType)
{
    const std::string n = lower(name);
    for (int t = 0; t <= (int)IndexType::BinaryMSTG; t++)
        if (lower(enumToString((IndexType)t)) == n)
            return (IndexType)t;
    throw SearchIndexException(MSVS_ERR_INVALID_ARGUMENT, "unknown vector index type `" + name + "`");
}

/// `TYPE DEFAULT` (VIDescriptions.cpp:133-137): the library's pick per data type.  The cloud build's default is MSTG /
/// BinaryMSTG (tests/queries/2_vector_search: `TYPE default('metric_type=IP')`, `default('metric_type=Jaccard')` on binary
/// columns); both are served here (MSTG's partition scan by IVFFLAT).  Forwarding builds ask the original.
std::string getDefaultIndexType(const DataType & search_type)
{
#ifdef MSVS_SEARCH_FORWARD_SUFFIX
    return MSVS_FWD(getDefaultIndexType)(search_type);
#else
    return search_type == DataType::FloatVector ? "MSTG" : "BinaryMSTG";
#endif
}

/// The parameter table (parseVSParameters.cpp:78-222, VIDescriptions.cpp:172-330) of the types libmsvs serves: what
/// msvs_index_create parses (ncentroids, kmeans_iters, train_sample, seed) and what msvs_index_search takes (nprobe; MSTG's
/// `alpha`, seen in the functional tests as distance('alpha=4') / 'alpha=3.7', is accepted and ignored: the stand-in scans
/// nprobe partitions), plus the `metric_type` every type carries (`metric`: the spelling of 00005's IVFFLAT DDL) and MSTG's
/// `disk_mode` (00028: accepted, no effect -- the index lives in HBM).
static const char * const msvs_parameter_table = R"JSON({
"FLAT": {"metric_type": {"type": "string", "case_sensitive": false, "range": [], "candidates": ["L2", "Cosine", "IP"]}},
"IVFFLAT": {"metric_type": {"type": "string", "case_sensitive": false, "range": [], "candidates": ["L2", "Cosine", "IP"]},
            "metric": {"type": "string", "case_sensitive": false, "range": [], "candidates": ["L2", "Cosine", "IP"]},
            "ncentroids": {"type": "int", "case_sensitive": false, "range": [1, 1048576], "candidates": []},
            "kmeans_iters": {"type": "int", "case_sensitive": false, "range": [1, 1000], "candidates": []},
            "train_sample": {"type": "int", "case_sensitive": false, "range": [1, 2147483647], "candidates": []},
            "seed": {"type": "int", "case_sensitive": false, "range": [0, 2147483647], "candidates": []},
            "nprobe": {"type": "int", "case_sensitive": false, "range": [1, 1048576], "candidates": []}},
"MSTG": {"metric_type": {"type": "string", "case_sensitive": false, "range": [], "candidates": ["L2", "Cosine", "IP"]},
         "disk_mode": {"type": "int", "case_sensitive": false, "range": [0, 2], "candidates": []},
         "ncentroids": {"type": "int", "case_sensitive": false, "range": [1, 1048576], "candidates": []},
         "nprobe": {"type": "int", "case_sensitive": false, "range": [1, 1048576], "candidates": []},
         "alpha": {"type": "float", "case_sensitive": false, "range": [1, 4], "candidates": []}},
"BINARYFLAT": {"metric_type": {"type": "string", "case_sensitive": false, "range": [], "candidates": ["Hamming", "Jaccard"]}},
"BINARYMSTG": {"metric_type": {"type": "string", "case_sensitive": false, "range": [], "candidates": ["Hamming", "Jaccard"]},
               "disk_mode": {"type": "int", "case_sensitive": false, "range": [0, 2], "candidates": []},
               "alpha": {"type": "float", "case_sensitive": false, "range": [1, 4], "candidates": []}}
})JSON";

#ifdef MSVS_SEARCH_FORWARD_SUFFIX
/// forwarding build: the original library's table is the reference's truth for every type, including the ones served here
/// (initialised from the original's object: link the original as a shared library this one depends on, so that its
/// initialisers have run -- or make the original's table a constant-initialised string)
/// (cross-translation-unit dynamic initialisation: safe only when the original is a separately loaded shared library whose
/// initialisers have run -- an empty copy, the statically linked case, falls back to this library's own table instead of handing the
/// host's DDL check an empty JSON document)
static std::string forwarded_parameter_table()
{
    const std::string & theirs = MSVS_FWD(MYSCALE_VALID_INDEX_PARAMETER);
    return theirs.empty() ? std::string(msvs_parameter_table) : theirs;
}
const std::string MYSCALE_VALID_INDEX_PARAMETER = forwarded_parameter_table();
#else
const std::string MYSCALE_VALID_INDEX_PARAMETER = msvs_parameter_table;
#endif

template <typename IS, typename OS, typename Bitmap, DataType T>
std::shared_ptr<VectorIndex<IS, OS, Bitmap, T>> createVectorIndex(const std::string & name, IndexType type, Metric metric,
                                                                  size_t dimension, size_t total_vec, const Parameters & params,
                                                                  size_t max_threads, const std::string & cache_prefix,
                                                                  std::function<bool()> check_cancelled)
{
#ifdef MSVS_SEARCH_FORWARD_SUFFIX
    if (!served_by_msvs(type)) // HNSW*, IVFPQ, IVFSQ, SCANN: the original library's object behind the same interface
        return MSVS_FWD(createVectorIndex)<IS, OS, Bitmap, T>(name, type, metric, dimension, total_vec, params, max_threads,
                                                              cache_prefix, std::move(check_cancelled));
#else
    (void)name;
    (void)max_threads;
    (void)cache_prefix;
#endif
    if constexpr (T == DataType::FloatVector)
        return std::make_shared<MsvsVectorIndex<IS, OS, Bitmap>>(type, metric, dimension, total_vec, params, std::move(check_cancelled));
    else
        return std::make_shared<MsvsBinaryIndex<IS, OS, Bitmap>>(type, metric, dimension, total_vec, std::move(check_cancelled));
}

// the instantiation the host uses (VICommon.h:142-143)
template std::shared_ptr<VectorIndex<AbstractIStream, AbstractOStream, DenseBitmap, DataType::FloatVector>>
createVectorIndex<AbstractIStream, AbstractOStream, DenseBitmap, DataType::FloatVector>(const std::string &, IndexType, Metric,
                                                                                       size_t, size_t, const Parameters &,
                                                                                       size_t, const std::string &,
                                                                                       std::function<bool()>);

}

namespace Search
{
// ... and the second one (VICommon.h:142-143 BinaryVI; VIWithDataPart.cpp:431-446)
template std::shared_ptr<VectorIndex<AbstractIStream, AbstractOStream, DenseBitmap, DataType::BinaryVector>>
createVectorIndex<AbstractIStream, AbstractOStream, DenseBitmap, DataType::BinaryVector>(const std::string &, IndexType, Metric,
                                                                                        size_t, size_t, const Parameters &,
                                                                                        size_t, const std::string &,
                                                                                        std::function<bool()>);
}

// ------------------------------------------------------------------------------------------------ seam A2

namespace faiss
{

void knn_inner_product(const float * x, const float * y, size_t d, size_t nx, size_t ny, float_minheap_array_t * res,
                       const IDSelector *)
{
    check(msvs_knn_f32(x, y, d, res->k, nx, ny, MSVS_METRIC_IP, res->ids, res->val));
}

void knn_L2sqr(const float * x, const float * y, size_t d, size_t nx, size_t ny, float_maxheap_array_t * res, const IDSelector *)
{
    check(msvs_knn_f32(x, y, d, res->k, nx, ny, MSVS_METRIC_L2, res->ids, res->val));
}

/// The host passes its FLOAT distance buffer cast to int32_t* (BruteForceSearch.h:99) and reads it back as float
/// (MergeTreeVSManager.cpp:1652-1678 compares, :1507-1525 inserts into a Float32 column; the goldens of 00038 print 4, 8,
/// 12): the counts are therefore stored as float VALUES in that buffer.
void hammings_knn_mc(const uint8_t * a, const uint8_t * b, size_t na, size_t nb, size_t k, size_t ncodes, int32_t * distances,
                     int64_t * labels, const IDSelector *)
{
    check(msvs_knn_bin(a, b, ncodes, k, na, nb, MSVS_METRIC_HAMMING, nullptr, labels, reinterpret_cast<float *>(distances)));
}

}

void jaccard_knn(const uint8_t * a, const uint8_t * b, size_t na, size_t nb, size_t k, size_t ncodes, float * distances,
                 int64_t * labels, const faiss::IDSelector *)
{
    check(msvs_knn_bin(a, b, ncodes, k, na, nb, MSVS_METRIC_JACCARD, nullptr, labels, distances));
}
