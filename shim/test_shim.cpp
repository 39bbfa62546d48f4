// test_shim.cpp -- drives shim/HostShim.cpp the way the reference's host code drives the absent library
// (call-site pattern of VIWithDataPart.cpp:415-446 create, VIWithDataPart.h:332-337 build, VIWithDataPart.cpp:461-479
// serialize, :688-700 load, :922-926 search; BruteForceSearch.h:80-104 brute force), on inputs written by
// tests/test_shim.py, and writes the results back for the comparison with the oracle.
//   usage: test_shim <dir>     reads  <dir>/{meta.txt, x.bin, q.bin, alive.bin, bx.bin, bq.bin, balive.bin}
//                              writes <dir>/{out_*.bin, files.txt}
#include <SearchIndex/VectorIndex.h>
#include <faiss/utils/distances.h>

#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

using IS = Search::AbstractIStream;
using OS = Search::AbstractOStream;
using Bitmap = Search::DenseBitmap;
using FloatVI = Search::VectorIndex<IS, OS, Bitmap, Search::DataType::FloatVector>;

namespace
{

// ---- the host's disk streams, over an in-memory "disk" (what VectorIndexWriter / VectorIndexReader are over IDisk)
std::map<std::string, std::string> g_disk;

class MemWriter : public OS
{
public:
    explicit MemWriter(std::string name_) : name(std::move(name_)) { g_disk[name].clear(); }
    OS & write(const char * s, std::streamsize count) override
    {
        g_disk[name].append(s, (size_t)count);
        return *this;
    }
    bool good() override { return true; }
    void close() override { closed = true; }
    OS & seekp(std::streampos, std::ios_base::seekdir) override { return *this; }
    bool closed = false;

private:
    std::string name;
};

class MemReader : public IS
{
public:
    explicit MemReader(const std::string & name)
    {
        auto it = g_disk.find(name);
        if (it != g_disk.end())
            data = &it->second;
    }
    IS & read(char * s, std::streamsize count) override
    {
        last = 0;
        if (data)
        {
            last = std::min<size_t>((size_t)count, data->size() - pos);
            memcpy(s, data->data() + pos, last);
            pos += last;
        }
        return *this;
    }
    bool is_open() const override { return data != nullptr; }
    bool fail() const override { return data == nullptr; }
    bool eof() const override { return !data || pos >= data->size(); }
    std::streamsize gcount() const override { return (std::streamsize)last; }
    explicit operator bool() const override { return data != nullptr; }
    IS & seekg(std::streampos off, std::ios_base::seekdir) override
    {
        pos = (size_t)off;
        return *this;
    }

private:
    const std::string * data = nullptr;
    size_t pos = 0, last = 0;
};

// ---- the build feed: dense chunks + ids, like VIPartReader::readDataImpl
class PartReader : public Search::IndexSourceDataReader<float>
{
public:
    PartReader(const std::vector<float> & x_, size_t n_, size_t dim_) : x(x_), n(n_), dim(dim_) {}
    size_t numDataRead() const override { return pos; }
    size_t dataDimension() const override { return dim; }
    bool eof() override { return pos == n; }
    void seekg(std::streamsize, std::ios::seekdir) override { throw std::runtime_error("seekg() is not implemented"); }
    std::shared_ptr<DataChunk> sampleData(size_t m) override
    {
        const size_t save = pos;
        pos = 0;
        auto c = readDataImpl(m);
        pos = save;
        return c;
    }

protected:
    std::shared_ptr<DataChunk> readDataImpl(size_t m) override
    {
        m = std::min(m, n - pos);
        if (m == 0)
            return nullptr;
        float * data = new float[m * dim];
        Search::idx_t * ids = new Search::idx_t[m];
        memcpy(data, x.data() + pos * dim, m * dim * sizeof(float));
        for (size_t i = 0; i < m; i++)
            ids[i] = (Search::idx_t)(pos + i);
        auto chunk = std::make_shared<DataChunk>(data, m, dim, [=]() { delete[] data; });
        chunk->setDataID(ids, [=]() { delete[] ids; });
        pos += m;
        return chunk;
    }

private:
    const std::vector<float> & x;
    size_t n, dim, pos = 0;
};

template <typename T>
std::vector<T> slurp(const std::string & path)
{
    std::ifstream f(path, std::ios::binary);
    std::string s((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::vector<T> v(s.size() / sizeof(T));
    memcpy(v.data(), s.data(), v.size() * sizeof(T));
    return v;
}

template <typename T>
void dump(const std::string & path, const T * p, size_t n)
{
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char *>(p), (std::streamsize)(n * sizeof(T)));
}

}

int main(int argc, char ** argv)
{
    if (argc < 2)
        return 2;
    const std::string dir = argv[1];
    size_t n, d, nq, k, nlist, nb_rows, nb_bytes, nb_q;
    std::string metric_name, type_name;
    {
        std::ifstream m(dir + "/meta.txt");
        m >> n >> d >> nq >> k >> nlist >> metric_name >> type_name >> nb_rows >> nb_bytes >> nb_q;
    }
    auto x = slurp<float>(dir + "/x.bin");
    auto q = slurp<float>(dir + "/q.bin");
    auto alive = slurp<uint8_t>(dir + "/alive.bin"); // one byte per row
    try
    {
        // ---- seam A1: create / build / serialize (VIWithDataPart.cpp:395-525)
        const auto metric = Search::getMetricType(metric_name, Search::DataType::FloatVector);
        const auto type = Search::getVectorIndexType(type_name, Search::DataType::FloatVector);
        Search::Parameters des;
        des.setParam("ncentroids", nlist);
        des.setParam("metric_type", metric_name);
        auto cancel = []() { return false; };
        auto index = Search::createVectorIndex<IS, OS, Bitmap, Search::DataType::FloatVector>("v1", type, metric, d, n, des, 8,
                                                                                               "cache/", cancel);
        index->setTrainDataChunkSize((size_t)100 << 20);
        index->setAddDataChunkSize((size_t)1 << 20); // small: several add chunks
        const auto before = index->getResourceUsage();
        if (index->ready() || before.build_memory_usage_bytes == 0)
            throw std::runtime_error("fresh index state is wrong");
        PartReader part(x, n, d);
        index->build(&part, 4, cancel);
        if (!index->ready() || index->numData() != n)
            throw std::runtime_error("build did not produce a ready index");
        auto file_writer = Search::IndexDataFileWriter<OS>(
            "part/v1-", [&](const std::string & name, std::ios::openmode) { return std::make_shared<MemWriter>(name); });
        index->serialize(&file_writer);
        index->saveDataID(&file_writer);
        const std::string version = index->getVersion().toString();
        const auto usage = index->getResourceUsage();
        size_t disk = 0;
        {
            std::ofstream files(dir + "/files.txt");
            for (const auto & f : g_disk)
            {
                files << f.first << " " << f.second.size() << "\n";
                disk += f.second.size();
            }
            files << "version " << version << "\n";
        }
        if (usage.disk_usage_bytes != disk)
            throw std::runtime_error("getResourceUsage().disk_usage_bytes != bytes written");

        // ---- load into a fresh object (VIWithDataPart.cpp:650-700), then search (:922-926)
        Search::Parameters load_params = des;
        load_params.setParam("load_index_version", version);
        auto loaded = Search::createVectorIndex<IS, OS, Bitmap, Search::DataType::FloatVector>("v1", type, metric, d, n,
                                                                                                load_params, 8, "cache/", cancel);
        auto file_reader = Search::IndexDataFileReader<IS>(
            "part/v1-", [&](const std::string & name, std::ios::openmode) { return std::make_shared<MemReader>(name); });
        loaded->load(&file_reader, []() { return false; });
        loaded->loadDataID(&file_reader);
        if (loaded->numData() != n)
            throw std::runtime_error("load: numData mismatch");
        index.reset(); // the searches below run on the LOADED index

        auto queries = std::make_shared<Search::DataSet<float>>(q.data(), (int64_t)nq, (int64_t)d);
        Search::Parameters sp;
        sp.setParam("nprobe", nlist); // exhaustive: comparable with the exact scan whatever the clustering
        auto r1 = loaded->search(queries, (int32_t)k, sp, false, nullptr);
        dump(dir + "/out_ids.bin", r1->getResultIndices(), nq * k);
        dump(dir + "/out_dis.bin", r1->getResultDistances(), nq * k);
        if (r1->numQueries() != (int64_t)nq || r1->getNumCandidates() != (int64_t)k)
            throw std::runtime_error("SearchResult shape");
        auto filter = std::make_shared<Bitmap>(n);
        for (size_t i = 0; i < n; i++)
            if (alive[i])
                filter->set(i);
        auto deleted = std::make_shared<Bitmap>(n, true); // delete bitmap: all alive (VIWithDataPart.cpp:903-908)
        auto merged = deleted->all() ? filter : Search::intersectDenseBitmaps(filter, deleted);
        auto r2 = loaded->search(queries, (int32_t)k, sp, false, merged.get());
        dump(dir + "/out_ids_f.bin", r2->getResultIndices(), nq * k);
        dump(dir + "/out_dis_f.bin", r2->getResultDistances(), nq * k);

        // ---- cancellation (VIWithDataPart.cpp:425-430: the factory's callback; VIPartReader.h:175-176: the host's own error):
        // (a) the callback turns true after two add chunks, (b) it turns true in the middle of the k-means iterations (polled
        // inside the library) -- both builds must leave with code DB::ErrorCodes::ABORTED = 236 and an unbuilt index
        for (int scenario = 0; scenario < 2; scenario++)
        {
            int calls = 0;
            // scenario 0: build()'s chunk loop polls once per chunk (+ the library once per add): true from the 5th call on;
            // scenario 1: the FACTORY callback, true from its 3rd call on -- for IVFFLAT that is inside msvs_index_train
            const int limit = scenario == 0 ? 5 : 3;
            std::function<bool()> counting = [&calls, limit]() { return ++calls >= limit; };
            std::function<bool()> never = []() { return false; };
            auto doomed = Search::createVectorIndex<IS, OS, Bitmap, Search::DataType::FloatVector>(
                "v3", type, metric, d, n, des, 8, "cache/", scenario == 1 ? counting : never);
            doomed->setAddDataChunkSize((size_t)64 << 10); // many small chunks
            PartReader part2(x, n, d);
            int code = 0;
            std::string what;
            try
            {
                doomed->build(&part2, 4, scenario == 0 ? counting : never);
            }
            catch (const SearchIndexException & e)
            {
                code = e.getCode();
                what = e.what();
            }
            if (code != 236 || what.find("Cancelled building vector index") == std::string::npos || doomed->ready())
                throw std::runtime_error("cancelled build (scenario " + std::to_string(scenario) + "): code " + std::to_string(code) + " `" + what + "`");
            std::cout << "cancel_scenario_" << scenario << " code " << code << " after " << calls << " polls\n";
        }
        // an expired load (VIWithDataPart.cpp:698: check_index_expired) leaves the object as it was
        {
            auto fresh = Search::createVectorIndex<IS, OS, Bitmap, Search::DataType::FloatVector>("v1", type, metric, d, n, load_params, 8,
                                                                                                   "cache/", cancel);
            auto fr = Search::IndexDataFileReader<IS>(
                "part/v1-", [&](const std::string & name, std::ios::openmode) { return std::make_shared<MemReader>(name); });
            bool expired = false;
            try
            {
                fresh->load(&fr, []() { return true; });
            }
            catch (const SearchIndexException & e)
            {
                expired = e.getCode() == 236;
            }
            if (!expired || fresh->ready())
                throw std::runtime_error("expired load did not abort");
        }
        // the parameter table and the default type the DDL / search-argument checks read (parseVSParameters.cpp:78, VIDescriptions.cpp:133,172)
        {
            std::ofstream t(dir + "/param_table.json");
            t << Search::MYSCALE_VALID_INDEX_PARAMETER;
            std::ofstream dflt(dir + "/default_types.txt");
            dflt << Search::getDefaultIndexType(Search::DataType::FloatVector) << " "
                 << Search::getDefaultIndexType(Search::DataType::BinaryVector) << "\n";
        }

        // an index type libmsvs does not serve must surface as SearchIndexException (-> VIException in the host)
        bool threw = false;
        try
        {
            Search::createVectorIndex<IS, OS, Bitmap, Search::DataType::FloatVector>("v2", Search::IndexType::HNSWFLAT, metric, d, n,
                                                                                      des, 8, "cache/", cancel);
        }
        catch (const SearchIndexException & e)
        {
            threw = e.getCode() != 0;
        }
        if (!threw)
            throw std::runtime_error("unsupported index type did not throw");

        // ---- seam A2: the four brute-force calls (BruteForceSearch.h:80-104)
        {
            std::vector<int64_t> ids(nq * k);
            std::vector<float> dis(nq * k);
            faiss::float_maxheap_array_t l2 = {nq, k, ids.data(), dis.data()};
            faiss::knn_L2sqr(q.data(), x.data(), d, nq, n, &l2, nullptr);
            dump(dir + "/bf_l2_ids.bin", ids.data(), ids.size());
            dump(dir + "/bf_l2_dis.bin", dis.data(), dis.size());
            faiss::float_minheap_array_t ip = {nq, k, ids.data(), dis.data()};
            faiss::knn_inner_product(q.data(), x.data(), d, nq, n, &ip, nullptr);
            dump(dir + "/bf_ip_ids.bin", ids.data(), ids.size());
            dump(dir + "/bf_ip_dis.bin", dis.data(), dis.size());
        }
        {
            auto bx = slurp<uint8_t>(dir + "/bx.bin");
            auto bq = slurp<uint8_t>(dir + "/bq.bin");
            std::vector<int64_t> ids(nb_q * k);
            std::vector<float> dis(nb_q * k);
            faiss::hammings_knn_mc(bq.data(), bx.data(), nb_q, nb_rows, k, nb_bytes, reinterpret_cast<int32_t *>(dis.data()),
                                   ids.data(), nullptr);
            dump(dir + "/bf_ham_ids.bin", ids.data(), ids.size());
            dump(dir + "/bf_ham_dis.bin", dis.data(), dis.size());
            jaccard_knn(bq.data(), bx.data(), nb_q, nb_rows, k, nb_bytes, dis.data(), ids.data(), nullptr);
            dump(dir + "/bf_jac_ids.bin", ids.data(), ids.size());
            dump(dir + "/bf_jac_dis.bin", dis.data(), dis.size());
        }
        // ---- seam A1 for BinaryVector (VIWithDataPart.cpp:431-446 create, :928-935 search): BinaryFLAT over the same binary rows,
        // built from FixedString chunks with ids, serialised, loaded, searched with and without a filter bitmap
        {
            using BinaryVI = Search::VectorIndex<IS, OS, Bitmap, Search::DataType::BinaryVector>;
            auto bx = slurp<uint8_t>(dir + "/bx.bin");
            auto bq = slurp<uint8_t>(dir + "/bq.bin");
            auto balive = slurp<uint8_t>(dir + "/balive.bin");
            class BinReader : public Search::IndexSourceDataReader<bool>
            {
            public:
                BinReader(const std::vector<uint8_t> & x_, size_t n_, size_t nbytes_) : x(x_), n(n_), nbytes(nbytes_) {}
                size_t numDataRead() const override { return pos; }
                size_t dataDimension() const override { return nbytes * 8; }
                bool eof() override { return pos == n; }
                void seekg(std::streamsize, std::ios::seekdir) override { throw std::runtime_error("seekg() is not implemented"); }
                std::shared_ptr<DataChunk> sampleData(size_t) override { return nullptr; }

            protected:
                std::shared_ptr<DataChunk> readDataImpl(size_t m) override
                {
                    m = std::min(m, n - pos);
                    if (m == 0)
                        return nullptr;
                    bool * data = new bool[m * nbytes]; // one byte = 8 bits (VIPartReader.h:262-266)
                    Search::idx_t * ids = new Search::idx_t[m];
                    memcpy(data, x.data() + pos * nbytes, m * nbytes);
                    for (size_t i = 0; i < m; i++)
                        ids[i] = (Search::idx_t)(pos + i);
                    auto chunk = std::make_shared<DataChunk>(data, m, nbytes * 8, [=]() { delete[] data; });
                    chunk->setDataID(ids, [=]() { delete[] ids; });
                    pos += m;
                    return chunk;
                }

            private:
                const std::vector<uint8_t> & x;
                size_t n, nbytes, pos = 0;
            };
            auto cancel2 = []() { return false; };
            for (const char * mname : {"Hamming", "Jaccard"})
            {
                const auto bmetric = Search::getMetricType(mname, Search::DataType::BinaryVector);
                const auto btype = Search::getVectorIndexType("BinaryFLAT", Search::DataType::BinaryVector);
                Search::Parameters bdes;
                std::shared_ptr<BinaryVI> bindex = Search::createVectorIndex<IS, OS, Bitmap, Search::DataType::BinaryVector>(
                    "b1", btype, bmetric, nb_bytes * 8, nb_rows, bdes, 8, "cache/", cancel2);
                bindex->setAddDataChunkSize(nb_bytes * 700); // several chunks
                BinReader breader(bx, nb_rows, nb_bytes);
                bindex->build(&breader, 4, cancel2);
                if (!bindex->ready() || bindex->numData() != nb_rows)
                    throw std::runtime_error("binary build did not produce a ready index");
                g_disk.clear();
                auto bw = Search::IndexDataFileWriter<OS>(
                    "part/b1-", [&](const std::string & name, std::ios::openmode) { return std::make_shared<MemWriter>(name); });
                bindex->serialize(&bw);
                bindex->saveDataID(&bw);
                auto bloaded = Search::createVectorIndex<IS, OS, Bitmap, Search::DataType::BinaryVector>("b1", btype, bmetric, nb_bytes * 8,
                                                                                                        nb_rows, bdes, 8, "cache/", cancel2);
                auto br = Search::IndexDataFileReader<IS>(
                    "part/b1-", [&](const std::string & name, std::ios::openmode) { return std::make_shared<MemReader>(name); });
                bloaded->load(&br, []() { return false; });
                bloaded->loadDataID(&br);
                if (bloaded->numData() != nb_rows)
                    throw std::runtime_error("binary load: numData mismatch");
                bindex.reset();
                auto bqueries = std::make_shared<Search::DataSet<bool>>(reinterpret_cast<bool *>(bq.data()), (int64_t)nb_q,
                                                                        (int64_t)(nb_bytes * 8));
                Search::Parameters bsp;
                auto b1 = bloaded->search(bqueries, (int32_t)k, bsp, false, nullptr);
                const std::string tag = mname[0] == 'H' ? "ham" : "jac";
                dump(dir + "/bi_" + tag + "_ids.bin", b1->getResultIndices(), nb_q * k);
                dump(dir + "/bi_" + tag + "_dis.bin", b1->getResultDistances(), nb_q * k);
                auto bfilter = std::make_shared<Bitmap>(nb_rows);
                for (size_t i = 0; i < nb_rows; i++)
                    if (balive[i])
                        bfilter->set(i);
                auto b2 = bloaded->search(bqueries, (int32_t)k, bsp, false, bfilter.get());
                dump(dir + "/bi_" + tag + "_ids_f.bin", b2->getResultIndices(), nb_q * k);
                dump(dir + "/bi_" + tag + "_dis_f.bin", b2->getResultDistances(), nb_q * k);
            }
        }
    }
    catch (const std::exception & e)
    {
        std::cerr << "test_shim failed: " << e.what() << "\n";
        return 1;
    }
    std::cout << "test_shim ok\n";
    return 0;
}
