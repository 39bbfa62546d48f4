// STUB (see SearchIndexCommon.h): the stream / file-set / build-feed interfaces of the absent library as the host uses
// them -- VectorIndexIO.h:25-166 (AbstractIStream / AbstractOStream overrides), VIWithDataPart.cpp:461-464,688-691
// (IndexDataFileWriter / Reader with an opener callback), VIPartReader.h:38-167,296-303 (IndexSourceDataReader, DataChunk).
#pragma once

#include <functional>
#include <ios>
#include <memory>
#include <string>

#include "../SearchIndexCommon.h"

namespace Search
{

class AbstractIStream
{
public:
    virtual ~AbstractIStream() = default;
    virtual AbstractIStream & read(char * s, std::streamsize count) = 0;
    virtual bool is_open() const = 0;
    virtual bool fail() const = 0;
    virtual bool eof() const = 0;
    virtual std::streamsize gcount() const = 0;
    virtual explicit operator bool() const = 0;
    virtual AbstractIStream & seekg(std::streampos offset, std::ios_base::seekdir dir) = 0;
};

class AbstractOStream
{
public:
    virtual ~AbstractOStream() = default;
    virtual AbstractOStream & write(const char * s, std::streamsize count) = 0;
    virtual bool good() = 0;
    virtual void close() = 0;
    virtual AbstractOStream & seekp(std::streampos offset, std::ios_base::seekdir dir) = 0;
};

/// The file set of one index: every file is opened through the host's callback as <path_prefix><name>
/// (path_prefix = ".../<index_name>-", VIWithDataPart.cpp:458-464; the host's files end in .vidx3, VICommon.h:55).
template <typename OS>
class IndexDataFileWriter
{
public:
    using Opener = std::function<std::shared_ptr<OS>(const std::string & name, std::ios::openmode mode)>;
    IndexDataFileWriter(const std::string & path_prefix_, Opener opener_) : path_prefix(path_prefix_), opener(std::move(opener_)) {}
    std::shared_ptr<OS> open(const std::string & file) { return opener(path_prefix + file + ".vidx3", std::ios::out | std::ios::binary); }

private:
    std::string path_prefix;
    Opener opener;
};

template <typename IS>
class IndexDataFileReader
{
public:
    using Opener = std::function<std::shared_ptr<IS>(const std::string & name, std::ios::openmode mode)>;
    IndexDataFileReader(const std::string & path_prefix_, Opener opener_) : path_prefix(path_prefix_), opener(std::move(opener_)) {}
    std::shared_ptr<IS> open(const std::string & file) { return opener(path_prefix + file + ".vidx3", std::ios::in | std::ios::binary); }

private:
    std::string path_prefix;
    Opener opener;
};

/// Build feed: chunks of dense rows + their ids (VIPartReader::readDataImpl).
template <typename T>
class IndexSourceDataReader
{
public:
    class DataChunk
    {
    public:
        DataChunk(T * data_, size_t n_, size_t dim_, std::function<void()> deleter_)
            : data(data_), n(n_), dim(dim_), deleter(std::move(deleter_))
        {
        }
        ~DataChunk()
        {
            if (deleter)
                deleter();
            if (id_deleter)
                id_deleter();
        }
        void setDataID(idx_t * ids_, std::function<void()> deleter_)
        {
            ids = ids_;
            id_deleter = std::move(deleter_);
        }
        T * getData() const { return data; }
        idx_t * getDataID() const { return ids; }
        size_t numData() const { return n; }
        size_t dimension() const { return dim; }

    private:
        T * data;
        idx_t * ids = nullptr;
        size_t n, dim;
        std::function<void()> deleter, id_deleter;
    };

    virtual ~IndexSourceDataReader() = default;
    virtual size_t numDataRead() const = 0;
    virtual size_t dataDimension() const = 0;
    virtual bool eof() = 0;
    virtual void seekg(std::streamsize offset, std::ios::seekdir dir) = 0;
    virtual std::shared_ptr<DataChunk> sampleData(size_t n) = 0;
    /// what the library calls while building
    std::shared_ptr<DataChunk> readData(size_t n) { return readDataImpl(n); }

protected:
    virtual std::shared_ptr<DataChunk> readDataImpl(size_t n) = 0;
};

}
