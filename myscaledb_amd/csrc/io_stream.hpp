// io_stream.hpp -- a named file of an index file set, opened through the caller's msvs_io_t (include/msvs.h: the host's
// IndexDataFileWriter / IndexDataFileReader stream openers, VectorIndexIO.h:25-166); shared by the float and the binary index.
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdint>

#include "../../include/msvs.h"
#include "common.hpp"

namespace msvs
{

constexpr size_t IO_CHUNK = (size_t)64 << 20;

struct IoStream
{
    const msvs_io_t * io;
    void * h;
    const char * name;
    IoStream(const msvs_io_t * io_, const char * name_, int write) : io(io_), h(nullptr), name(name_)
    {
        if (!io || !io->open || !io->close || (write ? !io->write : !io->read))
            fail(MSVS_ERR_INVALID_ARGUMENT, "msvs_io_t lacks a callback");
        h = io->open(io->ctx, name, write);
        if (!h)
            fail(MSVS_ERR_IO, "cannot open index file `%s` for %s", name, write ? "writing" : "reading");
    }
    ~IoStream()
    {
        if (h)
            (void)io->close(io->ctx, h);
    }
    void write(const void * p, size_t n)
    {
        const char * c = static_cast<const char *>(p);
        while (n)
        {
            const size_t m = std::min(n, IO_CHUNK);
            if (io->write(io->ctx, h, c, m) != (int64_t)m)
                fail(MSVS_ERR_IO, "short write to index file `%s`", name);
            c += m;
            n -= m;
        }
    }
    void read(void * p, size_t n)
    {
        char * c = static_cast<char *>(p);
        while (n)
        {
            const int64_t got = io->read(io->ctx, h, c, std::min(n, IO_CHUNK));
            if (got <= 0)
                fail(MSVS_ERR_IO, "index file `%s` is truncated", name);
            c += got;
            n -= (size_t)got;
        }
    }
    void finish()
    {
        void * t = h;
        h = nullptr;
        if (io->close(io->ctx, t) != 0)
            fail(MSVS_ERR_IO, "closing index file `%s` failed", name);
    }
};


}
