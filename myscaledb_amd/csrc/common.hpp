// common.hpp -- host-side plumbing shared by the libmsvs translation units:
// error reporting (thread-local message + status code), RAII device buffers, small param parser.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/msvs.h"

namespace msvs
{

struct Error
{
    int code;
    std::string msg;
};

void set_last_error(const std::string & m);

[[noreturn]] inline void fail(int code, const char * fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw Error{code, buf};
}

#define MSVS_HIP(expr)                                                                                              \
    do                                                                                                              \
    {                                                                                                               \
        hipError_t e_ = (expr);                                                                                     \
        if (e_ != hipSuccess)                                                                                       \
            ::msvs::fail(e_ == hipErrorOutOfMemory ? MSVS_ERR_OUT_OF_MEMORY : MSVS_ERR_DEVICE, "%s failed: %s (%s:%d)", \
                         #expr, hipGetErrorString(e_), __FILE__, __LINE__);                                         \
    } while (0)

/// Runs a C-ABI body, translating exceptions into status codes + msvs_last_error().
template <typename F>
inline int guarded(F && f)
{
    try
    {
        f();
        return MSVS_OK;
    }
    catch (const Error & e)
    {
        set_last_error(e.msg);
        return e.code;
    }
    catch (const std::bad_alloc &)
    {
        set_last_error("host allocation failed");
        return MSVS_ERR_OUT_OF_MEMORY;
    }
    catch (const std::exception & e)
    {
        set_last_error(e.what());
        return MSVS_ERR_DEVICE;
    }
}

/// Non-owning device pointer with DevBuf's `.p` (arena slices used where a DevBuf used to be).
template <typename T>
struct DevView
{
    T * p;
};

/// Owning device allocation.
template <typename T>
struct DevBuf
{
    T * p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf &) = delete;
    DevBuf & operator=(const DevBuf &) = delete;
    DevBuf(DevBuf && o) noexcept : p(o.p), n(o.n)
    {
        o.p = nullptr;
        o.n = 0;
    }
    DevBuf & operator=(DevBuf && o) noexcept
    {
        if (this != &o)
        {
            release();
            p = o.p;
            n = o.n;
            o.p = nullptr;
            o.n = 0;
        }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t count)
    {
        release();
        n = count;
        if (count)
            MSVS_HIP(hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T)));
    }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    size_t bytes() const { return n * sizeof(T); }
};

inline size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }
inline size_t ceil_div(size_t x, size_t m) { return (x + m - 1) / m; }

/// "k=v,k=v" or a flat JSON object {"k":"v","k2":3}: enough for the reference's Search::Parameters
/// (string->string map, VICommon.h:127,186-212).
std::map<std::string, std::string> parse_params(const char * s);
long param_int(const std::map<std::string, std::string> & m, const char * key, long dflt);

}
