// common.h — error plumbing shared by the translation units of libmassfuse.so
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "massfuse.h"

namespace mf {

char *error_buffer();   // thread-local, 512 bytes (api.cpp)

inline int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define MF_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess)                                                           \
            return ::mf::fail(MF_ERR_HIP, "%s failed: %s (%s:%d)", #expr,               \
                              hipGetErrorString(_e), __FILE__, __LINE__);               \
    } while (0)

#define MF_LAUNCH_CHECK(name)                                                           \
    do {                                                                                \
        hipError_t _e = hipGetLastError();                                              \
        if (_e != hipSuccess)                                                           \
            return ::mf::fail(MF_ERR_HIP, "launch of %s failed: %s", name,              \
                              hipGetErrorString(_e));                                   \
    } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct DeviceInfo { int cus; int lds_per_cu; };
const DeviceInfo &device_info();   // cached for the current device (api.cpp)

}  // namespace mf
