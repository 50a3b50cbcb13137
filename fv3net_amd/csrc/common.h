// Shared helpers for libfv3hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/fv3hip.h"

namespace fv3hip {

// Thread-local message of the last failure (fv3hip_last_error()).
char *last_error_buffer();
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define FV3HIP_CHECK_HIP(expr)                                                              \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return ::fv3hip::fail(FV3HIP_EHIP, "%s failed: %s (%s:%d)", #expr,              \
                                  hipGetErrorString(_e), __FILE__, __LINE__);               \
    } while (0)

#define FV3HIP_REQUIRE(cond, ...)                                       \
    do {                                                                \
        if (!(cond)) return ::fv3hip::fail(FV3HIP_EINVAL, __VA_ARGS__); \
    } while (0)

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// After a launch: surface launch-configuration errors without synchronising.
inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FV3HIP_EHIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return FV3HIP_OK;
}

constexpr int kWave = 64;  // gfx950 wavefront

}  // namespace fv3hip
