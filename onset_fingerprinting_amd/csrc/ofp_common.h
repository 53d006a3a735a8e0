// Shared host-side helpers of libonsetfp.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/onsetfp.h"

namespace ofp {

// thread-local last-error message (ofp_last_error)
char* err_buf();
int fail(int code, const char* fmt, ...);

#define OFP_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t e__ = (call);                                                       \
        if (e__ != hipSuccess)                                                         \
            return ofp::fail(OFP_ERR_HIP, "%s failed: %s (%s:%d)", #call,              \
                             hipGetErrorString(e__), __FILE__, __LINE__);              \
    } while (0)

#define OFP_LAUNCH_CHECK(name)                                                         \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess)                                                         \
            return ofp::fail(OFP_ERR_HIP, "launch of %s failed: %s", name,             \
                             hipGetErrorString(e__));                                  \
    } while (0)

#define OFP_REQUIRE(cond, ...)                                                         \
    do {                                                                               \
        if (!(cond)) return ofp::fail(OFP_ERR_INVALID, __VA_ARGS__);                   \
    } while (0)

__host__ __device__ inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
__host__ __device__ inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace ofp
