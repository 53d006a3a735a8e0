// Shared host-side helpers of libonsetfp.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
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

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device's function object, and the launch
// functions are called from several host threads: the largest size already set is remembered per device, atomically
// (a process that drives a second GPU sets it there too).
struct LdsAttrCache {
    std::atomic<size_t> set[16];
    LdsAttrCache() {
        for (auto& v : set) v.store(0);
    }
};
inline int ensure_dynamic_lds(const void* fn, size_t lds, LdsAttrCache& cache, size_t threshold = 65536) {
    if (lds <= threshold) return OFP_OK;
    int dev = 0;
    OFP_HIP(hipGetDevice(&dev));
    const bool slot = dev >= 0 && dev < 16;
    if (slot && cache.set[dev].load(std::memory_order_acquire) >= lds) return OFP_OK;
    OFP_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (slot) {
        size_t cur = cache.set[dev].load(std::memory_order_relaxed);
        while (cur < lds && !cache.set[dev].compare_exchange_weak(cur, lds, std::memory_order_release)) {
        }
    }
    return OFP_OK;
}

__host__ __device__ inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
__host__ __device__ inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace ofp
