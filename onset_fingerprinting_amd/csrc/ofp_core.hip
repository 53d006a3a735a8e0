// Status / error reporting and the three legacy symbols of envelope_follower.so.
#include <cstring>
#include <vector>

#include "../../include/ofp_math.h"
#include "ofp_common.h"

namespace ofp {

char* err_buf() {
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace ofp

namespace {

// envelope_follower.c:6-25: one thread per channel walks the rows; row 0 continues
// from the LAST row of y (the previous call's final output).
__global__ __launch_bounds__(64) void k_legacy_ar(const float* __restrict__ x, float* __restrict__ y,
                                                  float attack, float release, int size, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= size) return;
    float yi = y[(int64_t)(n - 1) * size + i];
    for (int j = 0; j < n; ++j) {
        yi = ofp_ar_step(x[(int64_t)j * size + i], yi, attack, release);
        y[(int64_t)j * size + i] = yi;
    }
}

// envelope_follower.c:27-57
__global__ __launch_bounds__(64) void k_legacy_minmax(const float* __restrict__ x, float* mn_io,
                                                      float* mx_io, float alpha_min, float alpha_max,
                                                      float minmin, int n, int C) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float ia_min = ofp_ialpha(alpha_min), ia_max = ofp_ialpha(alpha_max);
    float mn = mn_io[c], mx = mx_io[c];
    for (int i = 0; i < n; ++i) {
        float xi = x[(int64_t)i * C + c];
        mn = ofp_min_step(xi, mn, ia_min, alpha_min, minmin);
        mx = ofp_max_step(xi, mx, ia_max, alpha_max);
    }
    mn_io[c] = mn;
    mx_io[c] = mx;
}

// envelope_follower.c:59-85, one thread per onset
__global__ __launch_bounds__(64) void k_legacy_backtrack(const float* __restrict__ buffer,
                                                         const long* __restrict__ channels, long* deltas,
                                                         float alpha, float tol, long N, long n_onsets,
                                                         long C, long B) {
    long j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_onsets) return;
    float omba = (float)(1.0 - (double)alpha);
    long channel = channels[j];
    long delta = deltas[j];
    long i = B - delta;
    long idx = (N - i) * C + channel;
    float cur = buffer[idx];
    idx -= C;
    float prev = buffer[idx];
    float ps = alpha * prev + omba * cur;
    while (cur > ps && fabsf(ps - prev) > tol && (i + 1 < N)) {
        delta -= 1;
        i += 1;
        idx -= C;
        cur = ps;
        prev = buffer[idx];
        ps = alpha * prev + omba * cur;
    }
    deltas[j] = delta;
}

// scipy.signal.lfilter(b, a, x, axis=0, zi) in float32, direct form II transposed, any order
// <= 8 (detection.py:487-501); one thread per channel; b,a are normalised by a[0] in fp32 first
constexpr int LF_MAX = 8;
struct LfArgs {
    const float* x;
    float* y;
    float* zi;  // [order][C]
    float b[LF_MAX + 1], a[LF_MAX + 1];
    int order, C;
    long n;
};
__global__ __launch_bounds__(64) void k_lfilter(LfArgs p) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.C) return;
    float z[LF_MAX];
    for (int k = 0; k < LF_MAX; ++k) z[k] = k < p.order ? p.zi[k * p.C + c] : 0.0f;
    for (long t = 0; t < p.n; ++t) {
        const float xv = p.x[t * p.C + c];
        float yv;
        if (p.order == 0) {
            yv = xv * p.b[0];
        } else {
            yv = z[0] + p.b[0] * xv;
#pragma unroll
            for (int k = 0; k < LF_MAX - 1; ++k)
                if (k < p.order - 1) z[k] = (z[k + 1] + xv * p.b[k + 1]) - yv * p.a[k + 1];
#pragma unroll
            for (int k = 0; k < LF_MAX; ++k)
                if (k == p.order - 1) z[k] = xv * p.b[k + 1] - yv * p.a[k + 1];
        }
        p.y[t * p.C + c] = yv;
    }
    for (int k = 0; k < LF_MAX; ++k)
        if (k < p.order) p.zi[k * p.C + c] = z[k];
}

struct DevBuf {
    void* p = nullptr;
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
};

#define LEGACY_HIP(call)                                                                      \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            ofp::fail(OFP_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e__));           \
            return;                                                                           \
        }                                                                                     \
    } while (0)

}  // namespace

extern "C" {

int ofp_abi_version(void) { return OFP_ABI_VERSION; }

const char* ofp_last_error(void) { return ofp::err_buf(); }

int ofp_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        ofp::fail(OFP_ERR_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
        return -OFP_ERR_NODEVICE;
    }
    return n;
}

int ofp_device_check(int device, char* buf, int buflen) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return ofp::fail(OFP_ERR_NODEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (buf && buflen > 0) {
        std::strncpy(buf, prop.gcnArchName, buflen - 1);
        buf[buflen - 1] = 0;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return ofp::fail(OFP_ERR_NODEVICE, "device %d is %s; this library is built for gfx950 only", device,
                         prop.gcnArchName);
    return OFP_OK;
}

int ofp_lfilter(const float* x, float* y, const float* b, const float* a, int order, float* zi, long n,
                int n_channels) {
    OFP_REQUIRE(x && y && b && a && (zi || order == 0), "ofp_lfilter: NULL argument");
    OFP_REQUIRE(order >= 0 && order <= LF_MAX, "ofp_lfilter: order %d not in [0, %d]", order, LF_MAX);
    OFP_REQUIRE(a[0] != 0.0f, "ofp_lfilter: a[0] must be non-zero");
    if (n <= 0 || n_channels <= 0) return OFP_OK;
    const size_t bytes = (size_t)n * n_channels * sizeof(float), zb = (size_t)order * n_channels * sizeof(float);
    DevBuf dx, dy, dz;
    OFP_HIP(dx.alloc(bytes));
    OFP_HIP(dy.alloc(bytes));
    OFP_HIP(dz.alloc(zb));
    OFP_HIP(hipMemcpy(dx.p, x, bytes, hipMemcpyHostToDevice));
    if (zb) OFP_HIP(hipMemcpy(dz.p, zi, zb, hipMemcpyHostToDevice));
    LfArgs p;
    p.x = (const float*)dx.p;
    p.y = (float*)dy.p;
    p.zi = (float*)dz.p;
    p.order = order;
    p.C = n_channels;
    p.n = n;
    for (int k = 0; k <= LF_MAX; ++k) {
        p.b[k] = k <= order ? b[k] / a[0] : 0.0f;
        p.a[k] = k <= order ? a[k] / a[0] : 0.0f;
    }
    hipLaunchKernelGGL(k_lfilter, dim3((n_channels + 63) / 64), dim3(64), 0, 0, p);
    OFP_LAUNCH_CHECK("k_lfilter");
    OFP_HIP(hipMemcpy(y, dy.p, bytes, hipMemcpyDeviceToHost));
    if (zb) OFP_HIP(hipMemcpy(zi, dz.p, zb, hipMemcpyDeviceToHost));
    return OFP_OK;
}

void ar_envelope(float* x, float* y, float attack, float release, int size, int num_samples) {
    if (size <= 0 || num_samples <= 0) return;
    size_t bytes = (size_t)size * num_samples * sizeof(float);
    DevBuf dx, dy;
    LEGACY_HIP(dx.alloc(bytes));
    LEGACY_HIP(dy.alloc(bytes));
    LEGACY_HIP(hipMemcpy(dx.p, x, bytes, hipMemcpyHostToDevice));
    LEGACY_HIP(hipMemcpy(dy.p, y, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_legacy_ar, dim3((size + 63) / 64), dim3(64), 0, 0, (const float*)dx.p, (float*)dy.p,
                       attack, release, size, num_samples);
    LEGACY_HIP(hipGetLastError());
    LEGACY_HIP(hipMemcpy(y, dy.p, bytes, hipMemcpyDeviceToHost));
}

void minmax_envelope(float* x, float* min_val, float* max_val, float alpha_min, float alpha_max,
                     float minmin, int n_samples, int n_channels) {
    if (n_channels <= 0) return;
    size_t bytes = (size_t)n_samples * n_channels * sizeof(float);
    size_t cb = (size_t)n_channels * sizeof(float);
    DevBuf dx, dmn, dmx;
    LEGACY_HIP(dx.alloc(bytes));
    LEGACY_HIP(dmn.alloc(cb));
    LEGACY_HIP(dmx.alloc(cb));
    if (bytes) LEGACY_HIP(hipMemcpy(dx.p, x, bytes, hipMemcpyHostToDevice));
    LEGACY_HIP(hipMemcpy(dmn.p, min_val, cb, hipMemcpyHostToDevice));
    LEGACY_HIP(hipMemcpy(dmx.p, max_val, cb, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_legacy_minmax, dim3((n_channels + 63) / 64), dim3(64), 0, 0, (const float*)dx.p,
                       (float*)dmn.p, (float*)dmx.p, alpha_min, alpha_max, minmin, n_samples, n_channels);
    LEGACY_HIP(hipGetLastError());
    LEGACY_HIP(hipMemcpy(min_val, dmn.p, cb, hipMemcpyDeviceToHost));
    LEGACY_HIP(hipMemcpy(max_val, dmx.p, cb, hipMemcpyDeviceToHost));
}

void backtrack_onsets(float* buffer, long* channels, long* deltas, float alpha, float tol,
                      long buffer_length, long n_onsets, long n_channels, long block_size) {
    if (n_onsets <= 0) return;
    size_t bytes = (size_t)buffer_length * n_channels * sizeof(float);
    size_t ob = (size_t)n_onsets * sizeof(long);
    DevBuf db, dc, dd;
    LEGACY_HIP(db.alloc(bytes));
    LEGACY_HIP(dc.alloc(ob));
    LEGACY_HIP(dd.alloc(ob));
    LEGACY_HIP(hipMemcpy(db.p, buffer, bytes, hipMemcpyHostToDevice));
    LEGACY_HIP(hipMemcpy(dc.p, channels, ob, hipMemcpyHostToDevice));
    LEGACY_HIP(hipMemcpy(dd.p, deltas, ob, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_legacy_backtrack, dim3((unsigned)((n_onsets + 63) / 64)), dim3(64), 0, 0,
                       (const float*)db.p, (const long*)dc.p, (long*)dd.p, alpha, tol, buffer_length, n_onsets,
                       n_channels, block_size);
    LEGACY_HIP(hipGetLastError());
    LEGACY_HIP(hipMemcpy(deltas, dd.p, ob, hipMemcpyDeviceToHost));
}

}  // extern "C"
