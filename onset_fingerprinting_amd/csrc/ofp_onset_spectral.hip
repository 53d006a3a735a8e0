// Spectral-flux onset detector on the device:
// detect_onsets_spectral (reference detection.py:89-128).  The STFT is ofp_stft_power on the
// centre-padded signal; this file adds the A-weighted positive spectral flux, the exact order
// statistic the percentile normalisation needs, and librosa.util.peak_pick (restated from
// librosa's published definition: the library is absent, parity unpinned).
#include <algorithm>

#include "ofp_common.h"

namespace {

using ofp::cdiv;

// oe[t] = mean_k max(0, w_k * mag[t+1][k] - w_k * mag[t][k]),  mag = sqrt(power)   (detection.py:106-110)
// one wave per frame pair, lanes over bins; the weighting is rounded to fp32 before the
// difference, as the reference's in-place float32 multiply does.
__global__ __launch_bounds__(256) void k_spectral_flux(const float* __restrict__ power, int64_t n_frames, int n_bins,
                                                       const float* __restrict__ w, float* __restrict__ oe) {
    const int lane = threadIdx.x & 63;
    const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (t >= n_frames - 1) return;
    const float* p0 = power + t * n_bins;
    const float* p1 = p0 + n_bins;
    float acc = 0.0f;
    for (int k = lane; k < n_bins; k += 64) {
        const float a = __fsqrt_rn(p0[k]) * w[k];
        const float b = __fsqrt_rn(p1[k]) * w[k];
        acc += fmaxf(0.0f, b - a);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) oe[t] = acc / (float)n_bins;
}

// Value of ascending rank `rank` among n non-negative floats (their bit patterns order like
// unsigned integers): 4 rounds of 8-bit radix selection, one workgroup.
__global__ __launch_bounds__(1024) void k_select_rank(const float* __restrict__ v, int64_t n, int64_t rank,
                                                      float* __restrict__ out) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned int s_prefix;
    __shared__ long long s_rank;
    unsigned int prefix = 0;  // bits decided so far (high to low)
    long long r = rank;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        const unsigned int himask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
            const unsigned int u = __float_as_uint(v[i]);
            if ((u & himask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            long long rr = r;
            int b = 0;
            for (; b < 256; ++b) {
                if (rr < (long long)hist[b]) break;
                rr -= hist[b];
            }
            s_prefix = prefix | ((unsigned int)min(b, 255) << shift);
            s_rank = rr;
        }
        __syncthreads();
        prefix = s_prefix;
        r = s_rank;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = __uint_as_float(prefix);
}

// in place: x[i] /= scale[0]   (detection.py:111; scale is computed on the device)
__global__ __launch_bounds__(256) void k_divide(float* __restrict__ x, int64_t n, const float* __restrict__ scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = __fdiv_rn(x[i], scale[0]);
}

// librosa.util.peak_pick, first two conditions: x[i] == max(x[i-pre_max : i+post_max]) and
// x[i] >= mean(x[i-pre_avg : i+post_avg]) + delta, windows clipped to the array.
__global__ __launch_bounds__(256) void k_peak_flags(const float* __restrict__ x, int64_t n, int pre_max, int post_max,
                                                    int pre_avg, int post_avg, float delta,
                                                    uint8_t* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    float mx = v;
    for (int64_t k = max<int64_t>(0, i - pre_max); k < min<int64_t>(n, i + post_max); ++k) mx = fmaxf(mx, x[k]);
    double s = 0.0;
    const int64_t a0 = max<int64_t>(0, i - pre_avg), a1 = min<int64_t>(n, i + post_avg);
    for (int64_t k = a0; k < a1; ++k) s += (double)x[k];
    const double mean = a1 > a0 ? s / (double)(a1 - a0) : (double)v;
    flag[i] = (v == mx && (double)v >= mean + (double)delta && v > 0.0f) ? 1 : 0;
}

// third condition: i - previous_peak > wait.  One wave walks the flags in order.
__global__ __launch_bounds__(64) void k_peak_wait(const uint8_t* __restrict__ flag, int64_t n, int64_t wait,
                                                  int64_t* __restrict__ peaks, int64_t cap, int64_t* __restrict__ count) {
    const int lane = threadIdx.x;
    int64_t last = INT64_MIN / 2, c = 0;
    for (int64_t i0 = 0; i0 < n; i0 += 64) {
        const int64_t i = i0 + lane;
        unsigned long long m = __ballot(i < n && flag[i] != 0);
        while (m) {
            const int l = __builtin_ctzll(m);
            const int64_t p = i0 + l;
            if (p > last + wait) {
                if (lane == 0 && c < cap) peaks[c] = p;
                ++c;
                last = p;
            }
            m &= m - 1;
        }
    }
    if (lane == 0) count[0] = c;
}

}  // namespace

extern "C" {

int ofp_spectral_flux(const float* d_power, int64_t n_frames, int32_t n_bins, const float* d_weight, float* d_oe,
                      void* stream) {
    if (n_frames < 2) return OFP_OK;
    OFP_REQUIRE(d_power && d_weight && d_oe && n_bins >= 1, "ofp_spectral_flux: bad argument");
    hipLaunchKernelGGL(k_spectral_flux, dim3((unsigned)cdiv(n_frames - 1, 4)), dim3(256), 0, (hipStream_t)stream,
                       d_power, n_frames, n_bins, d_weight, d_oe);
    OFP_LAUNCH_CHECK("k_spectral_flux");
    return OFP_OK;
}

int ofp_select_rank(const float* d_v, int64_t n, int64_t rank, float* d_out, void* stream) {
    OFP_REQUIRE(d_v && d_out && n >= 1 && rank >= 0 && rank < n, "ofp_select_rank: bad argument (n=%lld rank=%lld)",
                (long long)n, (long long)rank);
    hipLaunchKernelGGL(k_select_rank, dim3(1), dim3(1024), 0, (hipStream_t)stream, d_v, n, rank, d_out);
    OFP_LAUNCH_CHECK("k_select_rank");
    return OFP_OK;
}

int ofp_scale_inverse(float* d_x, int64_t n, const float* d_scale, void* stream) {
    if (n == 0) return OFP_OK;
    OFP_REQUIRE(d_x && d_scale && n > 0, "ofp_scale_inverse: bad argument");
    hipLaunchKernelGGL(k_divide, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, d_x, n, d_scale);
    OFP_LAUNCH_CHECK("k_divide");
    return OFP_OK;
}

int ofp_peak_pick(const float* d_x, int64_t n, int32_t pre_max, int32_t post_max, int32_t pre_avg, int32_t post_avg,
                  float delta, int64_t wait, int64_t* d_peaks, int64_t cap, int64_t* d_count, uint8_t* d_flags,
                  void* stream_) {
    OFP_REQUIRE(d_x && d_peaks && d_count && d_flags && n >= 0 && cap >= 0 && pre_max >= 0 && post_max >= 1 &&
                    pre_avg >= 0 && post_avg >= 1 && wait >= 0,
                "ofp_peak_pick: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    if (n > 0) {
        hipLaunchKernelGGL(k_peak_flags, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, d_x, n, pre_max, post_max,
                           pre_avg, post_avg, delta, d_flags);
        OFP_LAUNCH_CHECK("k_peak_flags");
    }
    hipLaunchKernelGGL(k_peak_wait, dim3(1), dim3(64), 0, stream, (const uint8_t*)d_flags, n, wait, d_peaks, cap,
                       d_count);
    OFP_LAUNCH_CHECK("k_peak_wait");
    return OFP_OK;
}

}  // extern "C"
