// Small per-onset post-processing callables of the reference, batched on the GPU (SURVEY.md 8f N3 and the
// surface next to rows a8 / N3):
//   filter_data          detection.py:355-370   null samples whose first difference has the wrong sign
//   detect_onset_region  detection.py:454-484   |x| -> median filter -> threshold -> binary opening -> first True
//   StretchFrameExtractor data.py:195-223       scipy.signal.resample of a window to frame_length (Fourier method)
// One workgroup per item; the data of an item lives in LDS.
#include <algorithm>

#include "ofp_common.h"

namespace {

using ofp::cdiv;

// ---- filter_data: y[t][c] = x[t][c] unless (x[t][c] - x[t-1][c]) has the nulled sign; row 0 is kept
// (np.diff(..., prepend=x[:1]) is 0 there).  direction 1 "up": null where diff < 0; 2 "down": diff > 0.
__global__ __launch_bounds__(256) void k_filter_direction(const float* __restrict__ x, int64_t n, int C, int direction,
                                                          float* __restrict__ y) {
    const int64_t total = n * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        const float d = i >= C ? v - x[i - C] : 0.0f;
        const bool kill = direction == 1 ? d < 0.0f : d > 0.0f;
        y[i] = kill ? 0.0f : v;
    }
}

// ---- detect_onset_region for a batch of onsets of one 1-D signal.  One wave per onset, region <= 4096.
constexpr int REG_MAX = 4096;
__global__ __launch_bounds__(64) void k_onset_region(const float* __restrict__ audio, int64_t n_audio,
                                                     const int64_t* __restrict__ onsets, int n_half, int med,
                                                     float factor, int64_t* __restrict__ out) {
    __shared__ float a[REG_MAX];       // |region|
    __shared__ float f[REG_MAX];       // median filtered
    __shared__ unsigned char b0[REG_MAX], b1[REG_MAX];
    const int lane = threadIdx.x;
    const int64_t on = onsets[blockIdx.x];
    const int64_t start = max<int64_t>(on - n_half, 0);
    const int64_t end = min<int64_t>(on + n_half, n_audio);
    const int len = (int)max<int64_t>(end - start, 0);
    if (len == 0) {  // np.argmax of an empty array raises in the reference; report the start
        if (lane == 0) out[blockIdx.x] = start;
        return;
    }
    for (int i = lane; i < len; i += 64) a[i] = fabsf(audio[start + i]);
    __syncthreads();
    // scipy.signal.medfilt: zero-padded window of `med` (odd) samples, the middle order statistic
    const int h = med / 2;
    float mx = 0.0f;
    for (int i = lane; i < len; i += 64) {
        float w[33];
        for (int k = 0; k < med; ++k) {
            const int j = i - h + k;
            w[k] = (j >= 0 && j < len) ? a[j] : 0.0f;
        }
        for (int p = 1; p < med; ++p) {  // insertion sort of <= 33 values
            const float v = w[p];
            int q = p - 1;
            while (q >= 0 && w[q] > v) {
                w[q + 1] = w[q];
                --q;
            }
            w[q + 1] = v;
        }
        f[i] = w[h];
        mx = fmaxf(mx, w[h]);
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    __syncthreads();
    const float thr = factor * mx;
    for (int i = lane; i < len; i += 64) b0[i] = f[i] > thr;
    __syncthreads();
    // binary_opening(structure = ones(5)): erosion (outside = 0) then dilation, centred structure
    for (int i = lane; i < len; i += 64) {
        bool e = true;
        for (int k = -2; k <= 2; ++k) {
            const int j = i + k;
            e = e && j >= 0 && j < len && b0[j];
        }
        b1[i] = e;
    }
    __syncthreads();
    int first = 0x7fffffff;
    for (int i = lane; i < len; i += 64) {
        bool d = false;
        for (int k = -2; k <= 2; ++k) {
            const int j = i + k;
            d = d || (j >= 0 && j < len && b1[j]);
        }
        if (d) first = min(first, i);
    }
    for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o));
    if (lane == 0) out[blockIdx.x] = start + (first == 0x7fffffff ? 0 : first);  // np.argmax of all-False is 0
}

// ---- scipy.signal.resample (real input, time domain, no window) of item (i, c):
// x = audio[start_i : start_i + nx_i, c] -> y[num]; the spectrum is taken and synthesised directly (DFT with
// tabulated fp64 twiddles: the lengths are arbitrary, a few hundred samples).  Samples outside the clip read 0.
struct ResampleArgs {
    const float* audio;     // [n_samples][C] (or C == 1)
    int64_t n_samples;
    int C;
    const int64_t* start;   // [n_items]
    const int32_t* nx;      // [n_items]
    int num;                // output length
    int max_nx;             // LDS was sized for inputs up to this length
    float* out;             // [n_items][C][num]
};

__global__ __launch_bounds__(256) void k_resample(ResampleArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int item = blockIdx.x / a.C, c = blockIdx.x % a.C;
    const int Nx = max(1, min(a.nx[item], a.max_nx)), num = a.num;  // (the host checks nx <= max_nx; never overrun the LDS)
    const int N = min(num, Nx), K = N / 2;  // bins 0..K are copied
    double2* twx = reinterpret_cast<double2*>(smem);             // e^{-2 pi i j / Nx}, j < Nx
    double2* twy = twx + Nx;                                     // e^{+2 pi i j / num}, j < num
    double2* Y = twy + num;                                      // [K + 1]
    float* xs = reinterpret_cast<float*>(Y + K + 1);             // [Nx]
    const int64_t st = a.start[item];
    for (int j = threadIdx.x; j < Nx; j += 256) {
        double s, co;
        sincospi(-2.0 * (double)j / (double)Nx, &s, &co);
        twx[j] = make_double2(co, s);
        const int64_t t = st + j;
        xs[j] = (t >= 0 && t < a.n_samples) ? a.audio[t * a.C + c] : 0.0f;
    }
    for (int j = threadIdx.x; j < num; j += 256) {
        double s, co;
        sincospi(2.0 * (double)j / (double)num, &s, &co);
        twy[j] = make_double2(co, s);
    }
    __syncthreads();
    for (int k = threadIdx.x; k <= K; k += 256) {  // rfft bins that survive
        double re = 0.0, im = 0.0;
        int ph = 0;  // k * n mod Nx
        for (int n = 0; n < Nx; ++n) {
            const double v = (double)xs[n];
            re += v * twx[ph].x;
            im += v * twx[ph].y;
            ph += k;
            if (ph >= Nx) ph -= Nx;
        }
        if (N % 2 == 0 && k == K) {  // the Nyquist component of the shorter length
            if (num < Nx) { re *= 2.0; im *= 2.0; }
            else if (Nx < num) { re *= 0.5; im *= 0.5; }
        }
        Y[k] = make_double2(re, im);
    }
    __syncthreads();
    const double scale = ((double)num / (double)Nx) / (double)num;  // irfft's 1/num, then y *= num / Nx
    float* dst = a.out + ((int64_t)item * a.C + c) * num;
    for (int m = threadIdx.x; m < num; m += 256) {
        double acc = Y[0].x;
        int ph = 0;
        for (int k = 1; k <= K; ++k) {
            ph += m;
            if (ph >= num) ph -= num;
            if (2 * k == num) acc += Y[k].x * twy[ph].x;  // irfft takes the last bin of an even length as real
            else acc += 2.0 * (Y[k].x * twy[ph].x - Y[k].y * twy[ph].y);
        }
        dst[m] = (float)(acc * scale);
    }
}

}  // namespace

extern "C" {

int ofp_filter_direction(const float* d_x, int64_t n, int32_t n_channels, int32_t direction, float* d_y, void* stream) {
    if (n == 0) return OFP_OK;
    OFP_REQUIRE(d_x && d_y && d_x != d_y && n > 0 && n_channels >= 1, "ofp_filter_direction: bad argument");
    OFP_REQUIRE(direction == 1 || direction == 2, "ofp_filter_direction: direction %d (1 up, 2 down)", direction);
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv(n * n_channels, 256), 256 * 16);
    hipLaunchKernelGGL(k_filter_direction, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_x, n, n_channels, direction, d_y);
    OFP_LAUNCH_CHECK("k_filter_direction");
    return OFP_OK;
}

int ofp_onset_region(const float* d_audio, int64_t n_audio, const int64_t* d_onsets, int64_t n_onsets, int32_t n,
                     int32_t median_filter_size, float threshold_factor, int64_t* d_out, void* stream) {
    if (n_onsets == 0) return OFP_OK;
    OFP_REQUIRE(d_audio && d_onsets && d_out && n_audio >= 0 && n_onsets > 0 && n_onsets < (1ll << 31),
                "ofp_onset_region: bad argument");
    OFP_REQUIRE(n >= 0 && 2 * (n / 2) <= REG_MAX, "ofp_onset_region: regions of up to %d samples (n = %d)", REG_MAX, n);
    OFP_REQUIRE(median_filter_size >= 1 && median_filter_size <= 33 && median_filter_size % 2 == 1,
                "ofp_onset_region: median_filter_size must be odd and <= 33 (got %d)", median_filter_size);
    hipLaunchKernelGGL(k_onset_region, dim3((unsigned)n_onsets), dim3(64), 0, (hipStream_t)stream, d_audio, n_audio,
                       d_onsets, n / 2, median_filter_size, threshold_factor, d_out);
    OFP_LAUNCH_CHECK("k_onset_region");
    return OFP_OK;
}

int ofp_resample_windows(const float* d_audio, int64_t n_samples, int32_t n_channels, const int64_t* d_start,
                         const int32_t* d_nx, int64_t n_items, int32_t max_nx, int32_t num, float* d_out, void* stream) {
    if (n_items == 0) return OFP_OK;
    OFP_REQUIRE(d_audio && d_start && d_nx && d_out && n_channels >= 1 && n_items > 0 && n_items * n_channels < (1ll << 31),
                "ofp_resample_windows: bad argument");
    OFP_REQUIRE(num >= 1 && max_nx >= 1 && num <= 2048 && max_nx <= 2048,
                "ofp_resample_windows: windows of up to 2048 samples (num = %d, longest input %d)", num, max_nx);
    ResampleArgs a{d_audio, n_samples, n_channels, d_start, d_nx, num, max_nx, d_out};
    const size_t lds = (size_t)(max_nx + num + std::min(num, max_nx) / 2 + 1) * 16 + (size_t)max_nx * 4 + 16;
    static ofp::LdsAttrCache attr;
    if (int rc = ofp::ensure_dynamic_lds(reinterpret_cast<const void*>(k_resample), lds, attr)) return rc;
    hipLaunchKernelGGL(k_resample, dim3((unsigned)(n_items * n_channels)), dim3(256), lds, (hipStream_t)stream, a);
    OFP_LAUNCH_CHECK("k_resample");
    return OFP_OK;
}

}  // extern "C"
