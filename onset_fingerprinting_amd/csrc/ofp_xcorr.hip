// Batched small cross-correlations and onset fixing on the device:
// cross_correlation_lag (reference detection.py:195-268), adjust_onset (:299-352) and
// fix_onsets (:373-451).
//
// Canon for the correlation values (the tests' CPU checker accumulates the same way): every
// lag's dot product is accumulated in fp64 over ascending i by ONE thread (fp32 x fp32 products
// are exact in fp64), rounded once to fp32 and divided by its contribution count in fp32.  Only
// the lags inside the searched window are computed, never the full 2n-1.
#include <algorithm>

#include "ofp_common.h"

namespace {

constexpr int XT = 256;         // threads per workgroup
constexpr int XW = 64;          // wavefront
constexpr int X_MAXN = 4096;    // longest sequence staged in LDS (2 x 16 KiB)
constexpr int X_MAXC = 128;     // most channels in one onset group
constexpr int X_MAXF = 15;      // largest median filter
constexpr int X_MAXD = 4;       // highest difference order

// Python's a[start:stop] on an array of `len`: negative indices wrap once, then clamp.
__device__ __forceinline__ void py_slice(int start, int stop, int len, int* lo, int* hi) {
    if (start < 0) start = max(start + len, 0);
    if (stop < 0) stop = max(stop + len, 0);
    start = min(start, len);
    stop = min(stop, len);
    *lo = start;
    *hi = max(stop, start);
}

// Entry j of the normalised full correlation of xs, ys (length n), detection.py:244-250.
__device__ __forceinline__ float cc_entry(const float* xs, const float* ys, int n, int cutoff, int j) {
    const int k = j - (n - 1);
    const int i0 = k < 0 ? -k : 0, i1 = k > 0 ? n - k : n;
    double acc = 0.0;
    for (int i = i0; i < i1; ++i) acc += (double)xs[i + k] * (double)ys[i];
    const int m = j < n ? j : 2 * n - 2 - j;
    const int cnt = m < cutoff ? cutoff : m + 1;
    return __fdiv_rn((float)acc, (float)cnt);
}

// First index of the maximum of cc[lo, hi) (np.argmax), relative to lo; -1 when the window is
// empty.  Whole workgroup; s_v/s_i hold one slot per wave.
__device__ int cc_argmax(const float* xs, const float* ys, int n, int cutoff, int lo, int hi, float* cc_out,
                         float* s_v, int* s_i) {
    float bv = -INFINITY;
    int bi = INT_MAX;
    for (int j = lo + (int)threadIdx.x; j < hi; j += XT) {
        float v = cc_entry(xs, ys, n, cutoff, j);
        if (cc_out) cc_out[j - lo] = v;
        if (v > bv || bi == INT_MAX) {  // ascending j per thread: strict > keeps the first
            bv = v;
            bi = j - lo;
        }
    }
    for (int o = XW / 2; o > 0; o >>= 1) {
        float ov = __shfl_xor(bv, o);
        int oi = __shfl_xor(bi, o);
        if (oi != INT_MAX && (bi == INT_MAX || ov > bv || (ov == bv && oi < bi))) {
            bv = ov;
            bi = oi;
        }
    }
    const int lane = threadIdx.x & (XW - 1), wave = threadIdx.x / XW;
    __syncthreads();
    if (lane == 0) {
        s_v[wave] = bv;
        s_i[wave] = bi;
    }
    __syncthreads();
    bv = s_v[0];
    bi = s_i[0];
    for (int w = 1; w < XT / XW; ++w) {
        float ov = s_v[w];
        int oi = s_i[w];
        if (oi != INT_MAX && (bi == INT_MAX || ov > bv || (ov == bv && oi < bi))) {
            bv = ov;
            bi = oi;
        }
    }
    return bi == INT_MAX ? -1 : bi;
}

// In-place d-th difference of s[0..n_in) (np.diff: repeated first differences, each rounded to
// fp32), optional abs.  Whole workgroup; afterwards s[0..n_in-d) is valid.
__device__ void diff_abs_inplace(float* s, int n_in, int d, int take_abs) {
    for (int r = 0; r < d; ++r) {
        const int m = n_in - r - 1;
        for (int t0 = 0; t0 < m; t0 += XT) {  // tile by tile: read both operands, sync, write
            int t = t0 + threadIdx.x;
            float v = t < m ? s[t + 1] - s[t] : 0.0f;
            __syncthreads();
            if (t < m) s[t] = v;
            __syncthreads();
        }
    }
    if (take_abs)
        for (int t = threadIdx.x; t < n_in - d; t += XT) s[t] = fabsf(s[t]);
    __syncthreads();
}

// One workgroup per pair.
__global__ __launch_bounds__(XT) void k_xcorr_lag(const float* __restrict__ x, const float* __restrict__ y, int n_in,
                                                  int d, int take_abs, int cutoff, const int32_t* __restrict__ lo,
                                                  const int32_t* __restrict__ hi, int32_t* __restrict__ argmax,
                                                  float* __restrict__ cc, int cc_stride) {
    __shared__ float xs[X_MAXN], ys[X_MAXN];
    __shared__ float s_v[XT / XW];
    __shared__ int s_i[XT / XW];
    const int64_t p = blockIdx.x;
    for (int t = threadIdx.x; t < n_in; t += XT) {
        xs[t] = x[p * n_in + t];
        ys[t] = y[p * n_in + t];
    }
    __syncthreads();
    diff_abs_inplace(xs, n_in, d, take_abs);
    diff_abs_inplace(ys, n_in, d, take_abs);
    const int n = n_in - d;
    int l = lo[p], h = hi[p];
    l = max(0, min(l, 2 * n - 1));
    h = max(l, min(h, 2 * n - 1));
    if (cc) h = min(h, l + cc_stride);
    int r = cc_argmax(xs, ys, n, cutoff, l, h, cc ? cc + p * cc_stride : nullptr, s_v, s_i);
    if (threadIdx.x == 0) argmax[p] = r;
}

// Full cross-correlation of row pairs, what the reference gets from a grouped F.conv1d
// (batch_cc, data.py:226-230; paired_xcorr, model.py:12-45): out[i][j] = sum_t a[i][t + j - (L-1)]
// * b[i][t], j in [0, 2L-1).  One workgroup per row pair, rows in LDS, one lag per thread, fp32
// fma over ascending t.  `mean_k` > 1 averages groups of mean_k consecutive rows (the mean over the
// K feature maps of paired_xcorr) into one output row.
__global__ __launch_bounds__(XT) void k_xcorr_full(const float* __restrict__ a, const float* __restrict__ b, int L,
                                                   int64_t a_stride, int64_t b_stride, int mean_k,
                                                   float* __restrict__ out) {
    __shared__ float as[X_MAXN], bs[X_MAXN];
    const int64_t row0 = (int64_t)blockIdx.x * mean_k;
    const int nl = 2 * L - 1;
    for (int j0 = 0; j0 < nl; j0 += XT) {  // accumulators for this thread's lags, over the mean_k rows
        const int j = j0 + threadIdx.x;
        float acc_mean = 0.0f;
        for (int m = 0; m < mean_k; ++m) {
            __syncthreads();
            for (int t = threadIdx.x; t < L; t += XT) {
                as[t] = a[(row0 + m) * a_stride + t];
                bs[t] = b[(row0 + m) * b_stride + t];
            }
            __syncthreads();
            if (j < nl) {
                const int k = j - (L - 1);
                const int t0 = k < 0 ? -k : 0, t1 = k > 0 ? L - k : L;
                float acc = 0.0f;
                for (int t = t0; t < t1; ++t) acc = fmaf(as[t + k], bs[t], acc);
                acc_mean += acc;
            }
        }
        if (j < nl) out[(int64_t)blockIdx.x * nl + j] = mean_k > 1 ? acc_mean / (float)mean_k : acc_mean;
    }
}

// adjust_onset (detection.py:299-352) for one pair, executed by one wave: which of the two onsets moves to
// make their lag `new_lag`, decided by the exponentially weighted signal between the old and the new
// position (weights np.exp(np.linspace(0, -e, |lag_diff|)), sums in fp64 as np.sum of the float64
// products, normalised by the signals' maxima mx / my).  *ca / *cb: what to add to onset 0 / onset 1
// (valid on lane 0).
__device__ __forceinline__ void adjust_onset_wave(const float* xs, const float* ys, int n, int o0, int o1, int new_lag,
                                                  float mx, float my, int lane, int64_t* ca, int64_t* cb) {
    const int lag_diff = (o1 - o0) - new_lag;
    const int k = lag_diff < 0 ? -lag_diff : lag_diff;
    int x_start, x_end, y_start, y_end;
    if (lag_diff < 0) {
        x_start = max(o0 + lag_diff, 0);
        x_end = min(o0, n);
        y_start = min(o1, n);
        y_end = min(o1 - lag_diff, n);
    } else {
        x_start = o0;
        x_end = min(o0 + lag_diff, n);
        y_start = max(o1 - lag_diff, 0);
        y_end = min(o1, n);
    }
    x_start = max(0, min(x_start, n));
    y_start = max(0, min(y_start, n));
    // weights np.exp(np.linspace(0, -e, k)): w[m] = exp(m * (-e / (k - 1))), w[k-1] = exp(-e)
    const double step = k > 1 ? -2.718281828459045 / (double)(k - 1) : 0.0;
    const int Lx = x_end - x_start, Ly = y_end - y_start;
    double sa = 0.0, sb = 0.0;
    for (int q = lane; q < Lx; q += XW) {
        int m = k - Lx + q;
        double w = exp(m == k - 1 && k > 1 ? -2.718281828459045 : (double)m * step);
        sa += (double)xs[x_start + q] * w;
    }
    for (int q = lane; q < Ly; q += XW) {
        int m = k - 1 - q;
        double w = exp(m == k - 1 && k > 1 ? -2.718281828459045 : (double)m * step);
        sb += (double)ys[y_start + q] * w;
    }
    for (int o = XW / 2; o > 0; o >>= 1) {
        sa += __shfl_xor(sa, o);
        sb += __shfl_xor(sb, o);
    }
    const double da = Lx > 0 ? sa / (double)mx : 0.0;
    const double db = Ly > 0 ? sb / (double)my : 0.0;
    if (da > db) {
        if (o0 + lag_diff < 0) {
            *ca = 0;
            *cb = -lag_diff;
        } else {
            *ca = lag_diff;
            *cb = 0;
        }
    } else {
        *ca = 0;
        *cb = -lag_diff;
    }
}

// adjust_onset for a batch of pairs, one wave per pair: x, y [P][n] rows, onsets [P][2], new_lag [P] ->
// moves [P][2] (what the reference returns: the amounts to add to the two onsets).
__global__ __launch_bounds__(XW) void k_adjust_onset(const float* __restrict__ x, const float* __restrict__ y, int n,
                                                     const int32_t* __restrict__ onsets,
                                                     const int32_t* __restrict__ new_lag, int32_t* __restrict__ moves) {
    const int64_t p = blockIdx.x;
    const int lane = threadIdx.x;
    const float* xs = x + p * n;
    const float* ys = y + p * n;
    float mx = -INFINITY, my = -INFINITY;
    for (int i = lane; i < n; i += XW) {
        mx = fmaxf(mx, xs[i]);
        my = fmaxf(my, ys[i]);
    }
    for (int o = XW / 2; o > 0; o >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, o));
        my = fmaxf(my, __shfl_xor(my, o));
    }
    int64_t ca = 0, cb = 0;
    adjust_onset_wave(xs, ys, n, onsets[2 * p], onsets[2 * p + 1], new_lag[p], mx, my, lane, &ca, &cb);
    if (lane == 0) {
        moves[2 * p] = (int32_t)ca;
        moves[2 * p + 1] = (int32_t)cb;
    }
}

// ---- fix_onsets ---------------------------------------------------------------------------
struct FixArgs {
    const float* audio;  // [n_clips][N][C]
    int64_t n_samples;
    int C;
    int64_t* onsets;          // [n_clips][cap_groups][C] in/out
    int64_t cap_groups;       // rows per clip
    const int64_t* n_groups;  // [n_clips] rows in use per clip, or NULL (all)
    int filter_size, d, direction, take_abs, zero_left, cutoff, tol, shift;
    int32_t* status;  // [G]
    float* sec;       // [G][C][max_section]
    int max_section;
};

__device__ __forceinline__ float median_at(const float* __restrict__ a, int C, int c, int64_t base, int n_sec, int t,
                                           int fs) {
    float w[X_MAXF];
    const int left = fs / 2;
    for (int k = 0; k < fs; ++k) {
        int u = t - left + k;  // scipy.ndimage 'reflect': (d c b a | a b c d | d c b a)
        while (u < 0 || u >= n_sec) u = u < 0 ? -u - 1 : 2 * n_sec - 1 - u;
        float v = a[(base + u) * C + c];
        int q = k;
        while (q > 0 && w[q - 1] > v) {
            w[q] = w[q - 1];
            --q;
        }
        w[q] = v;
    }
    return w[fs / 2];
}

// One workgroup per onset group; the pairs of a group are processed in order because every pair
// moves the anchor onset the next pair starts from (detection.py:431-450).
__global__ __launch_bounds__(XT) void k_fix_onsets(FixArgs A) {
    __shared__ float xs[X_MAXN], ys[X_MAXN];
    __shared__ float s_v[XT / XW];
    __shared__ int s_i[XT / XW];
    __shared__ int64_t og[X_MAXC];
    __shared__ int idx[X_MAXC];
    __shared__ int s_hdr[4];
    const int C = A.C;
    const int64_t g = blockIdx.x, clip = g / A.cap_groups;
    if (A.n_groups && g % A.cap_groups >= A.n_groups[clip]) {  // row not in use (uniform over the workgroup)
        if (threadIdx.x == 0) A.status[g] = 2;
        return;
    }
    const float* audio = A.audio + clip * A.n_samples * A.C;
    const int look = A.cutoff + A.tol;
    const int lane = threadIdx.x & (XW - 1), wave = threadIdx.x / XW;

    if (threadIdx.x == 0) {
        for (int c = 0; c < C; ++c) og[c] = A.onsets[g * C + c] + A.shift;
        for (int c = 0; c < C; ++c) {  // stable insertion argsort (np.argsort, detection.py:416)
            int q = c;
            while (q > 0 && og[idx[q - 1]] > og[c]) {
                idx[q] = idx[q - 1];
                --q;
            }
            idx[q] = c;
        }
        int64_t a = og[idx[0]], b = og[idx[C - 1]];
        int64_t n_sec = b - a + 2 * look;
        bool ok = a - look >= 0 && b + look <= A.n_samples && n_sec <= A.max_section && n_sec - A.d >= 1;
        s_hdr[0] = ok ? 1 : 0;
        s_hdr[1] = ok ? (int)n_sec : 0;
    }
    __syncthreads();
    const int64_t a = og[idx[0]];
    const int n_sec = s_hdr[1], n = n_sec - A.d;
    const int64_t base = a - look;
    if (!s_hdr[0]) {  // too close to the clip edge or longer than the work space: shift only
        for (int c = threadIdx.x; c < C; c += XT) A.onsets[g * C + c] = og[c];
        if (threadIdx.x == 0) A.status[g] = 1;
        return;
    }

    // section = diff(median_filter(audio[a-look : b+look], fs, axes=0), d), rectified (detection.py:419-428)
    float* sec = A.sec + g * (int64_t)C * A.max_section;
    for (int i = threadIdx.x; i < n * C; i += XT) {
        const int c = i % C, t = i / C;
        float v[X_MAXD + 1];
        for (int k = 0; k <= A.d; ++k) v[k] = median_at(audio, C, c, base, n_sec, t + k, A.filter_size);
        for (int r = 1; r <= A.d; ++r)
            for (int k = 0; k <= A.d - r; ++k) v[k] = v[k + 1] - v[k];
        float s = v[0];
        if (A.direction == 1 && s < 0.0f) s = 0.0f;
        if (A.direction == 2 && s > 0.0f) s = 0.0f;
        if (A.take_abs) s = fabsf(s);
        sec[(int64_t)c * A.max_section + t] = s;
    }
    __syncthreads();

    const int c0 = idx[0];
    for (int pi = 1; pi < C; ++pi) {
        const int c1 = idx[pi];
        const int o0 = (int)(og[c0] - base), o1 = (int)(og[c1] - base);
        float* gx = sec + (int64_t)c0 * A.max_section;
        float* gy = sec + (int64_t)c1 * A.max_section;
        int z0 = 0, z1 = 0, dummy;
        if (A.zero_left) {  // x[:o0] = 0, y[:o1] = 0 on the section itself (detection.py:435-437)
            py_slice(0, o0, n, &dummy, &z0);
            py_slice(0, o1, n, &dummy, &z1);
        }
        float mx = -INFINITY, my = -INFINITY;
        for (int t = threadIdx.x; t < n; t += XT) {
            float vx = gx[t], vy = gy[t];
            if (t < z0) gx[t] = vx = 0.0f;
            if (t < z1) gy[t] = vy = 0.0f;
            xs[t] = vx;
            ys[t] = vy;
            mx = fmaxf(mx, vx);
            my = fmaxf(my, vy);
        }
        for (int o = XW / 2; o > 0; o >>= 1) {
            mx = fmaxf(mx, __shfl_xor(mx, o));
            my = fmaxf(my, __shfl_xor(my, o));
        }
        __syncthreads();
        if (lane == 0) {
            s_v[wave] = mx;
            ((float*)s_i)[wave] = my;
        }
        __syncthreads();
        mx = s_v[0];
        my = ((float*)s_i)[0];
        for (int w = 1; w < XT / XW; ++w) {
            mx = fmaxf(mx, s_v[w]);
            my = fmaxf(my, ((float*)s_i)[w]);
        }
        // cross_correlation_lag(x, y, o, cutoff, tol) (detection.py:259-268)
        const int current_lag = o1 - o0;
        int lo, hi;
        py_slice(n - current_lag - A.tol, n - current_lag + A.tol, 2 * n - 1, &lo, &hi);
        const int am = cc_argmax(xs, ys, n, A.cutoff, lo, hi, nullptr, s_v, s_i);
        if (am >= 0) {
            const int new_lag = current_lag + A.tol - am;
            // adjust_onset (detection.py:310-352)
            if (wave == 0) {
                int64_t ca, cb;
                adjust_onset_wave(xs, ys, n, o0, o1, new_lag, mx, my, lane, &ca, &cb);
                if (lane == 0) {
                    og[c0] += ca;
                    og[c1] += cb;
                }
            }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < C; c += XT) A.onsets[g * C + c] = og[c];
    if (threadIdx.x == 0) A.status[g] = 0;
}

}  // namespace

extern "C" {

int ofp_adjust_onset(const float* d_x, const float* d_y, int64_t n_pairs, int32_t n, const int32_t* d_onsets,
                     const int32_t* d_new_lag, int32_t* d_moves, void* stream) {
    if (n_pairs == 0) return OFP_OK;
    OFP_REQUIRE(d_x && d_y && d_onsets && d_new_lag && d_moves, "ofp_adjust_onset: NULL argument");
    OFP_REQUIRE(n_pairs > 0 && n_pairs < (1ll << 31) && n >= 1, "ofp_adjust_onset: bad size");
    hipLaunchKernelGGL(k_adjust_onset, dim3((unsigned)n_pairs), dim3(XW), 0, (hipStream_t)stream, d_x, d_y, n, d_onsets,
                       d_new_lag, d_moves);
    OFP_LAUNCH_CHECK("k_adjust_onset");
    return OFP_OK;
}

int ofp_xcorr_lag(const float* d_x, const float* d_y, int64_t n_pairs, int32_t n_in, int32_t d, int32_t take_abs,
                  int32_t cutoff, const int32_t* d_lo, const int32_t* d_hi, int32_t* d_argmax, float* d_cc,
                  int32_t cc_stride, void* stream) {
    if (n_pairs == 0) return OFP_OK;
    OFP_REQUIRE(d_x && d_y && d_lo && d_hi && d_argmax, "ofp_xcorr_lag: NULL argument");
    OFP_REQUIRE(n_pairs > 0 && n_pairs < (1ll << 31) && d >= 0 && d <= X_MAXD && n_in - d >= 1 && n_in <= X_MAXN &&
                    cutoff >= 0 && (!d_cc || cc_stride >= 1),
                "ofp_xcorr_lag: bad size (n_pairs=%lld n=%d d=%d; sequences up to %d samples, d up to %d)",
                (long long)n_pairs, n_in, d, X_MAXN, X_MAXD);
    hipLaunchKernelGGL(k_xcorr_lag, dim3((unsigned)n_pairs), dim3(XT), 0, (hipStream_t)stream, d_x, d_y, n_in, d,
                       take_abs, cutoff, d_lo, d_hi, d_argmax, d_cc, cc_stride);
    OFP_LAUNCH_CHECK("k_xcorr_lag");
    return OFP_OK;
}

int ofp_xcorr_full(const float* d_a, const float* d_b, int64_t n_rows, int32_t length, int64_t a_stride,
                   int64_t b_stride, int32_t mean_k, float* d_out, void* stream) {
    if (n_rows == 0) return OFP_OK;
    OFP_REQUIRE(d_a && d_b && d_out && length >= 1 && length <= X_MAXN && mean_k >= 1 && n_rows % mean_k == 0 &&
                    n_rows / mean_k < (1ll << 31),
                "ofp_xcorr_full: bad argument (rows=%lld length=%d mean_k=%d; rows of up to %d samples)",
                (long long)n_rows, length, mean_k, X_MAXN);
    hipLaunchKernelGGL(k_xcorr_full, dim3((unsigned)(n_rows / mean_k)), dim3(XT), 0, (hipStream_t)stream, d_a, d_b,
                       length, a_stride, b_stride, mean_k, d_out);
    OFP_LAUNCH_CHECK("k_xcorr_full");
    return OFP_OK;
}

int64_t ofp_fix_onsets_workspace_bytes(int64_t n_groups, int32_t n_channels, int32_t max_section) {
    if (n_groups < 0 || n_channels < 1 || max_section < 1) return -1;
    return n_groups * n_channels * (int64_t)max_section * 4;
}

int ofp_fix_onsets(const float* d_audio, int64_t n_clips, int64_t n_samples, int32_t n_channels, int64_t* d_onsets,
                   int64_t cap_groups, const int64_t* d_n_groups, int32_t filter_size, int32_t d, int32_t direction, int32_t take_abs, int32_t zero_left,
                   int32_t cutoff, int32_t tol, int32_t shift, int32_t max_section, int32_t* d_status, void* d_ws,
                   int64_t ws_bytes, void* stream) {
    const int64_t n_groups = n_clips * cap_groups;
    if (n_groups == 0) return OFP_OK;
    OFP_REQUIRE(d_audio && d_onsets && d_status && d_ws, "ofp_fix_onsets: NULL argument");
    OFP_REQUIRE(n_clips > 0 && cap_groups > 0 && n_groups < (1ll << 31) && n_channels >= 1 && n_channels <= X_MAXC,
                "ofp_fix_onsets: %d channels per group (1..%d supported)", n_channels, X_MAXC);
    OFP_REQUIRE(filter_size >= 1 && filter_size <= X_MAXF && d >= 0 && d <= X_MAXD && direction >= 0 &&
                    direction <= 2 && cutoff >= 0 && tol >= 0,
                "ofp_fix_onsets: filter_size %d (1..%d), d %d (0..%d), direction %d (0 none, 1 up, 2 down)",
                filter_size, X_MAXF, d, X_MAXD, direction);
    OFP_REQUIRE(max_section >= 1 && max_section <= X_MAXN, "ofp_fix_onsets: max_section %d outside 1..%d",
                max_section, X_MAXN);
    OFP_REQUIRE(ws_bytes >= ofp_fix_onsets_workspace_bytes(n_groups, n_channels, max_section),
                "ofp_fix_onsets: work space too small");
    FixArgs A{d_audio, n_samples, n_channels, d_onsets, cap_groups, d_n_groups, filter_size, d, direction, take_abs, zero_left, cutoff, tol,
              shift, d_status, (float*)d_ws, max_section};
    hipLaunchKernelGGL(k_fix_onsets, dim3((unsigned)n_groups), dim3(XT), 0, (hipStream_t)stream, A);
    OFP_LAUNCH_CHECK("k_fix_onsets");
    return OFP_OK;
}

}  // extern "C"
