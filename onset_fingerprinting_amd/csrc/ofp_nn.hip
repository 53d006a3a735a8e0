// Classifier forward passes on gfx950: fused dense layer on the fp32 matrix
// cores (v_mfma_f32_16x16x4_f32: exact fp32, k-ordered fma chain) and a direct
// Conv1d + activation.  Reference: calibration.py:463-527 (FCNN, eval mode,
// BatchNorm1d folded into scale/shift by the host), model.py:52-120 (CNN).
#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include "ofp_common.h"
#include "ofp_mlp.h"

namespace {

using ofp::cdiv;
typedef ofp_f32x4 f32x4;

__device__ __forceinline__ float activate(float v, int act) { return ofp_activate(v, act); }

// One wave computes a 16-row x 16-column output tile per MFMA chain.
// A fragment: lane l holds x[row0 + (l&15)][k0 + (l>>4)]
// B fragment: lane l holds W[col0 + (l&15)][k0 + (l>>4)]   (B[k][j] = W[j][k])
// C/D: acc[r] is row (l>>4)*4 + r, column l&15.
__global__ __launch_bounds__(256) void k_dense(const float* __restrict__ x, int64_t n, int in, int out,
                                               const float* __restrict__ w, const float* __restrict__ b,
                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                               int act, float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t row_tiles = cdiv(n, 16);
    const int col_tiles = (out + 15) / 16;
    const int li = lane & 15, lk = lane >> 4;
    for (int64_t rt = wave; rt < row_tiles; rt += n_waves) {
        const int64_t arow = rt * 16 + li;
        const float* xr = x + arow * in;
        const bool arow_ok = arow < n;
        for (int ct = 0; ct < col_tiles; ++ct) {
            const int bcol = ct * 16 + li;
            const float* wr = w + (int64_t)bcol * in;
            const bool bcol_ok = bcol < out;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < in; k0 += 4) {
                const int k = k0 + lk;
                float a = (arow_ok && k < in) ? xr[k] : 0.0f;
                float bb = (bcol_ok && k < in) ? wr[k] : 0.0f;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb, acc, 0, 0, 0);
            }
            const int col = ct * 16 + li;
            if (col < out) {
                const float bias = b ? b[col] : 0.0f;
                const float sc = scale ? scale[col] : 1.0f;
                const float sh = shift ? shift[col] : 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = rt * 16 + lk * 4 + r;
                    if (row < n) y[row * out + col] = activate((acc[r] + bias) * sc + sh, act);
                }
            }
        }
    }
}

// The whole network in one launch (ofp_mlp.h): the parameters are copied to LDS once per workgroup,
// every wave takes 16-row tiles (grid stride), loads the rows with coalesced reads into its tile A and
// runs all layers there.  LDS: params, then per wave tile A [16][st_a] and tile B [16][st_b].
constexpr int MLP_WAVES = 4;
__global__ __launch_bounds__(64 * MLP_WAVES) void k_mlp(MlpPlan p, const float* __restrict__ x, int64_t n,
                                                        float* __restrict__ y) {
    extern __shared__ __align__(16) float sm[];
    float* prm = sm;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* ta = sm + ((p.n_params + 3) & ~3) + (size_t)w * 16 * (p.st_a + p.st_b);
    float* tb = ta + 16 * p.st_a;
    for (int i = threadIdx.x; i < p.n_params; i += blockDim.x) prm[i] = p.params[i];
    __syncthreads();
    const int in = p.dims[0], out = p.dims[p.n_layers];
    const int64_t row_tiles = cdiv(n, 16);
    for (int64_t rt = (int64_t)blockIdx.x * MLP_WAVES + w; rt < row_tiles; rt += (int64_t)gridDim.x * MLP_WAVES) {
        const int64_t r0 = rt * 16;
        const int nr = (int)min<int64_t>(16, n - r0);
        const float* src = x + r0 * in;
        for (int i = lane; i < 16 * in; i += 64) {
            const int r = i / in, k = i - r * in;
            ta[r * p.st_a + k] = r < nr ? src[i] : 0.0f;
        }
        ofp_wave_lds_sync();
        ofp_mlp_tile(p, prm, ta, tb, lane, [&](int row, int col, float v) {
            if (row < nr) y[(r0 + row) * out + col] = v;
        });
    }
}

// Conv1d (any stride, any groups) + bias + activation, then the optional per-channel affine of an
// eval-mode BatchNorm1d and MaxPool1d(2, 2) -- the layer of model.py:91-107 in one pass.
// Thread per output element (after pooling).
__global__ __launch_bounds__(256) void k_conv1d(const float* __restrict__ x, int64_t n, int cin, int w,
                                                const float* __restrict__ wt, const float* __restrict__ b,
                                                int cout, int k, int padding, int dilation, int groups, int stride,
                                                int act, const float* __restrict__ bn_scale,
                                                const float* __restrict__ bn_shift, int pool, int wout,
                                                float* __restrict__ y) {
    const int64_t total = n * cout * wout;
    const int cin_g = cin / groups, cout_g = cout / groups;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % wout);
        const int64_t t = i / wout;
        const int o = (int)(t % cout);
        const int64_t s = t / cout;
        const float* xs = x + (s * cin + (int64_t)(o / cout_g) * cin_g) * w;
        const float* ws = wt + (int64_t)o * cin_g * k;
        float best = 0.0f;
        for (int h = 0; h <= pool; ++h) {
            const int pc = (pool ? 2 * p + h : p) * stride;  // first tap, in input samples, before padding
            float acc = b ? b[o] : 0.0f;
            for (int ci = 0; ci < cin_g; ++ci) {
                for (int kk = 0; kk < k; ++kk) {
                    int q = pc - padding + kk * dilation;
                    if (q >= 0 && q < w) acc = fmaf(xs[(int64_t)ci * w + q], ws[ci * k + kk], acc);
                }
            }
            float v = activate(acc, act);
            if (bn_scale) v = fmaf(v, bn_scale[o], bn_shift[o]);
            best = h == 0 ? v : fmaxf(best, v);
        }
        y[i] = best;
    }
}

// nn.GroupNorm(1, K) (what CCCNN's batch_norm=True builds, model.py:497-501): every item [K][V] is
// normalised by the mean and (biased) variance of all its K*V values, then scaled and shifted per
// channel.  One workgroup per item; sums in fp64.  Optional MaxPool1d(2, 2) on the way out.
__global__ __launch_bounds__(256) void k_groupnorm1(const float* __restrict__ x, int K, int V,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float eps, int pool, float* __restrict__ y) {
    __shared__ double red[2][4];
    __shared__ float stat[2];
    const int64_t item = blockIdx.x;
    const float* src = x + item * (int64_t)K * V;
    const int n = K * V;
    double s = 0.0, ss = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double v = (double)src[i];
        s += v;
        ss += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        ss += __shfl_xor(ss, o);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s;
        red[1][threadIdx.x >> 6] = ss;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double ts = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        const double tss = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const double mean = ts / n;
        const double var = fmax(tss / n - mean * mean, 0.0);
        stat[0] = (float)mean;
        stat[1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const float mean = stat[0], rstd = stat[1];
    const int Vo = pool ? V / 2 : V;
    float* dst = y + item * (int64_t)K * Vo;
    for (int i = threadIdx.x; i < K * Vo; i += 256) {
        const int k = i / Vo, p = i - k * Vo;
        const float g = gamma ? gamma[k] : 1.0f, b = beta ? beta[k] : 0.0f;
        if (pool) {
            const float a0 = (src[k * V + 2 * p] - mean) * rstd * g + b;
            const float a1 = (src[k * V + 2 * p + 1] - mean) * rstd * g + b;
            dst[i] = fmaxf(a0, a1);
        } else {
            dst[i] = (src[k * V + p] - mean) * rstd * g + b;
        }
    }
}

// CCCNN correlation head (model.py:524-534): for each item (b, c) the K feature maps [K][V] are
// auto-correlated (full, 2V-1 lags: cc[j] = sum_i f[i + j - (V-1)] * f[i]), summed over the maps
// and soft-maxed over the lags.  One workgroup per item; the maps live in LDS.
__global__ __launch_bounds__(256) void k_autocorr_softmax(const float* __restrict__ x, int K, int V,
                                                          float* __restrict__ out) {
    extern __shared__ float sm[];  // [K*V] maps, then [2V-1] cc, then 64 scratch
    const int64_t item = blockIdx.x;
    const int L = 2 * V - 1;
    float* f = sm;
    float* cc = sm + (size_t)K * V;
    float* red = cc + L;
    const float* src = x + item * (int64_t)K * V;
    for (int i = threadIdx.x; i < K * V; i += blockDim.x) f[i] = src[i];
    __syncthreads();
    for (int j = threadIdx.x; j < L; j += blockDim.x) {
        const int sh = j - (V - 1);  // lag
        const int lo = sh < 0 ? -sh : 0, hi = sh > 0 ? V - sh : V;
        float acc = 0.0f;
        for (int k = 0; k < K; ++k) {
            const float* fk = f + k * V;
            for (int i = lo; i < hi; ++i) acc = fmaf(fk[i + sh], fk[i], acc);
        }
        cc[j] = acc;
    }
    __syncthreads();
    // softmax over the L lags
    float m = -INFINITY;
    for (int j = threadIdx.x; j < L; j += blockDim.x) m = fmaxf(m, cc[j]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float ssum = 0.0f;
    for (int j = threadIdx.x; j < L; j += blockDim.x) {
        const float e = expf(cc[j] - m);
        cc[j] = e;
        ssum += e;
    }
    for (int o = 32; o > 0; o >>= 1) ssum += __shfl_xor(ssum, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ssum;
    __syncthreads();
    ssum = red[0] + red[1] + red[2] + red[3];
    float* dst = out + item * (int64_t)L;
    for (int j = threadIdx.x; j < L; j += blockDim.x) dst[j] = cc[j] / ssum;
}

}  // namespace

extern "C" {

int ofp_autocorr_softmax(const float* d_x, int64_t n, int32_t K, int32_t V, float* d_out, void* stream) {
    if (n == 0) return OFP_OK;
    OFP_REQUIRE(d_x && d_out && K >= 1 && V >= 1, "ofp_autocorr_softmax: bad argument");
    const size_t lds = ((size_t)K * V + 2 * V + 64) * sizeof(float);
    OFP_REQUIRE(lds <= 160 * 1024, "ofp_autocorr_softmax: K*V = %d floats do not fit the LDS", K * V);
    if (lds > 65536)
        OFP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_autocorr_softmax),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_autocorr_softmax, dim3((unsigned)n), dim3(256), lds, (hipStream_t)stream, d_x, K, V, d_out);
    OFP_LAUNCH_CHECK("k_autocorr_softmax");
    return OFP_OK;
}


int ofp_dense(const float* d_x, int64_t n, int32_t in, int32_t out, const float* d_w, const float* d_b,
              const float* d_scale, const float* d_shift, int32_t act, float* d_y, void* stream) {
    if (n == 0) return OFP_OK;
    OFP_REQUIRE(d_x && d_w && d_y, "ofp_dense: NULL argument");
    OFP_REQUIRE(in >= 1 && out >= 1 && n > 0, "ofp_dense: bad sizes");
    OFP_REQUIRE(act >= OFP_ACT_IDENTITY && act <= OFP_ACT_TANH, "ofp_dense: unknown activation %d", act);
    int64_t row_tiles = cdiv(n, 16);
    unsigned grid = (unsigned)std::min<int64_t>(cdiv(row_tiles, 4), 256 * 8);
    hipLaunchKernelGGL(k_dense, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_x, n, in, out, d_w, d_b,
                       d_scale, d_shift, act, d_y);
    OFP_LAUNCH_CHECK("k_dense");
    return OFP_OK;
}

// ---- whole-network handle -----------------------------------------------------------------
int ofp_mlp_create(int32_t n_layers, const int32_t* dims, const int32_t* act, const float* const* h_w,
                   const float* const* h_b, const float* const* h_scale, const float* const* h_shift,
                   ofp_mlp** out) {
    OFP_REQUIRE(dims && act && h_w && out, "ofp_mlp_create: NULL argument");
    OFP_REQUIRE(n_layers >= 1 && n_layers <= OFP_MLP_MAX_LAYERS, "ofp_mlp_create: %d layers (1..%d supported)", n_layers,
                OFP_MLP_MAX_LAYERS);
    MlpPlan p;
    std::memset(&p, 0, sizeof(p));
    p.n_layers = n_layers;
    std::vector<float> host;
    int wa = 1, wb = 1;
    for (int L = 0; L <= n_layers; ++L) {
        OFP_REQUIRE(dims[L] >= 1 && dims[L] <= 4096, "ofp_mlp_create: layer width %d out of range", dims[L]);
        p.dims[L] = dims[L];
        if (L < n_layers) (L % 2 == 0 ? wa : wb) = std::max(L % 2 == 0 ? wa : wb, dims[L]);
    }
    auto put = [&](const float* src, int count) {
        const int off = (int)host.size();
        host.insert(host.end(), src, src + count);
        while (host.size() % 4) host.push_back(0.0f);
        return off;
    };
    for (int L = 0; L < n_layers; ++L) {
        OFP_REQUIRE(h_w[L], "ofp_mlp_create: layer %d has no weight", L);
        OFP_REQUIRE(act[L] >= OFP_ACT_IDENTITY && act[L] <= OFP_ACT_TANH, "ofp_mlp_create: unknown activation %d", act[L]);
        OFP_REQUIRE((!h_scale || !h_scale[L]) == (!h_shift || !h_shift[L]), "ofp_mlp_create: give both scale and shift");
        p.act[L] = act[L];
        p.w_off[L] = put(h_w[L], dims[L] * dims[L + 1]);
        p.b_off[L] = (h_b && h_b[L]) ? put(h_b[L], dims[L + 1]) : -1;
        p.sc_off[L] = (h_scale && h_scale[L]) ? put(h_scale[L], dims[L + 1]) : -1;
        p.sh_off[L] = (h_shift && h_shift[L]) ? put(h_shift[L], dims[L + 1]) : -1;
    }
    p.n_params = (int)host.size();
    p.st_a = wa | 1;  // odd strides: the 16 rows of a fragment read fall into different LDS banks
    p.st_b = wb | 1;
    ofp_mlp* m = new (std::nothrow) ofp_mlp();
    if (!m) return ofp::fail(OFP_ERR_INVALID, "out of host memory");
    hipError_t e = hipMalloc(&m->d_params, host.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(m->d_params, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (m->d_params) (void)hipFree(m->d_params);
        delete m;
        return ofp::fail(OFP_ERR_HIP, "ofp_mlp_create: %s", hipGetErrorString(e));
    }
    p.params = m->d_params;
    m->plan = p;
    *out = m;
    return OFP_OK;
}

int ofp_mlp_destroy(ofp_mlp* m) {
    if (!m) return OFP_OK;
    if (m->d_params) (void)hipFree(m->d_params);
    delete m;
    return OFP_OK;
}

int64_t ofp_mlp_lds_bytes(const ofp_mlp* m) {
    if (!m) return -1;
    const MlpPlan& p = m->plan;
    return ((int64_t)((p.n_params + 3) & ~3) + (int64_t)MLP_WAVES * 16 * (p.st_a + p.st_b)) * 4;
}

int ofp_mlp_forward(const ofp_mlp* m, const float* d_x, int64_t n, float* d_y, void* stream) {
    OFP_REQUIRE(m, "ofp_mlp_forward: NULL handle");
    if (n == 0) return OFP_OK;
    OFP_REQUIRE(d_x && d_y && n > 0, "ofp_mlp_forward: NULL argument");
    const size_t lds = (size_t)ofp_mlp_lds_bytes(m);
    OFP_REQUIRE(lds <= 160 * 1024, "ofp_mlp_forward: the network (%d parameters) does not fit the LDS; run it layer by "
                "layer with ofp_dense", m->plan.n_params);
    static ofp::LdsAttrCache attr;
    if (int rc = ofp::ensure_dynamic_lds(reinterpret_cast<const void*>(k_mlp), lds, attr)) return rc;
    const int64_t row_tiles = cdiv(n, 16);
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv(row_tiles, MLP_WAVES), 256 * 8);
    hipLaunchKernelGGL(k_mlp, dim3(grid), dim3(64 * MLP_WAVES), lds, (hipStream_t)stream, m->plan, d_x, n, d_y);
    OFP_LAUNCH_CHECK("k_mlp");
    return OFP_OK;
}

int ofp_conv1d(const float* d_x, int64_t n, int32_t cin, int32_t w, const float* d_w, const float* d_b,
               int32_t cout, int32_t k, int32_t padding, int32_t dilation, int32_t groups, int32_t stride,
               int32_t act, const float* d_bn_scale, const float* d_bn_shift, int32_t pool, float* d_y,
               void* stream) {
    if (n == 0) return OFP_OK;
    OFP_REQUIRE(d_x && d_w && d_y, "ofp_conv1d: NULL argument");
    OFP_REQUIRE(groups >= 1 && cin % groups == 0 && cout % groups == 0,
                "ofp_conv1d: groups %d must divide cin %d and cout %d", groups, cin, cout);
    OFP_REQUIRE((d_bn_scale == nullptr) == (d_bn_shift == nullptr), "ofp_conv1d: give both bn_scale and bn_shift");
    OFP_REQUIRE(stride >= 1, "ofp_conv1d: stride %d", stride);
    int wout = w + 2 * padding - dilation * (k - 1);
    OFP_REQUIRE(wout >= 1, "ofp_conv1d: empty output (w=%d k=%d padding=%d dilation=%d)", w, k, padding, dilation);
    wout = (wout - 1) / stride + 1;
    if (pool) wout /= 2;  // MaxPool1d(kernel_size=2, stride=2), floor
    OFP_REQUIRE(wout >= 1, "ofp_conv1d: nothing left after pooling");
    OFP_REQUIRE(act >= OFP_ACT_IDENTITY && act <= OFP_ACT_TANH, "ofp_conv1d: unknown activation %d", act);
    int64_t total = n * cout * wout;
    unsigned grid = (unsigned)std::min<int64_t>(cdiv(total, 256), 256 * 16);
    hipLaunchKernelGGL(k_conv1d, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_x, n, cin, w, d_w, d_b, cout,
                       k, padding, dilation, groups, stride, act, d_bn_scale, d_bn_shift, pool ? 1 : 0, wout, d_y);
    OFP_LAUNCH_CHECK("k_conv1d");
    return OFP_OK;
}

int ofp_groupnorm1(const float* d_x, int64_t n, int32_t K, int32_t V, const float* d_gamma, const float* d_beta,
                   float eps, int32_t pool, float* d_y, void* stream) {
    if (n == 0) return OFP_OK;
    OFP_REQUIRE(d_x && d_y && d_x != d_y && K >= 1 && V >= 1 && n < (1ll << 31), "ofp_groupnorm1: bad argument");
    OFP_REQUIRE(!pool || V >= 2, "ofp_groupnorm1: nothing left after pooling");
    hipLaunchKernelGGL(k_groupnorm1, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, d_x, K, V, d_gamma, d_beta,
                       eps, pool ? 1 : 0, d_y);
    OFP_LAUNCH_CHECK("k_groupnorm1");
    return OFP_OK;
}

}  // extern "C"
