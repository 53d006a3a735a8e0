// The whole FCNN (calibration.py:463-527: [Linear -> BatchNorm1d? -> act -> Dropout?] x k -> Linear,
// eval mode) as ONE device routine: a wave pushes a tile of 16 rows through every layer with the
// activations staying in LDS.  The arithmetic per output element is exactly that of k_dense
// (csrc/ofp_nn.hip): the same v_mfma_f32_16x16x4_f32 chain over k in steps of 4, then
// act((acc + bias) * scale + shift) -- so the fused forms (ofp_mlp_forward, the epilogue of
// ofp_stft_power_mel_mlp, the per-hop kernel of csrc/ofp_hop.hip) are bit-identical to the layer-by-
// layer ofp_dense chain.  An output element depends on its own row only, so tiles may be partly
// filled (rows that hold no frame are computed and dropped).
#pragma once
#include "ofp_common.h"

#define OFP_MLP_MAX_LAYERS 8

typedef float ofp_f32x4 __attribute__((ext_vector_type(4)));

struct MlpPlan {
    int n_layers;
    int dims[OFP_MLP_MAX_LAYERS + 1];  // dims[0] = input width, dims[n_layers] = output width
    int act[OFP_MLP_MAX_LAYERS];
    int w_off[OFP_MLP_MAX_LAYERS];     // float offsets into params; W is [out][in] row-major (torch Linear.weight)
    int b_off[OFP_MLP_MAX_LAYERS];     // -1: absent
    int sc_off[OFP_MLP_MAX_LAYERS];    // folded eval-mode BatchNorm1d scale / shift, -1: identity
    int sh_off[OFP_MLP_MAX_LAYERS];
    int n_params;                      // floats in params
    int st_a, st_b;                    // row strides (floats) of the two activation tiles: tile A holds the
                                       // inputs of the even layers, tile B those of the odd ones
    const float* params;               // device
};

// the opaque handle of include/onsetfp.h
struct ofp_mlp {
    MlpPlan plan;
    float* d_params = nullptr;
};

__device__ __forceinline__ float ofp_activate(float v, int act) {
    switch (act) {
        case OFP_ACT_RELU: return v > 0.0f ? v : 0.0f;
        case OFP_ACT_SILU: return v / (1.0f + expf(-v));
        case OFP_ACT_LEAKYRELU: return v >= 0.0f ? v : 0.01f * v;
        case OFP_ACT_ELU: return v > 0.0f ? v : expm1f(v);
        case OFP_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// LDS visibility inside one wave (the tile routine is executed by exactly one wave)
__device__ __forceinline__ void ofp_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One wave, all 64 lanes active.  ta: [16][st_a] inputs (row r at ta + r*st_a); tb: [16][st_b]
// scratch; prm: the packed parameters (LDS or global).  store(row, col, value) receives the
// network's outputs.  Fragment layout as k_dense: A lane l = x[l&15][k0 + (l>>4)],
// B lane l = W[col0 + (l&15)][k0 + (l>>4)], acc[r] = row (l>>4)*4 + r, column l&15.
template <class Store>
__device__ __forceinline__ void ofp_mlp_tile(const MlpPlan& p, const float* prm, float* ta, float* tb, int lane,
                                             Store&& store) {
    const int li = lane & 15, lk = lane >> 4;
    float* src = ta;
    float* dst = tb;
    int ss = p.st_a, ds = p.st_b;
#pragma unroll 1
    for (int L = 0; L < p.n_layers; ++L) {
        const int in = p.dims[L], out = p.dims[L + 1];
        const float* W = prm + p.w_off[L];
        const bool last = L == p.n_layers - 1;
        const int act = p.act[L];
#pragma unroll 1
        for (int ct = 0; ct * 16 < out; ++ct) {
            const int bcol = ct * 16 + li;
            const bool bok = bcol < out;
            ofp_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < in; k0 += 4) {
                const int k = k0 + lk;
                const float a = k < in ? src[li * ss + k] : 0.0f;
                const float b = (bok && k < in) ? W[bcol * in + k] : 0.0f;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
            }
            if (bok) {
                const float bias = p.b_off[L] >= 0 ? prm[p.b_off[L] + bcol] : 0.0f;
                const float sc = p.sc_off[L] >= 0 ? prm[p.sc_off[L] + bcol] : 1.0f;
                const float sh = p.sh_off[L] >= 0 ? prm[p.sh_off[L] + bcol] : 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = ofp_activate((acc[r] + bias) * sc + sh, act);
                    if (last) store(lk * 4 + r, bcol, v);
                    else dst[(lk * 4 + r) * ds + bcol] = v;
                }
            }
        }
        ofp_wave_lds_sync();
        float* t = src; src = dst; dst = t;
        const int ts = ss; ss = ds; ds = ts;
    }
}
