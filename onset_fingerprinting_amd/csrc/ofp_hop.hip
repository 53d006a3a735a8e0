// Per-hop streaming session (BASELINE config 5): the reference's realtime call pattern
//
//   PortAudio callback (realtime/audio.py:81-120): ring-buffer write (:97) -> AmplitudeOnsetDetector
//   on the hop (:62-74, detection.py:727-798) -> classifier (multilateration.py:555-557 ->
//   calibration.py:552-560), and the per-hop spectral frame of the trailing n_fft samples
//   (realtime/recording.py:273-280)
//
// as ONE captured hipGraph per hop on a device-resident ring buffer:
//
//   H2D of the hop (B x C floats, pinned)  ->  k_hop_begin (hop counter, onset count = 0)
//   ->  k_stream (the detector, csrc/ofp_stream.hip: state in HBM)
//   ->  k_hop_spectral (one workgroup per channel: ring write, Hann x trailing n_fft samples,
//       rFFT, |X|^2, mel bands, the whole FCNN -- nothing but mel + logits leaves the chip)
//   ->  D2H of one packed result block {count, records, logits, mel, rel}.
//
// Fused form (default whenever the detector fits one workgroup): the five nodes above collapse into ONE
// kernel node, k_hop_fused -- workgroup 0 runs the detector while workgroups 1..C run the spectral
// branch, the hop is read from and the result block written to pinned host memory directly.
//
// A replay needs no argument update: the write cursor is a counter in device memory.  The frame the
// spectral kernel computes for the hop that ends at sample e is bit-identical to frame (e - n_fft)/B
// of the dense kernel (k_stft_power, hop = B) on the same stream: same tables, same FFT, same mel
// and classifier epilogue.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "ofp_detector.h"
#include "ofp_fft.h"
#include "ofp_mlp.h"
#include "ofp_stream_dev.h"

using namespace ofpfft;

namespace {

// Per-hop onset strength of the channel mean (realtime/recording.py:273-311, RecAnalysis.fft +
// onset_strength): symmetric float32 Hann x mean over channels of audio[-n_fft:], rFFT, |X|^2 in dB floored 80 dB
// below a tracked maximum, positive flux against the previous frame averaged over the bins, normalised by a
// tracked min / max, moving max / mean over the last frames.  The two trackers are loopmate.EMA_MinMaxTracker
// objects -- loopmate is not available, their arithmetic is ASSUMED to be that of envelope_follower.c:27-57
// with one alpha (PARITY UNPINNED).
struct StrengthArgs {
    int enabled, n_ring, max_length, avg_length;
    float ls_alpha, ls_minmax, oe_alpha, oe_minmin;
    const float* wsym;  // [F] scipy.signal.windows.hann(F) as float32
    float* prev;        // [F/2+1] power spectrum of the previous hop's frame
    float* st;          // [4] ls_max, oe_min, oe_max
    float* ring;        // [n_ring] normalised onset envelope, one entry per hop
    float* out;         // [4 + tg_len] raw flux, normalised, moving max, moving mean, then the tempogram (result block)
    int tg_len;         // tempogram window (realtime/recording.py:313-327), 0: off
    const float* tgw;   // [tg_len] scipy.signal.windows.hann(tg_len) as float32
};

struct HopArgs {
    StrengthArgs sg;
    int C, B, n_mels, nnz, mean_mode;
    int64_t R;             // rows of the ring buffer
    int64_t* ctl;          // [0] hops pushed so far (incremented by k_hop_begin), [1] spare
    const float* hop;      // [B][C] the hop just uploaded
    float* ring;           // [R][C]
    const float2* twM;     // precomputed once per session by k_hop_tables (the same values
    const float2* twF;     // k_stft_power builds in LDS)
    const float* win;
    const int32_t *flo, *flen, *foff;
    const float* fw;
    MlpPlan plan;          // n_layers == 0: no classifier
    float* logits;         // [C][n_out]
    float* mel;            // [C][n_mels]
    int64_t* count;        // onset count of the hop (zeroed by k_hop_begin)
    int64_t* hop_index;    // result header: index of the hop this block belongs to
    volatile int64_t* done_flag;  // fused form: hops completed, written to pinned host memory by the workgroup that
                                  // finishes last, after every result of the hop (the host polls it)
};

__global__ void k_hop_begin(HopArgs a) {
    if (threadIdx.x == 0) {
        const int64_t h = a.ctl[0];
        a.ctl[0] = h + 1;
        *a.count = 0;
        *a.hop_index = h;
    }
}

template <int F>
__global__ void k_hop_tables(float2* twM, float2* twF, float* win, float* wsym) {
    build_tables<F>(twM, twF, nullptr, F);
    // the periodic Hann exactly as k_stft_power holds it: F/2 + 1 computed entries, mirrored above
    for (int n = threadIdx.x; n <= F / 2; n += blockDim.x) win[n] = (float)(0.5 - 0.5 * cospi(2.0 * (double)n / (double)F));
    __syncthreads();
    for (int n = F / 2 + 1 + threadIdx.x; n < F; n += blockDim.x) win[n] = win[F - n];
    for (int n = threadIdx.x; n < F; n += blockDim.x)  // symmetric Hann (recording.py:249), fp64 then float32
        wsym[n] = (float)(0.5 - 0.5 * cospi(2.0 * (double)n / (double)(F - 1)));
}

template <int F>
struct HopCfg {
    static constexpr int M = F / 2;
    static constexpr int T = Cfg<F>::T;
    static constexpr int WGS = T < 64 ? 64 : T;
};

// One channel's share of a hop: ring write, Hann x trailing n_fft samples, rFFT, |X|^2, mel bands and the
// classifier.  Executed by one workgroup of WGS lanes (WGS == T when a frame needs more than one wave:
// the FFT passes then synchronise with workgroup barriers); h = hops pushed including this one.
template <int F, int WGS>
__device__ __forceinline__ void hop_spectral_body(const HopArgs& a, int c, int64_t h, unsigned char* smem) {
    constexpr int M = HopCfg<F>::M, T = HopCfg<F>::T;
    static_assert(T <= 64 ? WGS >= 64 : WGS == T, "lanes of a multi-wave frame must be the whole workgroup");
    float2* twM = reinterpret_cast<float2*>(smem);
    float2* twF = twM + M;
    float* win = reinterpret_cast<float*>(twF + M + 2);
    float2* A = reinterpret_cast<float2*>(win + F);
    float* fw = reinterpret_cast<float*>(A + FftBuf<M>::words);
    int32_t* flo = reinterpret_cast<int32_t*>(fw + a.nnz);
    int32_t* flen = flo + a.n_mels;
    int32_t* foff = flen + a.n_mels;
    MelSegs* segs = reinterpret_cast<MelSegs*>((reinterpret_cast<uintptr_t>(foff + a.n_mels) + 15) & ~(uintptr_t)15);
    float* partial = reinterpret_cast<float*>(segs + 1);
    float* mprm = partial + MEL_MAXSEG;
    float* tileA = mprm + ((a.plan.n_params + 3) & ~3);
    float* tileB = tileA + 16 * a.plan.st_a;
    float* hcol = tileB + 16 * a.plan.st_b;  // [B] this channel's column of the hop
    const int C = a.C, B = a.B;
    const int tid = threadIdx.x;
    // the hop may live in pinned host memory: every sample is fetched exactly once
    for (int t = tid; t < B; t += WGS) hcol[t] = a.hop[(int64_t)t * C + c];
    // tables, filterbank and classifier parameters: L2-resident copies -> LDS
    for (int k = tid; k < M; k += WGS) twM[k] = a.twM[k];
    for (int k = tid; k <= M; k += WGS) twF[k] = a.twF[k];
    for (int n = tid; n < F; n += WGS) win[n] = a.win[n];
    for (int i = tid; i < a.nnz; i += WGS) fw[i] = a.fw[i];
    for (int i = tid; i < a.n_mels; i += WGS) {
        flo[i] = a.flo[i];
        flen[i] = a.flen[i];
        foff[i] = a.foff[i];
    }
    for (int i = tid; i < a.plan.n_params; i += WGS) mprm[i] = a.plan.params[i];
    if (a.plan.n_layers > 0)
        for (int i = tid; i < 16 * a.plan.st_a; i += WGS) tileA[i] = 0.0f;
    if (tid == 0) mel_build_segs(segs, a.flen, a.n_mels);
    const int64_t first = (h - 1) * B;  // stream index of this hop's first sample
    __syncthreads();
    // ring-buffer write of this channel's column (realtime/audio.py:97)
    for (int t = tid; t < B; t += WGS) a.ring[((first + t) % a.R) * C + c] = hcol[t];
    // trailing n_fft samples of the stream (realtime/recording.py:276: audio[-n_fft:]); samples before
    // the stream started read as the zeros the ring was created with
    const int64_t base = h * B - F;
    auto sample = [&](int64_t s) -> float {
        if (s < 0) return 0.0f;
        if (s >= first) return hcol[s - first];
        return a.ring[(s % a.R) * C + c];
    };
    for (int p = tid; p < M; p += WGS)
        A[fft_pad<M>(p)] = make_float2(sample(base + 2 * p) * win[2 * p], sample(base + 2 * p + 1) * win[2 * p + 1]);
    __syncthreads();
    if (tid < T) {
        cfft<M, T>(A, twM, tid);
        // the same paired power bins, LDS staging and band sums as k_stft_power (bit-identical to its frame)
        constexpr int NQ = (M / 2) / T + 1;
        float pa[NQ], pb[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pp = tid + q * T;
            pa[q] = pb[q] = 0.0f;
            if (pp <= M / 2) rfft_power_pair<M>(A, twF, pp, pa[q], pb[q]);
        }
        frame_sync<T>();  // every bin of the spectrum has been read
        float* pf = reinterpret_cast<float*>(A);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pp = tid + q * T;
            if (pp <= M / 2) {
                pf[pp] = pa[q];
                if (pp < M / 2) pf[M - pp] = pb[q];
            }
        }
        frame_sync<T>();
        mel_bands(segs, pf, fw, flo, flen, foff, a.n_mels, tid, T, partial, [] { frame_sync<T>(); },
                  [&](int b, float acc) {
                      a.mel[c * a.n_mels + b] = acc;
                      if (a.plan.n_layers > 0) tileA[b] = acc;  // row 0 of the classifier's tile
                  });
    }
    if (a.plan.n_layers > 0) {
        __syncthreads();
        if (tid < 64) {
            const int nout = a.plan.dims[a.plan.n_layers];
            ofp_mlp_tile(a.plan, mprm, tileA, tileB, tid, [&](int r, int col, float v) {
                if (r == 0) a.logits[c * nout + col] = v;
            });
        }
    }
}

// One workgroup: the channel-mean frame of the hop that ends at sample h*B and its onset strength.
template <int F, int WGS>
__device__ __forceinline__ void hop_strength_body(const HopArgs& a, int64_t h, unsigned char* smem) {
    constexpr int M = HopCfg<F>::M, T = HopCfg<F>::T;
    static_assert(T <= 64 ? WGS >= 64 : WGS == T, "lanes of a multi-wave frame must be the whole workgroup");
    const StrengthArgs& g = a.sg;
    float2* twM = reinterpret_cast<float2*>(smem);
    float2* twF = twM + M;
    float2* A = twF + M + 2;
    float* hbuf = reinterpret_cast<float*>(A + FftBuf<M>::words);   // [B][C] the hop
    float* red = hbuf + (size_t)a.B * a.C;            // [WGS / 64 + 2]
    const int C = a.C, B = a.B, tid = threadIdx.x;
    for (int k = tid; k < M; k += WGS) twM[k] = a.twM[k];
    for (int k = tid; k <= M; k += WGS) twF[k] = a.twF[k];
    for (int i = tid; i < B * C; i += WGS) hbuf[i] = a.hop[i];
    __syncthreads();
    const int64_t first = (h - 1) * B, base = h * B - F;
    auto mean_sample = [&](int64_t s) -> float {  // audio[-n_fft:].mean(-1): float32 sum in channel order, then / C
        if (s < 0) return 0.0f;
        float m = 0.0f;
        if (s >= first) {
            for (int c = 0; c < C; ++c) m += hbuf[(s - first) * C + c];
        } else {
            const float* r = a.ring + (s % a.R) * C;
            for (int c = 0; c < C; ++c) m += r[c];
        }
        return m / (float)C;
    };
    for (int p = tid; p < M; p += WGS)
        A[fft_pad<M>(p)] = make_float2(g.wsym[2 * p] * mean_sample(base + 2 * p), g.wsym[2 * p + 1] * mean_sample(base + 2 * p + 1));
    __syncthreads();
    constexpr int NK = M / T + 1;
    float pw[NK], sdb[NK];
    float smax = -INFINITY;
    if (tid < T) {
        cfft<M, T>(A, twM, tid);
#pragma unroll
        for (int q = 0; q < NK; ++q) {
            const int k = tid + q * T;
            pw[q] = 0.0f;
            sdb[q] = -INFINITY;
            if (k <= M) {
                const float2 X = rfft_bin<M>(A, twF, k);
                pw[q] = X.x * X.x + X.y * X.y;
                sdb[q] = 10.0f * log10f(fmaxf(1e-10f, pw[q]));   // :290
                smax = fmaxf(smax, sdb[q]);
            }
        }
    }
    // block maximum of the dB spectrum -> the tracked maximum (:291), floor 80 dB below it (:292-294)
    for (int o = 32; o > 0; o >>= 1) smax = fmaxf(smax, __shfl_xor(smax, o));
    if ((tid & 63) == 0) red[tid >> 6] = smax;
    __syncthreads();
    if (tid == 0) {
        float m = red[0];
        for (int w = 1; w < (WGS + 63) / 64; ++w) m = fmaxf(m, red[w]);
        float ls = g.st[0];
        ls = m > ls ? m : (1.0f - g.ls_alpha) * ls + g.ls_alpha * m;
        ls = fmaxf(ls, g.ls_minmax);
        g.st[0] = ls;
        red[WGS / 64 + 1] = ls - 80.0f;
    }
    __syncthreads();
    const float floor_db = red[WGS / 64 + 1];
    float fsum = 0.0f;
    if (tid < T) {
#pragma unroll
        for (int q = 0; q < NK; ++q) {
            const int k = tid + q * T;
            if (k <= M) {
                const float s1 = fmaxf(sdb[q], floor_db);
                const float s0 = fmaxf(10.0f * log10f(fmaxf(1e-10f, g.prev[k])), floor_db);
                fsum += fmaxf(0.0f, s1 - s0);                       // :296
                g.prev[k] = pw[q];
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) fsum += __shfl_xor(fsum, o);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = fsum;
    __syncthreads();
    if (tid == 0) {
        float sum = 0.0f;
        for (int w = 0; w < (WGS + 63) / 64; ++w) sum += red[w];
        const float oe = sum / (float)(M + 1);
        float mn = g.st[1], mx = g.st[2];
        mx = oe > mx ? oe : (1.0f - g.oe_alpha) * mx + g.oe_alpha * oe;   // :299 add_sample
        mn = oe < mn ? oe : (1.0f - g.oe_alpha) * mn + g.oe_alpha * oe;
        mn = fmaxf(mn, g.oe_minmin);
        g.st[1] = mn;
        g.st[2] = mx;
        const float norm = (oe - mn) / (mx - mn);                          // :300-302 normalize_sample
        g.ring[(h - 1) % g.n_ring] = norm;
        g.out[0] = oe;
        g.out[1] = norm;
    }
    __syncthreads();
    // moving max / mean over the last MAX_LENGTH / AVG_LENGTH entries (:304-311; entries before the stream
    // started are the zeros the ring was created with)
    float vmax = -INFINITY, vsum = 0.0f;
    for (int i = tid; i < max(g.max_length, g.avg_length); i += WGS) {
        const int64_t e = h - 1 - i;
        const float v = e >= 0 ? g.ring[e % g.n_ring] : 0.0f;
        if (i < g.max_length) vmax = fmaxf(vmax, v);
        if (i < g.avg_length) vsum += v;
    }
    for (int o = 32; o > 0; o >>= 1) {
        vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        vsum += __shfl_xor(vsum, o);
    }
    if ((tid & 63) == 0) {
        red[tid >> 6] = vmax;
        hbuf[tid >> 6] = vsum;
    }
    __syncthreads();
    if (tid == 0) {
        float m = red[0], sm = hbuf[0];
        for (int w = 1; w < (WGS + 63) / 64; ++w) {
            m = fmaxf(m, red[w]);
            sm += hbuf[w];
        }
        g.out[2] = m;
        g.out[3] = sm / (float)g.avg_length;
    }
    // Tempogram frame (realtime/recording.py:313-327): irfft(|rfft(w * onset_env[-W:], n = 2W - 1)|^2)[:W] -- the
    // transform length 2W - 1 makes the circular correlation the LINEAR autocorrelation of the windowed envelope, so the
    // W lags are computed as what they are, tg[k] = sum_i y[i] y[i + k] (float32 fma chain over ascending i; the
    // reference's float FFT round trip agrees to its own rounding), then tg / (max(tg) + 1e-10).  Lags k and W-1-k go
    // to the same thread: W + 1 products each.  PARITY UNPINNED like the envelope it reads.
    if (g.tg_len > 0) {
        const int W = g.tg_len;
        float* y = red + 64;   // [W] windowed envelope, then [W] the raw lags (LDS reserved by ofp_hop_create)
        float* tg = y + W;
        for (int i = tid; i < W; i += WGS) {
            const int64_t e = h - W + i;   // onset_env[-W + i]; entries before the stream started are the ring's zeros
            y[i] = e >= 0 ? g.tgw[i] * g.ring[e % g.n_ring] : 0.0f;
        }
        __syncthreads();
        float mx = -INFINITY;
        for (int k = tid; 2 * k < W; k += WGS) {
            const int k2 = W - 1 - k;
            float a0 = 0.0f, a1 = 0.0f;
            for (int i = 0; i + k < W; ++i) a0 = fmaf(y[i], y[i + k], a0);
            if (k2 != k)
                for (int i = 0; i + k2 < W; ++i) a1 = fmaf(y[i], y[i + k2], a1);
            tg[k] = a0;
            mx = fmaxf(mx, a0);
            if (k2 != k) {
                tg[k2] = a1;
                mx = fmaxf(mx, a1);
            }
        }
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = mx;
        __syncthreads();
        float m = red[0];
        for (int w = 1; w < (WGS + 63) / 64; ++w) m = fmaxf(m, red[w]);
        const float den = m + 1e-10f;
        for (int k = tid; k < W; k += WGS) g.out[4 + k] = tg[k] / den;
    }
}

template <int F>
__global__ __launch_bounds__(HopCfg<F>::WGS) void k_hop_strength(HopArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    hop_strength_body<F, HopCfg<F>::WGS>(a, a.ctl[0], smem);
}

template <int F>
__global__ __launch_bounds__(HopCfg<F>::WGS) void k_hop_spectral(HopArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    hop_spectral_body<F, HopCfg<F>::WGS>(a, blockIdx.x, a.ctl[0], smem);
}

// The whole hop in ONE launch: workgroup 0 is the detector (the phase-split block step of
// ofp_stream_dev.h), workgroups 1..C the spectral branch of one channel each -- the two do not depend
// on each other, so the hop takes max(detector, spectral) instead of their sum plus three launch gaps.
// The hop is read straight from the pinned host buffer and the result block written straight to pinned
// host memory (no copy nodes).  The hop counter is advanced by the workgroup that finishes last.
template <int F>
struct FusedCfg {
    static constexpr int WGS = HopCfg<F>::T <= 64 ? 256 : HopCfg<F>::T;
};

template <int F>
__global__ __launch_bounds__(FusedCfg<F>::WGS) void k_hop_fused(HopArgs a, ofpstream::StreamArgs sa) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int64_t done = a.ctl[0];  // hops completed before this one
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) *a.hop_index = done;
        ofpstream::stream_par_blocks(sa, reinterpret_cast<float*>(smem));
    } else if ((int)blockIdx.x <= a.C) {
        hop_spectral_body<F, FusedCfg<F>::WGS>(a, (int)blockIdx.x - 1, done + 1, smem);
    } else {  // the onset-strength workgroup (only launched when enabled)
        hop_strength_body<F, FusedCfg<F>::WGS>(a, done + 1, smem);
    }
    // EVERY thread drains its own stores to the result block (pinned host memory) and the ring before the barrier:
    // the barrier does not wait for outstanding stores, and a fence by thread 0 alone covers only its own wave, so
    // the completion word could otherwise become visible while another wave's `rel` / `mel` rows are still in flight
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();  // (orders the ticket after the barrier's view of the other waves' fences)
        const unsigned long long t = atomicAdd(reinterpret_cast<unsigned long long*>(a.ctl + 1), 1ull);
        if (t == gridDim.x - 1) {  // every workgroup has read ctl[0] and published its results
            a.ctl[1] = 0;
            a.ctl[0] = done + 1;
            __threadfence_system();
            *a.done_flag = done + 1;
        }
    }
}

}  // namespace

struct ofp_hop_session {
    ofp_detector* det = nullptr;
    int C = 0, B = 0, n_fft = 0, n_mels = 0, n_out = 0, want_rel = 0;
    int64_t R = 0;
    hipStream_t stream = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    void* d_state = nullptr;
    float* d_hop = nullptr;
    float* d_ring = nullptr;
    int64_t* d_ctl = nullptr;
    float2* d_twM = nullptr;
    float2* d_twF = nullptr;
    float* d_win = nullptr;
    float* d_wsym = nullptr;
    float* d_tgw = nullptr;   // tempogram window
    float* d_sg = nullptr;      // onset strength: prev power [bins] | st [4] | ring [n_ring]
    float sg_init[4] = {10.0f, 0.0f, 1.0f, 0.0f};  // ls_max0, oe_min0, oe_max0
    size_t sg_floats = 0;
    size_t lds_strength = 0;
    int32_t* d_fb_i = nullptr;  // lo | len | off
    float* d_fb_w = nullptr;
    float* d_prm = nullptr;     // own copy of the classifier's parameters
    unsigned char* d_res = nullptr;
    float* h_hop = nullptr;           // pinned
    unsigned char* h_res = nullptr;   // pinned
    // result block layout (bytes)
    int64_t o_count = 0, o_index = 8, o_done = 16, o_rec = 24, o_logits = 0, o_mel = 0, o_rel = 0, o_sg = 0, res_bytes = 0;
    HopArgs args;
    ofpstream::StreamArgs sargs;  // fused form: the detector workgroup's arguments
    bool fused = false;
    size_t lds_fused = 0;
    size_t lds = 0;
    int64_t pushed = 0;   // hops submitted
    bool in_flight = false;
    bool retired = true;   // the last hop's kernel is known to have left the stream (a polled completion is not that)
};

namespace {

template <int F>
int hop_tables(ofp_hop_session* s) {
    hipLaunchKernelGGL(k_hop_tables<F>, dim3(1), dim3(256), 0, s->stream, s->d_twM, s->d_twF, s->d_win, s->d_wsym);
    OFP_LAUNCH_CHECK("k_hop_tables");
    return OFP_OK;
}

template <int F>
int hop_spectral(ofp_hop_session* s) {
    if (s->lds > 65536)
        OFP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hop_spectral<F>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds));
    hipLaunchKernelGGL(k_hop_spectral<F>, dim3((unsigned)s->C), dim3(HopCfg<F>::WGS), s->lds, s->stream, s->args);
    OFP_LAUNCH_CHECK("k_hop_spectral");
    return OFP_OK;
}

template <int F>
int hop_strength(ofp_hop_session* s) {
    if (s->lds_strength > 65536)
        OFP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hop_strength<F>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_strength));
    hipLaunchKernelGGL(k_hop_strength<F>, dim3(1), dim3(HopCfg<F>::WGS), s->lds_strength, s->stream, s->args);
    OFP_LAUNCH_CHECK("k_hop_strength");
    return OFP_OK;
}

int dispatch_strength(ofp_hop_session* s) {
    switch (s->n_fft) {
        case 256: return hop_strength<256>(s);
        case 512: return hop_strength<512>(s);
        case 1024: return hop_strength<1024>(s);
        case 2048: return hop_strength<2048>(s);
        case 4096: return hop_strength<4096>(s);
    }
    return ofp::fail(OFP_ERR_INVALID, "n_fft %d not supported (256,512,1024,2048,4096)", s->n_fft);
}

template <int F>
int hop_fused(ofp_hop_session* s) {
    if (s->lds_fused > 65536 - 20480)  // (the detector's static LDS comes on top)
        OFP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hop_fused<F>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_fused));
    hipLaunchKernelGGL(k_hop_fused<F>, dim3((unsigned)s->C + 1 + (s->args.sg.enabled ? 1 : 0)), dim3(FusedCfg<F>::WGS), s->lds_fused, s->stream, s->args,
                       s->sargs);
    OFP_LAUNCH_CHECK("k_hop_fused");
    return OFP_OK;
}

int fused_threads(int n_fft) {
    switch (n_fft) {
        case 2048: return FusedCfg<2048>::WGS;
        case 4096: return FusedCfg<4096>::WGS;
        default: return 256;
    }
}

int dispatch_fused(ofp_hop_session* s) {
    switch (s->n_fft) {
        case 256: return hop_fused<256>(s);
        case 512: return hop_fused<512>(s);
        case 1024: return hop_fused<1024>(s);
        case 2048: return hop_fused<2048>(s);
        case 4096: return hop_fused<4096>(s);
    }
    return ofp::fail(OFP_ERR_INVALID, "n_fft %d not supported (256,512,1024,2048,4096)", s->n_fft);
}

int dispatch_tables(ofp_hop_session* s) {
    switch (s->n_fft) {
        case 256: return hop_tables<256>(s);
        case 512: return hop_tables<512>(s);
        case 1024: return hop_tables<1024>(s);
        case 2048: return hop_tables<2048>(s);
        case 4096: return hop_tables<4096>(s);
    }
    return ofp::fail(OFP_ERR_INVALID, "n_fft %d not supported (256,512,1024,2048,4096)", s->n_fft);
}

int dispatch_spectral(ofp_hop_session* s) {
    switch (s->n_fft) {
        case 256: return hop_spectral<256>(s);
        case 512: return hop_spectral<512>(s);
        case 1024: return hop_spectral<1024>(s);
        case 2048: return hop_spectral<2048>(s);
        case 4096: return hop_spectral<4096>(s);
    }
    return ofp::fail(OFP_ERR_INVALID, "n_fft %d not supported (256,512,1024,2048,4096)", s->n_fft);
}

// the per-hop sequence, enqueued on the session's stream (captured once, then replayed)
int enqueue_hop(ofp_hop_session* s) {
    if (s->fused) return dispatch_fused(s);  // one kernel: no copy nodes, no begin kernel
    OFP_HIP(hipMemcpyAsync(s->d_hop, s->h_hop, (size_t)s->B * s->C * sizeof(float), hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(k_hop_begin, dim3(1), dim3(64), 0, s->stream, s->args);
    OFP_LAUNCH_CHECK("k_hop_begin");
    int rc = ofp_stream_process(s->det, s->d_state, s->d_hop, 1, 0, 0, 0,
                                s->want_rel ? reinterpret_cast<float*>(s->d_res + s->o_rel) : nullptr,
                                reinterpret_cast<ofp_onset*>(s->d_res + s->o_rec), s->C,
                                reinterpret_cast<int64_t*>(s->d_res + s->o_count), s->stream);
    if (rc != OFP_OK) return rc;
    if (s->args.sg.enabled) {  // (before the spectral node: it reads ring rows the spectral node is about to overwrite
                               //  only for hops older than the ring, never the current one)
        rc = dispatch_strength(s);
        if (rc != OFP_OK) return rc;
    }
    rc = dispatch_spectral(s);
    if (rc != OFP_OK) return rc;
    OFP_HIP(hipMemcpyAsync(s->h_res, s->d_res, (size_t)s->res_bytes, hipMemcpyDeviceToHost, s->stream));
    return OFP_OK;
}

int reset_state(ofp_hop_session* s) {
    int rc = ofp_stream_state_init(s->det, s->d_state, s->stream);
    if (rc != OFP_OK) return rc;
    OFP_HIP(hipMemsetAsync(s->d_ring, 0, (size_t)s->R * s->C * sizeof(float), s->stream));
    OFP_HIP(hipMemsetAsync(s->d_ctl, 0, 2 * sizeof(int64_t), s->stream));
    OFP_HIP(hipMemsetAsync(s->d_res, 0, (size_t)s->res_bytes, s->stream));
    if (s->d_sg) {
        OFP_HIP(hipMemsetAsync(s->d_sg, 0, s->sg_floats * 4, s->stream));
        OFP_HIP(hipMemcpyAsync(s->args.sg.st, s->sg_init, 16, hipMemcpyHostToDevice, s->stream));
    }
    OFP_HIP(hipStreamSynchronize(s->stream));
    std::memset(s->h_res, 0, (size_t)s->res_bytes);
    s->pushed = 0;
    s->in_flight = false;
    return OFP_OK;
}

}  // namespace

extern "C" {

int ofp_hop_destroy(ofp_hop_session* s) {
    if (!s) return OFP_OK;
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->exec) (void)hipGraphExecDestroy(s->exec);
    if (s->graph) (void)hipGraphDestroy(s->graph);
    void* dev[] = {s->d_state, s->d_hop, s->d_ring, s->d_ctl, s->d_twM, s->d_twF, s->d_win, s->d_wsym, s->d_tgw, s->d_sg, s->d_fb_i, s->d_fb_w,
                   s->d_prm, s->d_res};
    for (void* p : dev)
        if (p) (void)hipFree(p);
    if (s->h_hop) (void)hipHostFree(s->h_hop);
    if (s->h_res) (void)hipHostFree(s->h_res);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
    return OFP_OK;
}

int ofp_hop_create(ofp_detector* det, const ofp_hop_config* cfg, ofp_hop_session** out) {
    OFP_REQUIRE(det && cfg && out, "ofp_hop_create: NULL argument");
    const int C = det->p.n_channels, B = det->p.block_size;
    OFP_REQUIRE(C <= 1024, "ofp_hop_create: at most 1024 channels (got %d)", C);
    OFP_REQUIRE(cfg->n_fft == 256 || cfg->n_fft == 512 || cfg->n_fft == 1024 || cfg->n_fft == 2048 || cfg->n_fft == 4096,
                "n_fft %d not supported (256,512,1024,2048,4096)", cfg->n_fft);
    OFP_REQUIRE(cfg->ring_samples >= cfg->n_fft && cfg->ring_samples >= B,
                "ofp_hop_create: the ring buffer (%lld rows) must hold n_fft = %d and one hop = %d samples",
                (long long)cfg->ring_samples, cfg->n_fft, B);
    OFP_REQUIRE(cfg->n_mels >= 1 && cfg->fb_lo && cfg->fb_len && cfg->fb_off && cfg->fb_w && cfg->fb_nnz >= 1,
                "ofp_hop_create: NULL / empty filterbank");
    OFP_REQUIRE(cfg->fb_nnz <= 4 * (cfg->n_fft / 2 + 1), "ofp_hop_create: filterbank with %d weights for %d bins",
                cfg->fb_nnz, cfg->n_fft / 2 + 1);
    OFP_REQUIRE(cfg->n_mels <= 127 && cfg->fb_nnz / MEL_SEG + cfg->n_mels <= MEL_MAXSEG,
                "ofp_hop_create: at most 127 bands and %d 32-tap segments", MEL_MAXSEG);
    OFP_REQUIRE(!cfg->strength || (cfg->strength_ring >= 1 && cfg->max_length >= 1 && cfg->avg_length >= 1 &&
                                   cfg->max_length <= cfg->strength_ring && cfg->avg_length <= cfg->strength_ring),
                "ofp_hop_create: onset strength needs 1 <= max_length, avg_length <= strength_ring");
    OFP_REQUIRE(cfg->tg_win_length >= 0 && cfg->tg_win_length <= 4096 && (cfg->tg_win_length == 0 || (cfg->strength && cfg->tg_win_length <= cfg->strength_ring && cfg->tg_win_length >= 2)),
                "ofp_hop_create: the tempogram needs the onset strength and 2 <= tg_win_length <= min(strength_ring, 4096)");
    OFP_REQUIRE(!cfg->mlp || cfg->mlp->plan.dims[0] == cfg->n_mels,
                "ofp_hop_create: the classifier takes %d inputs, the filterbank has %d bands",
                cfg->mlp ? cfg->mlp->plan.dims[0] : 0, cfg->n_mels);
    ofp_hop_session* s = new (std::nothrow) ofp_hop_session();
    if (!s) return ofp::fail(OFP_ERR_INVALID, "out of host memory");
    s->det = det;
    s->C = C;
    s->B = B;
    s->n_fft = cfg->n_fft;
    s->n_mels = cfg->n_mels;
    s->R = cfg->ring_samples;
    s->want_rel = cfg->want_rel ? 1 : 0;
    MlpPlan plan;
    std::memset(&plan, 0, sizeof(plan));
    if (cfg->mlp) plan = cfg->mlp->plan;
    s->n_out = cfg->mlp ? plan.dims[plan.n_layers] : 0;
    auto up8 = [](int64_t v) { return (v + 7) / 8 * 8; };
    s->o_logits = up8(s->o_rec + (int64_t)C * (int64_t)sizeof(ofp_onset));
    s->o_mel = up8(s->o_logits + (int64_t)C * s->n_out * 4);
    s->o_rel = up8(s->o_mel + (int64_t)C * s->n_mels * 4);
    s->o_sg = up8(s->o_rel + (s->want_rel ? (int64_t)B * C * 4 : 0));
    s->res_bytes = up8(s->o_sg + 16 + (cfg->strength ? 4 * (int64_t)cfg->tg_win_length : 0));
    const int M = s->n_fft / 2;
    const int st_a = cfg->mlp ? plan.st_a : 0, st_b = cfg->mlp ? plan.st_b : 0;
    s->lds = (size_t)(M + M + 2) * 8 + (size_t)s->n_fft * 4 + (size_t)(M <= 512 ? M + M / 16 : M) * 8 + (size_t)cfg->fb_nnz * 4 +
             (size_t)3 * s->n_mels * 4 + 16 + sizeof(MelSegs) + (size_t)MEL_MAXSEG * 4 + (size_t)((plan.n_params + 3) & ~3) * 4 + (size_t)16 * (st_a + st_b) * 4 + (size_t)B * 4 + 64;
    int rc = OFP_OK;
    auto fail = [&](int code) {
        ofp_hop_destroy(s);
        return code;
    };
    if (s->lds > 160 * 1024) {
        ofp_hop_destroy(s);
        return ofp::fail(OFP_ERR_INVALID, "ofp_hop_create: %zu bytes of LDS needed, 160 KiB available", s->lds);
    }
#define HOP_TRY(call)                                                                                           \
    do {                                                                                                        \
        hipError_t e__ = (call);                                                                                \
        if (e__ != hipSuccess) {                                                                                \
            ofp_hop_destroy(s);                                                                                 \
            return ofp::fail(OFP_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e__));                      \
        }                                                                                                       \
    } while (0)
    HOP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    const int64_t sb = ofp_stream_state_bytes(det);
    HOP_TRY(hipMalloc(&s->d_state, (size_t)sb));
    HOP_TRY(hipMalloc(&s->d_hop, (size_t)B * C * 4));
    HOP_TRY(hipMalloc(&s->d_ring, (size_t)s->R * C * 4));
    HOP_TRY(hipMalloc(&s->d_ctl, 16));
    HOP_TRY(hipMalloc(&s->d_twM, (size_t)M * 8));
    HOP_TRY(hipMalloc(&s->d_twF, (size_t)(M + 2) * 8));
    HOP_TRY(hipMalloc(&s->d_win, (size_t)s->n_fft * 4));
    HOP_TRY(hipMalloc(&s->d_wsym, (size_t)s->n_fft * 4));
    HOP_TRY(hipMalloc(&s->d_fb_i, (size_t)3 * s->n_mels * 4));
    HOP_TRY(hipMalloc(&s->d_fb_w, (size_t)cfg->fb_nnz * 4));
    HOP_TRY(hipMalloc(&s->d_res, (size_t)s->res_bytes));
    HOP_TRY(hipHostMalloc((void**)&s->h_hop, (size_t)B * C * 4, hipHostMallocMapped));
    HOP_TRY(hipHostMalloc((void**)&s->h_res, (size_t)s->res_bytes, hipHostMallocMapped));
    std::memset(s->h_res, 0, (size_t)s->res_bytes);
    std::memset(s->h_hop, 0, (size_t)B * C * 4);
    HOP_TRY(hipMemcpy(s->d_fb_i, cfg->fb_lo, (size_t)s->n_mels * 4, hipMemcpyHostToDevice));
    HOP_TRY(hipMemcpy(s->d_fb_i + s->n_mels, cfg->fb_len, (size_t)s->n_mels * 4, hipMemcpyHostToDevice));
    HOP_TRY(hipMemcpy(s->d_fb_i + 2 * s->n_mels, cfg->fb_off, (size_t)s->n_mels * 4, hipMemcpyHostToDevice));
    HOP_TRY(hipMemcpy(s->d_fb_w, cfg->fb_w, (size_t)cfg->fb_nnz * 4, hipMemcpyHostToDevice));
    if (cfg->mlp) {  // the session keeps its own copy: the handle may be destroyed afterwards
        HOP_TRY(hipMalloc(&s->d_prm, (size_t)plan.n_params * 4));
        HOP_TRY(hipMemcpy(s->d_prm, cfg->mlp->d_params, (size_t)plan.n_params * 4, hipMemcpyDeviceToDevice));
        plan.params = s->d_prm;
    }
    HopArgs& a = s->args;
    std::memset(&a, 0, sizeof(a));
    a.C = C;
    a.B = B;
    a.n_mels = s->n_mels;
    a.nnz = cfg->fb_nnz;
    a.R = s->R;
    a.ctl = s->d_ctl;
    a.hop = s->d_hop;
    a.ring = s->d_ring;
    a.twM = s->d_twM;
    a.twF = s->d_twF;
    a.win = s->d_win;
    a.flo = s->d_fb_i;
    a.flen = s->d_fb_i + s->n_mels;
    a.foff = s->d_fb_i + 2 * s->n_mels;
    a.fw = s->d_fb_w;
    a.plan = plan;
    a.logits = reinterpret_cast<float*>(s->d_res + s->o_logits);
    a.mel = reinterpret_cast<float*>(s->d_res + s->o_mel);
    a.count = reinterpret_cast<int64_t*>(s->d_res + s->o_count);
    a.hop_index = reinterpret_cast<int64_t*>(s->d_res + s->o_index);
    if (cfg->strength) {
        const int bins = s->n_fft / 2 + 1;
        s->sg_floats = (size_t)bins + 4 + (size_t)cfg->strength_ring;
        HOP_TRY(hipMalloc(&s->d_sg, s->sg_floats * 4));
        StrengthArgs& g = a.sg;
        g.enabled = 1;
        g.n_ring = cfg->strength_ring;
        g.max_length = cfg->max_length;
        g.avg_length = cfg->avg_length;
        g.ls_alpha = cfg->ls_alpha;
        g.ls_minmax = cfg->ls_minmax;
        g.oe_alpha = cfg->oe_alpha;
        g.oe_minmin = cfg->oe_minmin;
        g.wsym = s->d_wsym;
        g.prev = s->d_sg;
        g.st = s->d_sg + bins;
        g.ring = s->d_sg + bins + 4;
        g.out = reinterpret_cast<float*>(s->d_res + s->o_sg);
        s->sg_init[0] = cfg->ls_max0;
        s->sg_init[1] = cfg->oe_min0;
        s->sg_init[2] = cfg->oe_max0;
        s->lds_strength = (size_t)(3 * M + M / 16 + 2) * 8 + (size_t)B * C * 4 + 64 * 4 + (size_t)2 * cfg->tg_win_length * 4;
        g.tg_len = cfg->tg_win_length;
        g.tgw = nullptr;
        if (cfg->tg_win_length > 0) {  // scipy.signal.windows.hann(W) (symmetric) as float32 (recording.py:250)
            const int W = cfg->tg_win_length;
            std::vector<float> w(W);
            for (int i = 0; i < W; ++i) w[i] = (float)(0.5 - 0.5 * cos(2.0 * 3.14159265358979323846 * i / (W - 1)));
            HOP_TRY(hipMalloc(&s->d_tgw, (size_t)W * 4));
            HOP_TRY(hipMemcpy(s->d_tgw, w.data(), (size_t)W * 4, hipMemcpyHostToDevice));
            g.tgw = s->d_tgw;
        }
        if (s->lds_strength > 160 * 1024) {
            ofp_hop_destroy(s);
            return ofp::fail(OFP_ERR_INVALID, "ofp_hop_create: the hop (%d x %d samples) does not fit the LDS", B, C);
        }
    }
    // Fused form (one launch per hop, hop and result block in pinned host memory) whenever the detector
    // fits one workgroup of the fused kernel; OFP_HOP_GRAPH=nodes keeps the five-node graph.
    {
        const char* mode = getenv("OFP_HOP_GRAPH");
        const size_t par_lds = (size_t)3 * B * C * sizeof(float);
        s->fused = !(mode && mode[0] == 'n') && 2 * C <= fused_threads(s->n_fft) && C <= ofpstream::PAR_MAX_C &&
                   par_lds <= 96 * 1024;
        s->lds_fused = std::max(std::max(s->lds, par_lds), s->lds_strength);
        if (s->fused) {
            float* dev_hop = nullptr;
            unsigned char* dev_res = nullptr;
            HOP_TRY(hipHostGetDevicePointer((void**)&dev_hop, s->h_hop, 0));
            HOP_TRY(hipHostGetDevicePointer((void**)&dev_res, s->h_res, 0));
            a.hop = dev_hop;
            a.logits = reinterpret_cast<float*>(dev_res + s->o_logits);
            a.mel = reinterpret_cast<float*>(dev_res + s->o_mel);
            a.count = reinterpret_cast<int64_t*>(dev_res + s->o_count);
            a.hop_index = reinterpret_cast<int64_t*>(dev_res + s->o_index);
            if (a.sg.enabled) a.sg.out = reinterpret_cast<float*>(dev_res + s->o_sg);
            a.done_flag = reinterpret_cast<volatile int64_t*>(dev_res + s->o_done);
            ofpstream::StreamArgs& q = s->sargs;
            q = ofpstream::make_stream_args(det);
            q.state = s->d_state;
            q.x = dev_hop;
            q.n_blocks = 1;
            q.n_rows = 0;
            q.warmup = 0;
            q.sample_base = 0;
            q.rel = s->want_rel ? reinterpret_cast<float*>(dev_res + s->o_rel) : nullptr;
            q.records = reinterpret_cast<ofp_onset*>(dev_res + s->o_rec);
            q.cap = C;
            q.count = a.count;
            q.fresh_count = 1;
        }
    }
    if ((rc = dispatch_tables(s)) != OFP_OK) return fail(rc);
    // one un-captured pass over zeros loads every code object and sets the kernel attributes
    // (neither may happen during capture), then the state is reset and the sequence captured
    if ((rc = reset_state(s)) != OFP_OK) return fail(rc);
    if ((rc = enqueue_hop(s)) != OFP_OK) return fail(rc);
    HOP_TRY(hipStreamSynchronize(s->stream));
    if ((rc = reset_state(s)) != OFP_OK) return fail(rc);
    HOP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
    rc = enqueue_hop(s);
    hipError_t ce = hipStreamEndCapture(s->stream, &s->graph);
    if (rc != OFP_OK) return fail(rc);
    HOP_TRY(ce);
    HOP_TRY(hipGraphInstantiate(&s->exec, s->graph, nullptr, nullptr, 0));
#undef HOP_TRY
    *out = s;
    return OFP_OK;
}

int ofp_hop_reset(ofp_hop_session* s) {
    OFP_REQUIRE(s, "ofp_hop_reset: NULL session");
    return reset_state(s);
}

int ofp_hop_warmup(ofp_hop_session* s, const float* h_x, int64_t n_rows) {
    OFP_REQUIRE(s && (h_x || n_rows == 0), "ofp_hop_warmup: NULL argument");
    OFP_REQUIRE(!s->in_flight, "ofp_hop_warmup: a hop is in flight (collect it first)");
    if (n_rows <= 0) return OFP_OK;
    float* d = nullptr;
    OFP_HIP(hipMalloc(&d, (size_t)n_rows * s->C * 4));
    hipError_t e = hipMemcpyAsync(d, h_x, (size_t)n_rows * s->C * 4, hipMemcpyHostToDevice, s->stream);
    int rc = e == hipSuccess ? ofp_stream_process(s->det, s->d_state, d, 0, n_rows, 1, 0, nullptr, nullptr, 0, nullptr,
                                                  s->stream)
                             : ofp::fail(OFP_ERR_HIP, "ofp_hop_warmup: %s", hipGetErrorString(e));
    (void)hipStreamSynchronize(s->stream);
    s->retired = true;
    (void)hipFree(d);
    return rc;
}

int ofp_hop_submit(ofp_hop_session* s, const float* h_hop) {
    OFP_REQUIRE(s && h_hop, "ofp_hop_submit: NULL argument");
    OFP_REQUIRE(!s->in_flight, "ofp_hop_submit: the previous hop has not been collected");
    std::memcpy(s->h_hop, h_hop, (size_t)s->B * s->C * sizeof(float));
    OFP_HIP(hipGraphLaunch(s->exec, s->stream));
    s->in_flight = true;
    s->pushed += 1;
    return OFP_OK;
}

int ofp_hop_collect(ofp_hop_session* s, int64_t* n_onsets, ofp_onset* h_records, float* h_logits, float* h_mel,
                    float* h_rel, float* h_strength) {
    OFP_REQUIRE(s, "ofp_hop_collect: NULL session");
    OFP_REQUIRE(s->in_flight, "ofp_hop_collect: no hop in flight");
    if (s->fused) {
        // the last workgroup publishes the hop count after every result: poll it instead of paying the
        // stream synchronisation's wake-up (bounded: an error or a lost launch falls through to the sync)
        const volatile int64_t* flag = reinterpret_cast<const volatile int64_t*>(s->h_res + s->o_done);
        for (int spin = 0; spin < 2000000 && *flag != s->pushed; ++spin) __builtin_ia32_pause();
        if (*flag != s->pushed) OFP_HIP(hipStreamSynchronize(s->stream));
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        s->retired = false;  // the kernel has published its results; it may not have retired from the stream yet
    } else {
        OFP_HIP(hipStreamSynchronize(s->stream));
        s->retired = true;
    }
    s->in_flight = false;
    const unsigned char* r = s->h_res;
    int64_t count, index;
    std::memcpy(&count, r + s->o_count, 8);
    std::memcpy(&index, r + s->o_index, 8);
    if (index != s->pushed - 1)
        return ofp::fail(OFP_ERR_HIP, "ofp_hop_collect: result block of hop %lld, expected %lld", (long long)index,
                         (long long)(s->pushed - 1));
    if (n_onsets) *n_onsets = count;
    if (h_records) {
        const int64_t n = count < s->C ? count : s->C;
        std::memcpy(h_records, r + s->o_rec, (size_t)n * sizeof(ofp_onset));
        for (int64_t i = 0; i < n; ++i) h_records[i].sample += index * s->B;  // audio.py:65: current_index + delta
    }
    if (h_logits && s->n_out) std::memcpy(h_logits, r + s->o_logits, (size_t)s->C * s->n_out * 4);
    if (h_mel) std::memcpy(h_mel, r + s->o_mel, (size_t)s->C * s->n_mels * 4);
    if (h_rel && s->want_rel) std::memcpy(h_rel, r + s->o_rel, (size_t)s->B * s->C * 4);
    if (h_strength && s->args.sg.enabled) std::memcpy(h_strength, r + s->o_sg, 16 + (size_t)4 * s->args.sg.tg_len);
    return OFP_OK;
}

int ofp_hop_push(ofp_hop_session* s, const float* h_hop, int64_t* n_onsets, ofp_onset* h_records, float* h_logits,
                 float* h_mel, float* h_rel, float* h_strength) {
    int rc = ofp_hop_submit(s, h_hop);
    if (rc != OFP_OK) return rc;
    return ofp_hop_collect(s, n_onsets, h_records, h_logits, h_mel, h_rel, h_strength);
}

int ofp_hop_ring_read(ofp_hop_session* s, int64_t n_rows, float* h_out) {
    OFP_REQUIRE(s && h_out && n_rows >= 0 && n_rows <= s->R, "ofp_hop_ring_read: bad argument");
    OFP_REQUIRE(!s->in_flight, "ofp_hop_ring_read: a hop is in flight (collect it first)");
    // the copies below are not stream-ordered: the last hop's kernel (polled, not synchronised) must have retired
    if (!s->retired) {
        OFP_HIP(hipStreamSynchronize(s->stream));
        s->retired = true;
    }
    // audio[-n_rows:] of the reference's CircularArray: the rows ending at the write cursor, oldest first
    const int64_t end = s->pushed * s->B;
    const size_t row = (size_t)s->C * 4;
    for (int64_t done = 0; done < n_rows;) {
        const int64_t t = end - n_rows + done;
        const int64_t p = ((t % s->R) + s->R) % s->R;
        const int64_t run = std::min<int64_t>(n_rows - done, s->R - p);
        OFP_HIP(hipMemcpy(h_out + done * s->C, reinterpret_cast<char*>(s->d_ring) + p * row, (size_t)run * row,
                          hipMemcpyDeviceToHost));
        done += run;
    }
    return OFP_OK;
}

}  // extern "C"
