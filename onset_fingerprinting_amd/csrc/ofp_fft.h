// Real FFT building blocks shared by the STFT kernels (csrc/ofp_spectral.hip) and the per-hop
// streaming kernel (csrc/ofp_hop.hip).
//
// An F-point real FFT is an M = F/2 point complex FFT of the packed sequence
// z[n] = x[2n] + i x[2n+1] followed by one split pass.  The complex FFT is a Stockham autosort
// FFT with radix-8/4 passes in place in one LDS buffer per frame; the twiddles W_M^k, the split
// twiddles W_F^k and the window are built in fp64 and rounded once.  T = M/8 lanes cooperate on
// one frame, so a 1024-point frame is exactly one 64-lane wavefront.
#pragma once
#include <cmath>

#include "ofp_common.h"

namespace ofpfft {

using ofp::cdiv;

// complex product with explicit fused multiply-adds: 4 instructions instead of 6 (the library is built with
// -ffp-contract=off for the detector's arithmetic canon, so the FMA is spelled out where it is wanted)
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }

__device__ __forceinline__ void dft2(float2& a, float2& b) {
    float2 t = csub(a, b);
    a = cadd(a, b);
    b = t;
}
__device__ __forceinline__ void dft4(float2* v) {  // outputs in natural order
    dft2(v[0], v[2]);
    dft2(v[1], v[3]);
    v[3] = mul_mi(v[3]);
    dft2(v[0], v[1]);
    dft2(v[2], v[3]);
    float2 t = v[1];
    v[1] = v[2];
    v[2] = t;
}
__device__ __forceinline__ void dft8(float2* v) {  // outputs in natural order
    const float h = 0.70710678118654752440f;
    dft2(v[0], v[4]);
    dft2(v[1], v[5]);
    dft2(v[2], v[6]);
    dft2(v[3], v[7]);
    v[5] = make_float2((v[5].x + v[5].y) * h, (v[5].y - v[5].x) * h);   // * W8^1
    v[6] = mul_mi(v[6]);                                                 // * W8^2
    v[7] = make_float2((v[7].y - v[7].x) * h, -(v[7].x + v[7].y) * h);  // * W8^3
    dft2(v[0], v[2]);
    dft2(v[1], v[3]);
    v[3] = mul_mi(v[3]);
    dft2(v[4], v[6]);
    dft2(v[5], v[7]);
    v[7] = mul_mi(v[7]);
    dft2(v[0], v[1]);
    dft2(v[2], v[3]);
    dft2(v[4], v[5]);
    dft2(v[6], v[7]);
    // bit-reversed -> natural
    float2 t;
    t = v[1]; v[1] = v[4]; v[4] = t;
    t = v[3]; v[3] = v[6]; v[6] = t;
}

template <int R>
__device__ __forceinline__ void dftR(float2* v) {
    if (R == 8) dft8(v);
    else if (R == 4) dft4(v);
    else dft2(v[0], v[1]);
}

// One Stockham pass of radix R over M points held in ONE buffer: every lane first reads the
// inputs of all its butterflies into registers, a workgroup barrier separates the reads from the
// writes, so the autosort permutation needs no second buffer (half the LDS per frame, more
// workgroups per CU).  Ns = product of the radices already done.
// Barrier between the lanes that share one frame.  Up to 64 lanes per frame, a frame lives inside
// one wave: its LDS instructions execute in order, so nothing has to wait for the other waves of the
// workgroup (which work on other frames) -- only the compiler must not move accesses across.
template <int T>
__device__ __forceinline__ void frame_sync() {
    if constexpr (T <= 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// LDS layout of a frame's M complex points: one pad word after every 16 (index w lives at w + w/16).  A wave's
// 8-byte accesses are served 16 lanes at a time, conflict-free when those 16 words fall into 16 different bank
// pairs: the pass-1 writes of the Stockham scheme go to words 8*lane + i (2 distinct bank pairs: 8-way
// conflicts), the pass-2 writes to 64*(lane/8) + lane%8 + 8 i (8-way); with the pad they are conflict-free
// and 2-way, and the unit-stride accesses stay conflict-free.  Measured before (rocprofv3 --pmc
// SQ_LDS_BANK_CONFLICT, SQ_ACTIVE_INST_LDS): 321 M conflict cycles against 198 M active LDS cycles per launch.
// The pad costs a few address registers: frames of up to 1024 points have them to spare, the 2048-point frame on
// one wave does not (it would drop from two waves per SIMD to one), so larger frames keep the plain layout.
template <int M>
__device__ __forceinline__ constexpr int fft_pad(int w) { return M <= 512 ? w + (w >> 4) : w; }
template <int M>
struct FftBuf { static constexpr int words = M <= 512 ? M + (M >> 4) : M; };  // float2 slots of one frame's buffer

// Offsets inside the padded layout are compile-time constants once the base word is padded: for a stride S that
// is a multiple of 16, fft_pad(w + i S) = fft_pad(w) + i (S + S/16); the first pass writes words 8 j + i
// (no carry into the next group of 16: + i), the second 64 a + k + 8 i with k < 8 (+ 8 i + i/2).
template <int M>
__device__ __forceinline__ constexpr int fft_pad_step(int Ns, int i) {
    return M > 512 ? i * Ns : (Ns == 1 ? i : (Ns == 8 ? 8 * i + (i >> 1) : i * (Ns + (Ns >> 4))));
}

// FROM_REGS: the inputs of the pass are handed over in registers (`in[b * R + i]` = point tid + b T + i M/R): the
// first pass of a frame whose lanes have just formed exactly those points (windowed samples) -- the frame never
// makes the round trip through the buffer before its first butterfly.
template <int R, int M, int T, bool FROM_REGS = false>
__device__ __forceinline__ void fft_pass(float2* buf, const float2* tw, int Ns, int tid, const float2* in = nullptr) {
    constexpr int NB = (M / R) / T;  // butterflies per lane (1 for radix 8, 2 for radix 4)
    static_assert((M / R) % T == 0 && NB >= 1, "lanes per frame must divide the butterflies of a pass");
    static_assert((M / R) % 16 == 0, "the read stride of a pass must be a multiple of the pad period");
    float2 v[NB][R];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = tid + b * T;
        const float2* src = buf + fft_pad<M>(j);
#pragma unroll
        for (int i = 0; i < R; ++i)
            v[b][i] = FROM_REGS ? in[b * R + i] : src[i * ((M / R) + (M <= 512 ? (M / R) / 16 : 0))];
    }
    frame_sync<T>();  // all reads of this pass are done (FROM_REGS: the previous user of the buffer is done with it)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = tid + b * T;
        const int k = j & (Ns - 1);
        const int tstep = k * (M / (Ns * R));
        if (Ns > 1) {  // the first pass has k = 0: every twiddle is 1
#pragma unroll
            for (int i = 1; i < R; ++i) v[b][i] = cmul(v[b][i], tw[(i * tstep) & (M - 1)]);
        }
        dftR<R>(v[b]);
        float2* dst = buf + fft_pad<M>((j - k) * R + k);
#pragma unroll
        for (int i = 0; i < R; ++i) dst[fft_pad_step<M>(Ns, i)] = v[b][i];
    }
    frame_sync<T>();  // all writes are visible to the next pass
}

template <int M> struct Radices;
template <> struct Radices<128>  { static constexpr int n = 3; static constexpr int r[4] = {8, 4, 4, 1}; };
template <> struct Radices<256>  { static constexpr int n = 3; static constexpr int r[4] = {8, 8, 4, 1}; };
template <> struct Radices<512>  { static constexpr int n = 3; static constexpr int r[4] = {8, 8, 8, 1}; };
template <> struct Radices<1024> { static constexpr int n = 4; static constexpr int r[4] = {8, 8, 4, 4}; };
template <> struct Radices<2048> { static constexpr int n = 4; static constexpr int r[4] = {8, 8, 8, 4}; };

// complex FFT, in place, of the M points in `a` (every thread of the workgroup calls this
// together: the passes contain workgroup barriers).
template <int M, int T>
__device__ __forceinline__ void cfft(float2* a, const float2* tw, int tid) {
    using Rx = Radices<M>;
    int Ns = 1;
    frame_sync<T>();  // the frame has been written
#pragma unroll
    for (int p = 0; p < Rx::n; ++p) {
        if (Rx::r[p] == 8) fft_pass<8, M, T>(a, tw, Ns, tid);
        else fft_pass<4, M, T>(a, tw, Ns, tid);
        Ns *= Rx::r[p];
    }
}

// The same transform with the input in registers: lane tid holds the points tid + q T, q < M / T, in `in[q]`
// (the order the first radix-8 pass wants them in: point tid + b T + i M/8 = in[b + i (M/8)/T]).
template <int M, int T>
__device__ __forceinline__ void cfft_from_regs(float2* a, const float2* tw, int tid, const float2* in) {
    using Rx = Radices<M>;
    static_assert(Rx::r[0] == 8, "the first pass is radix 8");
    constexpr int NB = (M / 8) / T;
    float2 v0[NB * 8];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 8; ++i) v0[b * 8 + i] = in[b + i * NB];
    fft_pass<8, M, T, true>(a, tw, 1, tid, v0);
    int Ns = 8;
#pragma unroll
    for (int p = 1; p < Rx::n; ++p) {
        if (Rx::r[p] == 8) fft_pass<8, M, T>(a, tw, Ns, tid);
        else fft_pass<4, M, T>(a, tw, Ns, tid);
        Ns *= Rx::r[p];
    }
}

template <int F>
struct Cfg {
    static constexpr int M = F / 2;
    // lanes per frame: M/8 (one radix-8 butterfly per lane and pass), but never more than one wavefront up
    // to 2048 points -- a 2048-point frame on 64 lanes (two butterflies per lane) synchronises its passes
    // inside the wave instead of with workgroup barriers across two waves
    static constexpr int T = (M / 8) < 16 ? 16 : ((M / 8) > 64 && F <= 2048 ? 64 : (M / 8));
    // threads per workgroup.  1024-point frames: 512 (eight frame slots) -- the tables a workgroup keeps in LDS
    // (twiddles, window, filterbank, classifier) are then shared by eight waves instead of four, and two
    // workgroups per CU are four waves per SIMD where three workgroups of 48 KB were three
    static constexpr int WG = T > 256 ? T : (F == 1024 ? 512 : 256);
    static constexpr int FPW = WG / T;                      // frames per workgroup iteration
    // LDS: twM[M] + twF[M+1] (float2), window[F] (float), one buffer of M float2 per frame
    static constexpr int MP = FftBuf<M>::words;              // (padded) buffer of one frame (fft_pad)
    static constexpr size_t lds_bytes = (size_t)(M + M + 2) * 8 + (size_t)F * 4 + (size_t)FPW * MP * 8;
    // (k_stft_power keeps half the window, see build_half_window)
    static constexpr size_t lds_bytes_half_window = (size_t)(M + M + 2) * 8 + (size_t)(F / 2 + 4) * 4 + (size_t)FPW * MP * 8;
};

template <int F>
__device__ __forceinline__ void build_tables(float2* twM, float2* twF, float* win, int frame_length) {
    constexpr int M = F / 2;
    for (int k = threadIdx.x; k < M; k += blockDim.x) {
        double s, c;
        sincospi(-2.0 * (double)k / (double)M, &s, &c);
        twM[k] = make_float2((float)c, (float)s);
    }
    for (int k = threadIdx.x; k <= M; k += blockDim.x) {
        double s, c;
        sincospi(-2.0 * (double)k / (double)F, &s, &c);
        twF[k] = make_float2((float)c, (float)s);
    }
    if (win) {
        // periodic Hann of `frame_length`, centre-padded to F (data.py:627-629)
        const int lpad = (F - frame_length) / 2;
        for (int n = threadIdx.x; n < F; n += blockDim.x) {
            int q = n - lpad;
            double w = 0.0;
            if (q >= 0 && q < frame_length) w = 0.5 - 0.5 * cospi(2.0 * (double)q / (double)frame_length);
            win[n] = (float)w;
        }
    }
}

// The periodic Hann window w[n] = 0.5 - 0.5 cos(2 pi n / F) is symmetric about n = F/2 (w[n] = w[F - n], the same
// bits: cospi(2 - x) and cospi(x) reduce to the same argument), so F/2 + 1 entries serve all F samples.
template <int F>
__device__ __forceinline__ void build_half_window(float* win) {
    for (int n = threadIdx.x; n <= F / 2; n += blockDim.x) win[n] = (float)(0.5 - 0.5 * cospi(2.0 * (double)n / (double)F));
}
template <int F>
__device__ __forceinline__ float half_window(const float* win, int n) { return win[n <= F / 2 ? n : F - n]; }

// X[k], k in [0, M], from the packed transform Z (split pass of the real FFT)
template <int M>
__device__ __forceinline__ float2 rfft_bin(const float2* Z, const float2* twF, int k) {
    float2 zk = Z[fft_pad<M>(k & (M - 1))];
    float2 zm = Z[fft_pad<M>((M - k) & (M - 1))];
    zm.y = -zm.y;
    float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y + zm.y));
    float2 o = make_float2(0.5f * (zk.x - zm.x), 0.5f * (zk.y - zm.y));
    float2 t = cmul(twF[k], o);
    return cadd(e, mul_mi(t));
}

// |X[p]|^2 and |X[M-p]|^2 of the real FFT from ONE pair (Z[p], Z[M-p]) of the packed transform: with
// e = (Z[p] + conj Z[M-p]) / 2, o = (Z[p] - conj Z[M-p]) / 2, t = W_F^p o:  X[p] = e - i t,  X[M-p] = conj(e + i t)
// -- half the loads, twiddles and products of evaluating the two bins separately.  p in [0, M/2]; for
// p = M/2 the two bins coincide (pb is then that bin again).
template <int M>
__device__ __forceinline__ void rfft_power_pair(const float2* Z, const float2* twF, int p, float& pa, float& pb) {
    const float2 zk = Z[fft_pad<M>(p & (M - 1))];
    float2 zm = Z[fft_pad<M>((M - p) & (M - 1))];
    zm.y = -zm.y;
    const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y + zm.y));
    const float2 o = make_float2(0.5f * (zk.x - zm.x), 0.5f * (zk.y - zm.y));
    const float2 t = cmul(twF[p], o);
    const float ar = e.x + t.y, ai = e.y - t.x;   // X[p]
    const float br = e.x - t.y, bi = e.y + t.x;   // conj X[M-p]
    pa = fmaf(ar, ar, ai * ai);
    pb = fmaf(br, br, bi * bi);
}

// ---- mel band sums of one frame's power spectrum (LDS), shared by k_mel, the epilogue of k_stft_power and
// the per-hop kernel so that all three give the same bits.  A band's sum is DEFINED as the sum, in order, of
// its 32-tap segments, each segment a chain acc = fma(p[k], w[k], acc) from 0: the segments of all bands are
// spread over the lanes of the frame (a wide top band no longer makes one lane walk 64-256 taps while the
// others idle), then lane b adds the segments of band b.
constexpr int MEL_SEG = 32;
constexpr int MEL_MAXSEG = 256;

struct MelSegs {            // in LDS, built once per workgroup
    unsigned short first[128];   // first segment of band b (n_mels <= 127), first[n_mels] = total
    unsigned char band[MEL_MAXSEG];
    unsigned char part[MEL_MAXSEG];
};

// single thread (or all threads redundantly writing the same values): fills the segment table
__device__ __forceinline__ void mel_build_segs(MelSegs* ms, const int32_t* flen, int n_mels) {
    int s = 0;
    for (int b = 0; b < n_mels; ++b) {
        ms->first[b] = (unsigned short)s;
        const int n = (flen[b] + MEL_SEG - 1) / MEL_SEG;
        for (int q = 0; q < n && s < MEL_MAXSEG; ++q, ++s) {
            ms->band[s] = (unsigned char)b;
            ms->part[s] = (unsigned char)q;
        }
    }
    ms->first[n_mels] = (unsigned short)s;
}

__device__ __forceinline__ float mel_segment(const float* pf, const float* fw, const int32_t* flo, const int32_t* flen,
                                             const int32_t* foff, int b, int q) {
    const int k0 = q * MEL_SEG, k1 = min(flen[b], k0 + MEL_SEG);
    const float* p = pf + flo[b];
    const float* wb = fw + foff[b];
    float acc = 0.0f;
#pragma unroll 4
    for (int k = k0; k < k1; ++k) acc = fmaf(p[k], wb[k], acc);
    return acc;
}

// the T lanes of a frame (tid in [0, T)); partial: MEL_MAXSEG floats of LDS owned by this frame; SYNC() makes the
// partial sums visible to the frame's lanes
template <class Sync, class Store>
__device__ __forceinline__ void mel_bands(const MelSegs* ms, const float* pf, const float* fw, const int32_t* flo,
                                          const int32_t* flen, const int32_t* foff, int n_mels, int tid, int T,
                                          float* partial, Sync&& sync, Store&& store) {
    const int n_segs = ms->first[n_mels];
    for (int s = tid; s < n_segs; s += T) partial[s] = mel_segment(pf, fw, flo, flen, foff, ms->band[s], ms->part[s]);
    sync();
    for (int b = tid; b < n_mels; b += T) {
        const int s0 = ms->first[b], s1 = ms->first[b + 1];
        float acc = s1 > s0 ? partial[s0] : 0.0f;
        for (int s = s0 + 1; s < s1; ++s) acc += partial[s];
        store(b, acc);
    }
}

}  // namespace ofpfft
