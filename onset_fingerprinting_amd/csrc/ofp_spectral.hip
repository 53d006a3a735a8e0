// Framing + Hann window + real FFT (+ power, mel, MFCC) on gfx950.
//
// Reference behaviour: data.py:581-654 (stft_frame / stft: float64 periodic Hann
// times the frame, np.fft.rfft, stored as complex64), data.py:657-680 (mel, dB,
// DCT through librosa), data.py:90-120 (FrameExtractor gather).
//
// FFT structure: an F-point real FFT is an M = F/2 point complex FFT of the
// packed sequence z[n] = x[2n] + i x[2n+1] followed by one split pass.  The
// complex FFT is a Stockham autosort FFT with radix-8/4 passes through two LDS
// buffers; the twiddles W_M^k, the split twiddles W_F^k and the window are built
// once per workgroup in LDS (in fp64, rounded once) and reused for every frame
// the workgroup processes.  T = M/8 lanes cooperate on one frame, so a 1024-point
// frame is exactly one 64-lane wavefront.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "ofp_common.h"
#include "ofp_fft.h"
#include "ofp_mlp.h"

namespace {

using ofp::cdiv;

using namespace ofpfft;

// ---- dense power spectra --------------------------------------------------
// Mel filterbank applied in the epilogue of k_stft_power (band-CSR as for ofp_mel): the power
// spectrum of a frame is still in LDS when its n_mels band sums are taken, so the mel output costs
// no second pass over the 2 KB-per-frame power array.
struct MelFuse {
    const int32_t *lo, *len, *off;
    const float* w;
    int n_mels, nnz;
    float* mel;  // [total_frames][n_mels]; NULL: no mel output
    int on;      // take the band sums (mel output and / or the classifier epilogue)
    int seg_stride;  // floats of partial sums per frame slot: an upper bound of the 32-tap segments (nnz / 32 + n_mels)
};

// Classifier epilogue (MLP = true): the band sums of 16 frames are collected in an LDS tile per
// tile group (the lanes that share one tile: a wave, or the 128 / 256 lanes of a 2048 / 4096-point
// frame) and pushed through the whole FCNN there (ofp_mlp.h), so neither the power spectrum nor
// the mel bands have to round-trip HBM for the logits.  Tile B of the network aliases the group's
// FFT buffers, which are idle between two frames.
struct MlpFuse {
    MlpPlan plan;
    float* logits;  // [total_frames][plan.dims[n_layers]]
};

template <int F>
struct TileCfg {
    static constexpr int T = Cfg<F>::T;
    static constexpr int TG = T < 64 ? 64 : T;   // lanes per tile group
    static constexpr int FG = TG / T;            // frames a group finishes per iteration
    static constexpr int NGRP = Cfg<F>::WG / TG; // tile groups per workgroup
    static constexpr int ITERS = 16 / FG;        // iterations that fill a 16-row tile
};

// (three waves per SIMD up to 1024 points, two for 2048: the register allocation is held to what that needs)
// SLIDE (planar input, hop = F/4, frames of up to 1024 points): a wave works through CONSECUTIVE frames of one series
// and keeps the frame's raw samples in registers; the next frame shares three quarters of them, and the shift by
// one hop maps lane tid's pair q + NP/4 onto its pair q -- a rotation inside the lane.  Only the quarter that is new
// is loaded (two 8-byte loads per lane and frame instead of eight, each sample fetched once instead of four times).
// IL (SLIDE on the caller's INTERLEAVED input [clip][time][C], C = 4 or 8 -- no planar copy needed): the frame slots of
// a workgroup are the C channels of the same run of frames (FPW / C such runs per workgroup), so the new quarter of the
// next frame is ONE contiguous block of hop x C floats per run, of which every slot's lanes fetch their own channel
// with 4-byte loads: the C waves touch the same 32-byte sectors at about the same time and share them in the CU's
// vector cache.  (An LDS stage filled with 16-byte loads and one workgroup barrier per frame was measured too: 5 %
// slower alone, and its 16 KB drop the classifier variant from two workgroups per CU to one.)
template <int F, bool MLP, bool SLIDE, bool IL = false>
__global__ __launch_bounds__(Cfg<F>::WG) __attribute__((amdgpu_waves_per_eu(F == 2048 ? 2 : (IL ? 4 : 3)))) void k_stft_power(const float* __restrict__ x, int64_t n_samples,
                                                            int C, int hop, int64_t H, int64_t total_frames,
                                                            float* __restrict__ power, MelFuse mf, int64_t planar,
                                                            MlpFuse ml) {
    static_assert(!IL || SLIDE, "IL is a form of SLIDE");
    using G = Cfg<F>;
    using TC = TileCfg<F>;
    constexpr int M = G::M, T = G::T, FPW = G::FPW;
    extern __shared__ __align__(16) unsigned char smem[];
    float2* twM = reinterpret_cast<float2*>(smem);
    float2* twF = twM + M;
    float* win = reinterpret_cast<float*>(twF + M + 2);   // half table: the periodic Hann is symmetric, w[n] = w[F - n]
    float2* bufs = reinterpret_cast<float2*>(win + F / 2 + 4);
    // mel epilogue: the filterbank; a frame's power spectrum is written back into its FFT buffer
    // (M+1 floats fit into M float2) once every lane holds its bins in registers
    float* fw = reinterpret_cast<float*>(bufs + (size_t)FPW * G::MP);
    int32_t* flo = reinterpret_cast<int32_t*>(fw + mf.nnz);
    int32_t* flen = flo + mf.n_mels;
    int32_t* foff = flen + mf.n_mels;
    if (mf.on) {
        for (int i = threadIdx.x; i < mf.nnz; i += blockDim.x) fw[i] = mf.w[i];
        for (int i = threadIdx.x; i < mf.n_mels; i += blockDim.x) {
            flo[i] = mf.lo[i];
            flen[i] = mf.len[i];
            foff[i] = mf.off[i];
        }
    }
    build_tables<F>(twM, twF, nullptr, F);
    build_half_window<F>(win);
    const int sub = threadIdx.x / T;  // frame slot within the workgroup
    const int tid = threadIdx.x % T;
    float2* A = bufs + (size_t)sub * G::MP;
    // segment table of the band sums (ofp_fft.h: mel_bands) and one array of partial sums per frame slot
    MelSegs* segs = reinterpret_cast<MelSegs*>((reinterpret_cast<uintptr_t>(foff + mf.n_mels) + 15) & ~(uintptr_t)15);
    float* partial = reinterpret_cast<float*>(segs + 1) + (size_t)sub * mf.seg_stride;
    if (mf.on && threadIdx.x == 0) mel_build_segs(segs, mf.len, mf.n_mels);
    // classifier epilogue: parameters, one tile A and 16 frame indices per tile group
    float* mprm = reinterpret_cast<float*>(segs + 1) + (size_t)FPW * mf.seg_stride;
    float* tileA = nullptr;
    float* tileB = nullptr;
    long long* rowf = nullptr;
    int it_tile = 0;
    if (MLP) {
        const int grp_id = threadIdx.x / TC::TG;
        float* tiles = mprm + ((ml.plan.n_params + 3) & ~3);
        tileA = tiles + (size_t)grp_id * 16 * ml.plan.st_a;
        rowf = reinterpret_cast<long long*>((reinterpret_cast<uintptr_t>(tiles + (size_t)TC::NGRP * 16 * ml.plan.st_a) + 7) &
                                            ~(uintptr_t)7) + grp_id * 16;
        tileB = reinterpret_cast<float*>(bufs + (size_t)grp_id * TC::FG * G::MP);
        for (int i = threadIdx.x; i < ml.plan.n_params; i += blockDim.x) mprm[i] = ml.plan.params[i];
        if ((threadIdx.x % TC::TG) < 16) rowf[threadIdx.x % TC::TG] = -1;
    }
    const int64_t n_groups = cdiv(total_frames, FPW);
    // the samples of a frame are fetched one iteration ahead into registers, so that the global
    // loads of frame g+1 are in flight while frame g goes through the FFT
    constexpr int NP = M / T;  // sample pairs per lane
    float2 nx[NP];
    // (frame index arithmetic in 32 bits: the host refuses more than 2^31 frames per launch; a 64-bit division
    //  per frame costs more than a radix-8 butterfly)
    const uint32_t H32 = (uint32_t)H, C32 = (uint32_t)C;
    auto fetch = [&](int64_t grp) {
        const int64_t f = grp * FPW + sub;
#pragma unroll
        for (int q = 0; q < NP; ++q) nx[q] = make_float2(0.0f, 0.0f);
        if (grp < n_groups && f < total_frames) {
            const uint32_t cc = (uint32_t)f / H32;     // clip*C + c
            const uint32_t h = (uint32_t)f - cc * H32;
            if (planar) {
                // planar [clip][C][time] (the detector's transposed copy): consecutive lanes, consecutive float
                // pairs -- one 8-byte load per pair at a constant distance from the lane's first
                const float2* src = reinterpret_cast<const float2*>(x + (int64_t)cc * planar + (int64_t)h * hop) + tid;
                if ((reinterpret_cast<uintptr_t>(src) & 7u) == 0) {
#pragma unroll
                    for (int q = 0; q < NP; ++q) nx[q] = src[q * T];
                } else {  // odd hop or series offset: the pair straddles an 8-byte boundary
                    const float* s1 = reinterpret_cast<const float*>(src);
#pragma unroll
                    for (int q = 0; q < NP; ++q) nx[q] = make_float2(s1[2 * q * T], s1[2 * q * T + 1]);
                }
            } else {
                // interleaved [clip][time][C] (the caller's layout: a lane's two samples sit in two 32-B sectors of
                // which it uses 4 B each)
                const uint32_t clip = cc / C32, c = cc - clip * C32;
                const float* src = x + ((int64_t)clip * n_samples + (int64_t)h * hop) * C + c;
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const int n = tid + q * T;
                    nx[q] = make_float2(src[(int64_t)(2 * n) * C], src[(int64_t)(2 * n + 1) * C]);
                }
            }
        }
    };
    // frame of slot `sub` in iteration `it`.  Plain: the workgroups interleave groups of FPW consecutive frames.
    // SLIDE: workgroup b owns FPW K consecutive frames, slot `sub` the K consecutive ones from b FPW K + sub K.
    // IL: a run = the C slots sub / C * C ... of this workgroup; row-frame rf = clip H + h, the same for its C slots
    const int il_gpw = IL ? FPW / C : 1;                      // runs per workgroup
    const int il_c = IL ? sub % C : 0, il_g = IL ? sub / C : 0;
    const int64_t total_rf = IL ? total_frames / C : 0;
    const int64_t n_it = IL ? cdiv(total_rf, (int64_t)gridDim.x * il_gpw)
                            : (SLIDE ? cdiv(n_groups, (int64_t)gridDim.x)
                                     : (n_groups > (int64_t)blockIdx.x ? cdiv(n_groups - (int64_t)blockIdx.x, (int64_t)gridDim.x) : 0));
    auto frame_of = [&](int64_t it) -> int64_t {
        if (IL) {
            const int64_t rf = ((int64_t)blockIdx.x * il_gpw + il_g) * n_it + it;
            if (rf >= total_rf) return total_frames;  // (invalid)
            const uint32_t clip = (uint32_t)rf / H32, h = (uint32_t)rf - clip * H32;
            return ((int64_t)clip * C + il_c) * H + h;
        }
        return SLIDE ? ((int64_t)blockIdx.x * FPW + sub) * n_it + it : ((int64_t)blockIdx.x + it * gridDim.x) * FPW + sub;
    };
    // IL: the slot's series from frame h on as a buffer resource (base = that address, made a scalar: a slot is a whole
    // wave for T = 64): a lane's load is then ONE 32-bit offset register plus scalar / immediate parts.  (With plain
    // pointers the compiler carried a 64-bit address pair per load: 40 more registers than the planar form, one
    // workgroup per CU instead of two.)
    auto il_base = [&](uint32_t clip, uint32_t h) -> __amdgpu_buffer_rsrc_t {
        const int64_t o = ((int64_t)clip * n_samples + (int64_t)h * hop) * C + il_c;
        float* b = const_cast<float*>(x) + o;
        if (T == 64) {
            const uintptr_t u = reinterpret_cast<uintptr_t>(b);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
            b = reinterpret_cast<float*>(((uintptr_t)hi << 32) | lo);
        }
        return __builtin_amdgcn_make_buffer_rsrc(b, 0, 0x7fffffff, 0x00020000);
    };
    auto il_at = [](__amdgpu_buffer_rsrc_t r, uint32_t byte_off, uint32_t s_off) -> float {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, (int)s_off, 0));
    };

    constexpr int NS = NP / 4;  // SLIDE: pairs per lane that are new in the next frame
    float2 nw[NS > 0 ? NS : 1];
    if constexpr (!SLIDE) fetch(blockIdx.x);
    __syncthreads();  // tables ready
    for (int64_t it = 0; it < n_it; ++it) {
        const int64_t grp = (int64_t)blockIdx.x + it * gridDim.x;  // (plain mapping only)
        const int64_t f = frame_of(it);  // flattened (clip, channel, hop)
        const bool valid = f < total_frames;
        const bool last_it = it + 1 == n_it;
        if constexpr (IL) {
            const uint32_t cc = valid ? (uint32_t)f / H32 : 0u;
            const uint32_t h = valid ? (uint32_t)f - cc * H32 : 0u;
            const uint32_t clip = cc / C32;
            if (!valid) {
#pragma unroll
                for (int q = 0; q < NP; ++q) nx[q] = make_float2(0.0f, 0.0f);
            } else if (it == 0 || h == 0) {  // the first frame of this run, or of a clip: all of it, strided
                const __amdgpu_buffer_rsrc_t src = il_base(clip, h);
                const uint32_t o = (uint32_t)(8 * tid) * C32;  // the lane's byte offset; the pair index q T goes into the scalar part
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    nx[q] = make_float2(il_at(src, o, (uint32_t)(8 * q * T) * C32), il_at(src, o + 4u * C32, (uint32_t)(8 * q * T) * C32));
            } else {  // rotate, append the new quarter
#pragma unroll
                for (int q = 0; q < NP - NS; ++q) nx[q] = nx[q + NS];
#pragma unroll
                for (int q = 0; q < NS; ++q) nx[NP - NS + q] = nw[q];
            }
            // the lane's own four samples of the run's next block (hop rows x C channels from row h hop + F on): in flight
            // during this frame's FFT
            const bool more = valid && !last_it && h + 1 < H32;
            if (more) {
                const __amdgpu_buffer_rsrc_t src = il_base(clip, h);
                const uint32_t o = (uint32_t)(8 * tid) * C32;
#pragma unroll
                for (int q = 0; q < NS; ++q) {
                    const uint32_t so = (uint32_t)(4 * F + 8 * q * T) * C32;
                    nw[q] = make_float2(il_at(src, o, so), il_at(src, o + 4u * C32, so));
                }
            } else {
#pragma unroll
                for (int q = 0; q < NS; ++q) nw[q] = make_float2(0.0f, 0.0f);
            }
        } else if constexpr (SLIDE) {
            const uint32_t cc = valid ? (uint32_t)f / H32 : 0u;
            const uint32_t h = valid ? (uint32_t)f - cc * H32 : 0u;
            const float2* src = reinterpret_cast<const float2*>(x + (int64_t)cc * planar + (int64_t)h * hop) + tid;
            if (!valid) {
#pragma unroll
                for (int q = 0; q < NP; ++q) nx[q] = make_float2(0.0f, 0.0f);
            } else if (it == 0 || h == 0) {  // the first frame of this wave's run, or of a series: all of it
#pragma unroll
                for (int q = 0; q < NP; ++q) nx[q] = src[q * T];
            } else {  // frame f - 1 of the same series was the previous iteration: rotate, append the new quarter
#pragma unroll
                for (int q = 0; q < NP - NS; ++q) nx[q] = nx[q + NS];
#pragma unroll
                for (int q = 0; q < NS; ++q) nx[NP - NS + q] = nw[q];
            }
            // the new quarter of the next frame: in flight during this frame's FFT
            const bool more = valid && !last_it && h + 1 < H32 && f + 1 < total_frames;
#pragma unroll
            for (int q = 0; q < NS; ++q) nw[q] = more ? src[(hop >> 1) + (NP - NS + q) * T] : make_float2(0.0f, 0.0f);
        }
        // the windowed points go straight from the lane's registers into its first butterfly: lane tid holds exactly
        // the points tid + q T the first radix-8 pass of this lane reads (no round trip through the frame buffer)
        float2 wx[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int n = tid + q * T;
            wx[q] = make_float2(nx[q].x * half_window<F>(win, 2 * n), nx[q].y * half_window<F>(win, 2 * n + 1));
        }
        // the next frame's samples: in flight during this frame's FFT -- unless a lane holds 16+ pairs (2048-point
        // frames on one wave), where keeping them live across the passes costs more registers than the kernel
        // has: those are fetched after the passes, in flight during the epilogue
        if constexpr (!SLIDE && NP <= 8) fetch(grp + gridDim.x);
        cfft_from_regs<M, T>(A, twM, tid, wx);
        if constexpr (!SLIDE && NP > 8) fetch(grp + gridDim.x);
        // power spectrum, two bins (p, M - p) per pair of the packed transform: pairs p = tid, tid + T, ... <= M/2
        constexpr int NQ = (M / 2) / T + 1;
        float pa[NQ], pb[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pp = tid + q * T;
            pa[q] = pb[q] = 0.0f;
            if (pp <= M / 2) rfft_power_pair<M>(A, twF, pp, pa[q], pb[q]);
        }
        if (valid && power) {
            float* dst = power + f * (M + 1);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int pp = tid + q * T;
                if (pp <= M / 2) {
                    dst[pp] = pa[q];
                    if (pp < M / 2) dst[M - pp] = pb[q];
                }
            }
        }
        if (mf.on) {
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
            const int row = MLP ? it_tile * TC::FG + (sub % TC::FG) : 0;
            mel_bands(segs, pf, fw, flo, flen, foff, mf.n_mels, tid, T, partial, [] { frame_sync<T>(); },
                      [&](int b, float acc) {
                          if (valid && mf.mel) mf.mel[f * mf.n_mels + b] = acc;
                          if (MLP) tileA[row * ml.plan.st_a + b] = acc;
                      });
            if (MLP && tid == 0) rowf[row] = valid ? (long long)f : -1;
        }
        if (MLP) {
            // a tile is pushed through the network when its 16 rows are filled, or at the last iteration
            it_tile += 1;
            if (it_tile == TC::ITERS || last_it) {
                frame_sync<TC::TG>();  // tile A complete; the group's FFT buffers are idle
                if ((threadIdx.x % TC::TG) < 64) {
                    const int lane = threadIdx.x & 63;
                    const int nout = ml.plan.dims[ml.plan.n_layers];
                    ofp_mlp_tile(ml.plan, mprm, tileA, tileB, lane, [&](int r, int col, float v) {
                        const long long fr = rowf[r];
                        if (fr >= 0) ml.logits[fr * nout + col] = v;
                    });
                    ofp_wave_lds_sync();
                    if (lane < 16) rowf[lane] = -1;
                }
                frame_sync<TC::TG>();  // tile B (the FFT buffers) is free again
                it_tile = 0;
            }
        }
    }
}

// ---- gathered complex frames (data.stft) ------------------------------------
struct FrameArgs {
    const float* x;
    int64_t n_samples;
    int C;
    const int32_t* clip;
    const int32_t* channel;
    const int64_t* start;
    const int64_t* valid_lo;
    const int64_t* valid_hi;
    int64_t n_frames;
    int frame_length;
    const float* window;  // [F]
    float2* spec;         // [n_frames][F/2+1]
};

template <int F>
__global__ __launch_bounds__(Cfg<F>::WG) void k_stft_frames(FrameArgs a) {
    using G = Cfg<F>;
    constexpr int M = G::M, T = G::T, FPW = G::FPW;
    extern __shared__ __align__(16) unsigned char smem[];
    float2* twM = reinterpret_cast<float2*>(smem);
    float2* twF = twM + M;
    float* win = reinterpret_cast<float*>(twF + M + 2);
    float2* bufs = reinterpret_cast<float2*>(win + F);
    build_tables<F>(twM, twF, nullptr, F);
    for (int n = threadIdx.x; n < F; n += blockDim.x) win[n] = a.window[n];
    const int sub = threadIdx.x / T;
    const int tid = threadIdx.x % T;
    float2* A = bufs + (size_t)sub * G::MP;
    const int lpad = (F - a.frame_length) / 2;  // librosa.util.pad_center (data.py:588-589)
    const int64_t n_groups = cdiv(a.n_frames, FPW);
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int64_t f = grp * FPW + sub;
        const bool valid = f < a.n_frames;
        __syncthreads();
        if (valid) {
            const int64_t st = a.start[f], lo = a.valid_lo[f], hi = a.valid_hi[f];
            const float* src = a.x + ((int64_t)a.clip[f] * a.n_samples) * a.C + a.channel[f];
            for (int n = tid; n < M; n += T) {
                float v[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    int p = 2 * n + e;
                    int q = p - lpad;
                    int64_t idx = st + q;
                    float s = 0.0f;
                    if (q >= 0 && q < a.frame_length && idx >= lo && idx < hi && idx >= 0 && idx < a.n_samples)
                        s = src[idx * a.C];
                    v[e] = s * win[p];
                }
                A[fft_pad<M>(n)] = make_float2(v[0], v[1]);
            }
        }
        cfft<M, T>(A, twM, tid);
        if (valid) {
            float2* dst = a.spec + f * (M + 1);
            for (int k = tid; k <= M; k += T) dst[k] = rfft_bin<M>(A, twF, k);
        }
    }
}

// ---- FrameExtractor gather (data.py:90-120) -----------------------------------
__global__ __launch_bounds__(256) void k_extract(const float* __restrict__ x, int64_t n_samples, int C,
                                                 const int64_t* __restrict__ start, int64_t n_onsets, int width,
                                                 float* __restrict__ out) {
    const int64_t total = n_onsets * C * width;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int w = (int)(i % width);
        int64_t oc = i / width;
        int c = (int)(oc % C);
        int64_t t = start[oc] + w;
        out[i] = (t >= 0 && t < n_samples) ? x[t * C + c] : 0.0f;
    }
}

// ---- mel: sparse triangular filterbank, one thread per (row, band) ------------
__global__ __launch_bounds__(256) void k_mel(const float* __restrict__ power, int64_t n_rows, int n_bins,
                                             int n_mels, const int32_t* __restrict__ lo,
                                             const int32_t* __restrict__ len, const int32_t* __restrict__ off,
                                             const float* __restrict__ w, float* __restrict__ mel) {
    const int64_t total = n_rows * n_mels;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int b = (int)(i % n_mels);
        int64_t r = i / n_mels;
        // the band sum as ofp_fft.h defines it: 32-tap segments, each a chain from 0, added in order
        const float* pf = power + r * n_bins;
        const int nseg = (len[b] + MEL_SEG - 1) / MEL_SEG;
        float acc = 0.0f;
        for (int q = 0; q < nseg; ++q) {
            const float part = mel_segment(pf, w, lo, len, off, b, q);
            acc = q == 0 ? part : acc + part;
        }
        mel[i] = acc;
    }
}

// ---- power_to_db + DCT (data.py:677-679) ---------------------------------------
__global__ __launch_bounds__(256) void k_max(const float* __restrict__ v, int64_t n, float* out) {
    float m = 0.0f;  // inputs are non-negative powers
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        m = fmaxf(m, v[i]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(m));
}

__global__ __launch_bounds__(256) void k_mfcc(const float* __restrict__ mel, int64_t n_rows, int n_mels,
                                              int n_mfcc, float amin, float top_db,
                                              const float* __restrict__ dct, const float* __restrict__ gmax,
                                              float* __restrict__ out) {
    const int64_t total = n_rows * n_mfcc;
    float lo = -INFINITY;
    if (top_db >= 0.0f) lo = 10.0f * log10f(fmaxf(amin, *gmax)) - top_db;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int q = (int)(i % n_mfcc);
        int64_t r = i / n_mfcc;
        const float* m = mel + r * n_mels;
        const float* d = dct + (int64_t)q * n_mels;
        float acc = 0.0f;
        for (int b = 0; b < n_mels; ++b) {
            float db = fmaxf(10.0f * log10f(fmaxf(amin, m[b])), lo);
            acc = fmaf(db, d[b], acc);
        }
        out[i] = acc;
    }
}

template <int F, bool MLP, bool SLIDE, bool IL = false>
int launch_power_s(const float* x, int64_t n_samples, int C, int hop, int64_t H, int64_t total, float* power,
                   const MelFuse& mf, int64_t planar, const MlpFuse& ml, hipStream_t stream) {
    using G = Cfg<F>;
    using TC = TileCfg<F>;
    size_t lds = G::lds_bytes_half_window;
    if (mf.on) lds += (size_t)mf.nnz * 4 + (size_t)3 * mf.n_mels * 4 + 16 + sizeof(MelSegs) + (size_t)G::FPW * mf.seg_stride * 4;
    if (MLP) {
        // tile B lives in the tile group's idle FFT buffers
        OFP_REQUIRE((size_t)16 * ml.plan.st_b * 4 <= (size_t)TC::FG * G::M * 8,
                    "classifier epilogue: hidden layers wider than %d do not fit the FFT buffers of a %d-point frame",
                    (int)((size_t)TC::FG * G::M * 8 / 64) - 1, F);
        lds += (size_t)((ml.plan.n_params + 3) & ~3) * 4 + (size_t)TC::NGRP * 16 * ml.plan.st_a * 4 +
               (size_t)TC::NGRP * 16 * 8 + 16;
        OFP_REQUIRE(lds <= 160 * 1024, "classifier epilogue: %zu bytes of LDS needed, 160 KiB available", lds);
    }
    if (getenv("OFP_DEBUG_LDS")) fprintf(stderr, "k_stft_power<%d,%d>: %zu bytes of LDS per workgroup\n", F, (int)MLP, lds);
    static ofp::LdsAttrCache attr;  // (one per instantiation)
    if (int rc = ofp::ensure_dynamic_lds(reinterpret_cast<const void*>(k_stft_power<F, MLP, SLIDE, IL>), lds, attr)) return rc;
    int64_t groups = cdiv(total, G::FPW);
    unsigned grid = (unsigned)std::min<int64_t>(groups, 256 * 8);
    hipLaunchKernelGGL((k_stft_power<F, MLP, SLIDE, IL>), dim3(grid), dim3(G::WG), lds, stream, x, n_samples, C, hop, H, total,
                       power, mf, planar, ml);
    OFP_LAUNCH_CHECK("k_stft_power");
    return OFP_OK;
}

template <int F, bool MLP>
int launch_power_t(const float* x, int64_t n_samples, int C, int hop, int64_t H, int64_t total, float* power,
                   const MelFuse& mf, int64_t planar, const MlpFuse& ml, hipStream_t stream) {
    // consecutive frames per wave with the shared samples kept in registers: planar input, hop = F/4, 8-byte
    // aligned pairs (see the kernel)
    if constexpr (F <= 2048) {
        if (planar && hop * 4 == F && (planar & 1) == 0 && (reinterpret_cast<uintptr_t>(x) & 7u) == 0 && !getenv("OFP_STFT_NO_SLIDE"))
            return launch_power_s<F, MLP, true>(x, n_samples, C, hop, H, total, power, mf, planar, ml, stream);
    }
    // the caller's interleaved input with 4 or 8 channels: the same on strided 4-byte loads (see the kernel)
    if constexpr (F <= 1024) {
        if (!planar && hop * 4 == F && (C == 4 || C == 8) && Cfg<F>::FPW % C == 0 && (reinterpret_cast<uintptr_t>(x) & 3u) == 0 &&
            !getenv("OFP_STFT_NO_SLIDE"))
            return launch_power_s<F, MLP, true, true>(x, n_samples, C, hop, H, total, power, mf, planar, ml, stream);
    }
    return launch_power_s<F, MLP, false>(x, n_samples, C, hop, H, total, power, mf, planar, ml, stream);
}

template <int F>
int launch_power(const float* x, int64_t n_samples, int C, int hop, int64_t H, int64_t total, float* power,
                 const MelFuse& mf, int64_t planar, const MlpFuse* ml, hipStream_t stream) {
    if (ml) return launch_power_t<F, true>(x, n_samples, C, hop, H, total, power, mf, planar, *ml, stream);
    return launch_power_t<F, false>(x, n_samples, C, hop, H, total, power, mf, planar, MlpFuse{}, stream);
}

template <int F>
int launch_frames(const FrameArgs& a, hipStream_t stream) {
    using G = Cfg<F>;
    static ofp::LdsAttrCache attr;
    if (int rc = ofp::ensure_dynamic_lds(reinterpret_cast<const void*>(k_stft_frames<F>), G::lds_bytes, attr)) return rc;
    int64_t groups = cdiv(a.n_frames, G::FPW);
    unsigned grid = (unsigned)std::min<int64_t>(groups, 256 * 8);
    hipLaunchKernelGGL(k_stft_frames<F>, dim3(grid), dim3(G::WG), G::lds_bytes, stream, a);
    OFP_LAUNCH_CHECK("k_stft_frames");
    return OFP_OK;
}

}  // namespace

extern "C" {

static int stft_power_impl(const float* d_x, int64_t n_clips, int64_t n_samples, int32_t C, int32_t n_fft,
                           int32_t hop, float* d_power, const MelFuse& mf, int64_t planar, void* stream_,
                           const MlpFuse* ml = nullptr) {
    OFP_REQUIRE(d_x && (d_power || mf.mel || ml), "ofp_stft_power: NULL argument");
    OFP_REQUIRE(n_clips >= 1 && C >= 1 && hop >= 1, "ofp_stft_power: bad sizes");
    if (n_samples < n_fft) return OFP_OK;  // no complete frame
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t H = 1 + (n_samples - n_fft) / hop;
    const int64_t total = n_clips * C * H;
    OFP_REQUIRE(total < (1ll << 31), "ofp_stft_power: %lld frames in one launch (at most 2^31 - 1)", (long long)total);
    switch (n_fft) {
        case 256: return launch_power<256>(d_x, n_samples, C, hop, H, total, d_power, mf, planar, ml, stream);
        case 512: return launch_power<512>(d_x, n_samples, C, hop, H, total, d_power, mf, planar, ml, stream);
        case 1024: return launch_power<1024>(d_x, n_samples, C, hop, H, total, d_power, mf, planar, ml, stream);
        case 2048: return launch_power<2048>(d_x, n_samples, C, hop, H, total, d_power, mf, planar, ml, stream);
        case 4096: return launch_power<4096>(d_x, n_samples, C, hop, H, total, d_power, mf, planar, ml, stream);
        default: return ofp::fail(OFP_ERR_INVALID, "n_fft %d not supported (256,512,1024,2048,4096)", n_fft);
    }
}

int ofp_stft_power(const float* d_x, int64_t n_clips, int64_t n_samples, int32_t C, int32_t n_fft,
                   int32_t hop, float* d_power, void* stream) {
    return stft_power_impl(d_x, n_clips, n_samples, C, n_fft, hop, d_power, MelFuse{}, 0, stream);
}

int ofp_stft_power_mel(const float* d_x, int64_t n_clips, int64_t n_samples, int32_t C, int32_t n_fft,
                       int32_t hop, float* d_power, int32_t n_mels, const int32_t* d_fb_lo,
                       const int32_t* d_fb_len, const int32_t* d_fb_off, const float* d_fb_w, int32_t fb_nnz,
                       float* d_mel, int64_t planar_stride, void* stream) {
    OFP_REQUIRE(d_fb_lo && d_fb_len && d_fb_off && d_fb_w && d_mel && n_mels >= 1 && fb_nnz >= 1,
                "ofp_stft_power_mel: NULL / empty filterbank");
    OFP_REQUIRE(fb_nnz <= 4 * (n_fft / 2 + 1), "ofp_stft_power_mel: filterbank with %d weights for %d bins", fb_nnz,
                n_fft / 2 + 1);
    OFP_REQUIRE(n_mels <= 127 && fb_nnz / MEL_SEG + n_mels <= MEL_MAXSEG, "ofp_stft_power_mel: at most 127 bands and %d 32-tap segments",
                MEL_MAXSEG);
    MelFuse mf{d_fb_lo, d_fb_len, d_fb_off, d_fb_w, n_mels, fb_nnz, d_mel, 1, ((fb_nnz / MEL_SEG + n_mels + 3) & ~3)};
    return stft_power_impl(d_x, n_clips, n_samples, C, n_fft, hop, d_power, mf, planar_stride, stream);
}

int ofp_stft_power_mel_mlp(const float* d_x, int64_t n_clips, int64_t n_samples, int32_t C, int32_t n_fft,
                           int32_t hop, float* d_power, int32_t n_mels, const int32_t* d_fb_lo,
                           const int32_t* d_fb_len, const int32_t* d_fb_off, const float* d_fb_w, int32_t fb_nnz,
                           float* d_mel, int64_t planar_stride, const ofp_mlp* mlp, float* d_logits, void* stream) {
    OFP_REQUIRE(d_fb_lo && d_fb_len && d_fb_off && d_fb_w && n_mels >= 1 && fb_nnz >= 1,
                "ofp_stft_power_mel_mlp: NULL / empty filterbank");
    OFP_REQUIRE(fb_nnz <= 4 * (n_fft / 2 + 1), "ofp_stft_power_mel_mlp: filterbank with %d weights for %d bins", fb_nnz,
                n_fft / 2 + 1);
    OFP_REQUIRE(n_mels <= 127 && fb_nnz / MEL_SEG + n_mels <= MEL_MAXSEG, "ofp_stft_power_mel_mlp: at most 127 bands and %d 32-tap segments",
                MEL_MAXSEG);
    OFP_REQUIRE(mlp && d_logits, "ofp_stft_power_mel_mlp: NULL classifier / output");
    OFP_REQUIRE(mlp->plan.dims[0] == n_mels, "ofp_stft_power_mel_mlp: the classifier takes %d inputs, the filterbank has %d bands",
                mlp->plan.dims[0], n_mels);
    MelFuse mf{d_fb_lo, d_fb_len, d_fb_off, d_fb_w, n_mels, fb_nnz, d_mel, 1, ((fb_nnz / MEL_SEG + n_mels + 3) & ~3)};
    MlpFuse ml{mlp->plan, d_logits};
    return stft_power_impl(d_x, n_clips, n_samples, C, n_fft, hop, d_power, mf, planar_stride, stream, &ml);
}

int ofp_stft_frames(const float* d_x, int64_t n_clips, int64_t n_samples, int32_t C, const int32_t* d_clip,
                    const int32_t* d_channel, const int64_t* d_start, const int64_t* d_valid_lo,
                    const int64_t* d_valid_hi, int64_t n_frames, int32_t frame_length, int32_t n_fft,
                    const float* d_window, float* d_spec, void* stream_) {
    if (n_frames == 0) return OFP_OK;
    OFP_REQUIRE(d_x && d_clip && d_channel && d_start && d_valid_lo && d_valid_hi && d_window && d_spec,
                "ofp_stft_frames: NULL argument");
    OFP_REQUIRE(frame_length >= 1 && frame_length <= n_fft, "frame_length %d must be in [1, n_fft]", frame_length);
    (void)n_clips;
    FrameArgs a;
    a.x = d_x; a.n_samples = n_samples; a.C = C; a.clip = d_clip; a.channel = d_channel; a.start = d_start;
    a.valid_lo = d_valid_lo; a.valid_hi = d_valid_hi; a.n_frames = n_frames; a.frame_length = frame_length;
    a.window = d_window; a.spec = reinterpret_cast<float2*>(d_spec);
    hipStream_t stream = (hipStream_t)stream_;
    switch (n_fft) {
        case 256: return launch_frames<256>(a, stream);
        case 512: return launch_frames<512>(a, stream);
        case 1024: return launch_frames<1024>(a, stream);
        case 2048: return launch_frames<2048>(a, stream);
        case 4096: return launch_frames<4096>(a, stream);
        default: return ofp::fail(OFP_ERR_INVALID, "n_fft %d not supported (256,512,1024,2048,4096)", n_fft);
    }
}

int ofp_extract_frames(const float* d_x, int64_t n_samples, int32_t C, const int64_t* d_start,
                       int64_t n_onsets, int32_t width, float* d_out, void* stream) {
    if (n_onsets == 0) return OFP_OK;
    OFP_REQUIRE(d_x && d_start && d_out && width >= 1 && C >= 1, "ofp_extract_frames: bad argument");
    int64_t total = n_onsets * C * width;
    unsigned grid = (unsigned)std::min<int64_t>(cdiv(total, 256), 256 * 16);
    hipLaunchKernelGGL(k_extract, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_x, n_samples, C, d_start,
                       n_onsets, width, d_out);
    OFP_LAUNCH_CHECK("k_extract");
    return OFP_OK;
}

int ofp_mel(const float* d_power, int64_t n_rows, int32_t n_bins, int32_t n_mels, const int32_t* d_fb_lo,
            const int32_t* d_fb_len, const int32_t* d_fb_off, const float* d_fb_w, float* d_mel, void* stream) {
    if (n_rows == 0) return OFP_OK;
    OFP_REQUIRE(d_power && d_fb_lo && d_fb_len && d_fb_off && d_fb_w && d_mel, "ofp_mel: NULL argument");
    OFP_REQUIRE(n_mels <= 127, "ofp_mel: at most 127 bands (got %d)", n_mels);
    int64_t total = n_rows * n_mels;
    unsigned grid = (unsigned)std::min<int64_t>(cdiv(total, 256), 256 * 16);
    hipLaunchKernelGGL(k_mel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_power, n_rows, n_bins, n_mels,
                       d_fb_lo, d_fb_len, d_fb_off, d_fb_w, d_mel);
    OFP_LAUNCH_CHECK("k_mel");
    return OFP_OK;
}

int ofp_mfcc(const float* d_mel, int64_t n_rows, int32_t n_mels, int32_t n_mfcc, float amin, float top_db,
             const float* d_dct, float* d_mfcc, float* d_scratch, void* stream_) {
    if (n_rows == 0) return OFP_OK;
    OFP_REQUIRE(d_mel && d_dct && d_mfcc && d_scratch, "ofp_mfcc: NULL argument");
    hipStream_t stream = (hipStream_t)stream_;
    OFP_HIP(hipMemsetAsync(d_scratch, 0, 4, stream));
    int64_t n = n_rows * n_mels;
    if (top_db >= 0.0f) {
        unsigned g1 = (unsigned)std::min<int64_t>(cdiv(n, 256), 1024);
        hipLaunchKernelGGL(k_max, dim3(g1), dim3(256), 0, stream, d_mel, n, d_scratch);
        OFP_LAUNCH_CHECK("k_max");
    }
    int64_t total = n_rows * n_mfcc;
    unsigned grid = (unsigned)std::min<int64_t>(cdiv(total, 256), 256 * 16);
    hipLaunchKernelGGL(k_mfcc, dim3(grid), dim3(256), 0, stream, d_mel, n_rows, n_mels, n_mfcc, amin, top_db,
                       d_dct, (const float*)d_scratch, d_mfcc);
    OFP_LAUNCH_CHECK("k_mfcc");
    return OFP_OK;
}

}  // extern "C"
