// Amplitude onset detector on gfx950 -- offline (batched, time-parallel) form.
//
// Reference behaviour: detection.py:19-86 (driver), :595-888 (detector),
// envelope_follower.c (followers / tracker / backtracking).  The arithmetic of
// every step is the single definition in include/ofp_math.h.
//
// How a sequential recurrence is made time-parallel WITHOUT changing a bit
// ("chunk-Jacobi"): the stream of one chain (clip x channel) is cut into chunks;
// pass 0 runs every chunk from a guessed state after a speculative warm-up of W
// samples and records the state it actually started the chunk from (`used`) and
// the state it ended with (`end`).  Pass j >= 1 re-runs exactly those chunks whose
// recorded start state differs (bitwise) from the end state of the preceding
// chunk, starting from that end state.  Chunk 0 always starts from the true
// initial state, so by induction the fixed point of this iteration IS the
// sequential result; the host stops when a pass changes nothing.  With an
// adequate warm-up the common case is pass 0 + one verification pass that only
// compares states.  The worst case degenerates to sequential speed, never to a
// wrong answer.
//
// Stages (each its own chunk-Jacobi loop, in stream order):
//   hp : 4th-order high-pass IIR (detection.py:743-744)      state z[4]
//   db : rectified dB with floor (detection.py:747-748)      elementwise
//   ar : fast & slow attack/release followers (:751)         state (yf, ys)
//   rel: back to linear + clip (:753-754)                    elementwise
//   mm : EMA min/max tracker (:762) -> thresholds per block  state (mn, mx)
//   scan + state machine: threshold crossings, hysteresis, cooldown (:764-797)
//   backtrack (:800-825)
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/ofp_math.h"
#include "ofp_common.h"

#include "ofp_detector.h"

namespace {

using ofp::align_up;
using ofp::cdiv;

// ---------------------------------------------------------------------------
// stream geometry of one clip (all in samples)
struct Geom {
    int64_t N;     // samples per clip
    int64_t Nm;    // floor(N/B)*B main samples
    int64_t n_w;   // warm samples through the high-pass (detection.py:70,828-829)
    int64_t n_wb;  // warm samples through followers/tracker: full blocks only (:832-834)
    int64_t U;     // follower stream length n_wb + Nm
    int64_t V;     // high-pass stream length n_w + Nm
    int32_t C, B;
};

__host__ __device__ inline int64_t hp_src(const Geom& g, int64_t v) { return v < g.n_w ? v : v - g.n_w; }
__host__ __device__ inline int64_t hp_dst(const Geom& g, int64_t v) {
    return v < g.n_wb ? v : (v >= g.n_w ? v - g.n_w + g.n_wb : -1);
}
__host__ __device__ inline int64_t u_src(const Geom& g, int64_t u) { return u < g.n_wb ? u : u - g.n_wb; }

// ---------------------------------------------------------------------------
// stages.  Each stage describes one sequential recurrence over a stream:
//   NS / State      the carried state words
//   NBRK / brk(i)   stream positions where the stream<->memory mapping changes
//                   (between two breaks input and output addresses are affine in
//                   the position, so the inner loop only increments pointers)
//   in_ptr/out_ptr  address of element (clip, c, pos); out_ptr may be null
//   compute         one step (include/ofp_math.h), returns the dense output value
struct HpStage {
    static constexpr int NS = 4;
    static constexpr int NBRK = 2;
    static constexpr bool DENSE_OUT = true;
    struct State { float z[4]; };
    struct Sparse { int rem; };
    Geom g;
    const float* x;  // [clips][N][C]
    float* out;      // [clips][U][C]
    float b[5], a[5];
    int64_t L, W, n_chunks;
    __device__ int64_t len() const { return g.V; }
    __device__ int64_t brk(int i) const { return i == 0 ? g.n_wb : g.n_w; }
    __device__ State init(int64_t, int) const { return State{{0.f, 0.f, 0.f, 0.f}}; }
    __device__ State guess(int64_t, int, int64_t) const { return State{{0.f, 0.f, 0.f, 0.f}}; }
    __device__ const float* in_ptr(int64_t clip, int c, int64_t v) const {
        return x + (clip * g.N + hp_src(g, v)) * g.C + c;
    }
    __device__ float* out_ptr(int64_t clip, int c, int64_t v) const {
        int64_t u = hp_dst(g, v);
        return u >= 0 ? out + (clip * g.U + u) * g.C + c : nullptr;
    }
    // ofp_df2t4_step with the same operations in the same order, arranged as 2-wide
    // vectors so the compiler can use packed fp32 (v_pk_mul_f32 / v_pk_add_f32 are
    // per-lane IEEE fp32): pairs (z0,z2) and (z1,z3); the last tap adds -0.0f, which
    // is exact for every addend including signed zeros.
    __device__ float compute(State& s, float xv) const {
        typedef float v2f __attribute__((ext_vector_type(2)));
        const v2f B13 = {b[1], b[3]}, B24 = {b[2], b[4]}, A13 = {a[1], a[3]}, A24 = {a[2], a[4]};
        v2f O = {s.z[1], s.z[3]};
        v2f Wz = {s.z[2], -0.0f};
        const float y = s.z[0] + b[0] * xv;
        const v2f xx = {xv, xv}, yy = {y, y};
        const v2f E2 = (O + B13 * xx) - A13 * yy;   // (z0', z2')
        const v2f O2 = (Wz + B24 * xx) - A24 * yy;  // (z1', z3')
        s.z[0] = E2.x;
        s.z[2] = E2.y;
        s.z[1] = O2.x;
        s.z[3] = O2.y;
        return y;
    }
    __device__ Sparse sparse_begin(int64_t, int, int64_t) const { return Sparse{-1}; }
    __device__ void sparse_step(Sparse&, const State&) const {}
};

struct ArStage {
    static constexpr int NS = 2;
    static constexpr int NBRK = 0;
    static constexpr bool DENSE_OUT = true;
    struct State { float z[2]; };  // yf, ys
    struct Sparse { int rem; };
    Geom g;
    const float* xdb;  // [clips][U][C]
    float* dif;        // [clips][U][C]
    float fa, fr, sa, sr, floor_db;
    int64_t L, W, n_chunks;
    int64_t Wc;  // coarse (approximate-arithmetic) warm-up before the exact one
    __device__ int64_t len() const { return g.U; }
    __device__ int64_t brk(int) const { return 0; }
    __device__ State init(int64_t, int) const { return State{{floor_db, floor_db}}; }
    // Starting guess for the exact speculative warm-up at position u: the same
    // followers run over the preceding Wc samples in plain fp32 fma arithmetic (a
    // third of the instructions of the exact step).  Only a GUESS: exactness comes
    // from the exact warm-up that follows plus the chunk-Jacobi verification.
    struct Coarse {
        static constexpr bool DENSE_OUT = false;
        struct State { float z[2]; };
        struct Sparse { int rem; };
        Geom g;
        float fa, fr, sa, sr;
        __device__ float compute(State& s, float xv) const {
            float d0 = xv - s.z[0], d1 = xv - s.z[1];
            s.z[0] = fmaf(d0 > 0.0f ? fa : fr, d0, s.z[0]);
            s.z[1] = fmaf(d1 > 0.0f ? sa : sr, d1, s.z[1]);
            return 0.0f;
        }
        __device__ void sparse_step(Sparse&, const State&) const {}
    };
    __device__ State guess(int64_t clip, int c, int64_t u) const;
    __device__ const float* in_ptr(int64_t clip, int c, int64_t u) const { return xdb + (clip * g.U + u) * g.C + c; }
    __device__ float* out_ptr(int64_t clip, int c, int64_t u) const { return dif + (clip * g.U + u) * g.C + c; }
    __device__ float compute(State& s, float xv) const {
        s.z[0] = ofp_ar_step(xv, s.z[0], fa, fr);
        s.z[1] = ofp_ar_step(xv, s.z[1], sa, sr);
        return s.z[0] - s.z[1];
    }
    __device__ Sparse sparse_begin(int64_t, int, int64_t) const { return Sparse{-1}; }
    __device__ void sparse_step(Sparse&, const State&) const {}
};

struct MmStage {
    static constexpr int NS = 2;
    static constexpr int NBRK = 1;
    static constexpr bool DENSE_OUT = false;
    struct State { float z[2]; };  // mn, mx
    struct Sparse {                 // writes the tracker state after each MAIN block
        float* pmn;
        float* pmx;
        int rem;     // steps until the next block end (<0: warm-up region, no output)
        int stride;  // C
        int B;
    };
    Geom g;
    const float* rel;  // [clips][U][C]
    float* thr_mn;     // [clips][nb][C]
    float* thr_mx;
    float alpha_min, alpha_max, ialpha_min, ialpha_max, minmin, min0, max0;
    int64_t nb;
    int64_t L, W, n_chunks;
    __device__ int64_t len() const { return g.U; }
    __device__ int64_t brk(int) const { return g.n_wb; }
    __device__ State init(int64_t, int) const { return State{{min0, max0}}; }
    // speculative start: the min from ABOVE (+inf: the first sample sets it; it coalesces with
    // the true min at the first sample that resets the true one, which is frequent) and the max
    // from BELOW (0: coalesces at the first sample that resets the true max)
    __device__ State guess(int64_t, int, int64_t) const { return State{{__builtin_inff(), 0.f}}; }
    __device__ const float* in_ptr(int64_t clip, int c, int64_t u) const { return rel + (clip * g.U + u) * g.C + c; }
    __device__ float* out_ptr(int64_t, int, int64_t) const { return nullptr; }
    // ofp_min_step / ofp_max_step, same operations, the two EMAs as one 2-wide vector
    __device__ float compute(State& s, float xv) const {
        typedef float v2f __attribute__((ext_vector_type(2)));
        const v2f IA = {ialpha_min, ialpha_max}, AL = {alpha_min, alpha_max};
        const v2f m = {s.z[0], s.z[1]}, xx = {xv, xv};
        const v2f e = m * IA + xx * AL;
        float mn = xv < s.z[0] ? xv : e.x;
        mn = xv < minmin ? minmin : mn;
        const float mx = xv > s.z[1] ? xv : e.y;
        s.z[0] = mn;
        s.z[1] = mx;
        return 0.0f;
    }
    // max tracker only: the long part of the speculative warm-up exists for the max (it
    // coalesces only at samples that reset it); the min snaps to `minmin` at every sample
    // below it and needs only the short full-step tail of the warm-up.
    struct MaxOnly {
        static constexpr bool DENSE_OUT = false;
        struct State { float z[1]; };
        struct Sparse { int rem; };
        Geom g;
        float alpha_max, ialpha_max;
        __device__ float compute(State& s, float xv) const {
            s.z[0] = ofp_max_step(xv, s.z[0], ialpha_max, alpha_max);
            return 0.0f;
        }
        __device__ void sparse_step(Sparse&, const State&) const {}
    };
    static constexpr int64_t WARM_FULL = 4096;
    __device__ Sparse sparse_begin(int64_t clip, int c, int64_t u) const {
        Sparse sp;
        sp.stride = g.C;
        sp.B = g.B;
        int64_t m = u - g.n_wb;
        if (m < 0) {
            sp.rem = -1;
            sp.pmn = sp.pmx = nullptr;
        } else {
            int64_t j = m / g.B;
            sp.rem = (int)(g.B - 1 - (m - j * g.B));
            sp.pmn = thr_mn + (clip * nb + j) * g.C + c;
            sp.pmx = thr_mx + (clip * nb + j) * g.C + c;
        }
        return sp;
    }
    __device__ void sparse_step(Sparse& sp, const State& s) const {
        if (sp.rem == 0) {
            *sp.pmn = s.z[0];
            *sp.pmx = s.z[1];
            sp.pmn += sp.stride;
            sp.pmx += sp.stride;
            sp.rem = sp.B;
        }
        sp.rem -= 1;
    }
};

// run positions [t0, t1) of an affine span: inputs are prefetched PB steps ahead
// into registers (their addresses do not depend on the state), pointers advance
// by C floats per step.  HAS_OUT / SPARSE are decided once per span so the step
// sequence itself is straight-line code.
template <class S, bool HAS_OUT, bool SPARSE, int PB>
__device__ __forceinline__ void process_batch(const S& st, typename S::State& s, const float (&v)[PB], float*& op,
                                              typename S::Sparse& sp, int64_t stride) {
    if (SPARSE && sp.rem < PB) {  // a block ends inside this batch (once every B/PB batches)
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            st.compute(s, v[i]);
            st.sparse_step(sp, s);
        }
    } else {
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            float o = st.compute(s, v[i]);
            if (HAS_OUT) op[i * stride] = o;
        }
        if (HAS_OUT) op += PB * stride;
        if (SPARSE) sp.rem -= PB;
    }
}

template <class S, bool HAS_OUT, bool SPARSE, int PB = 16>
__device__ __forceinline__ void run_affine_impl(const S& st, typename S::State& s, const float* ip, float* op,
                                                typename S::Sparse sp, int64_t n) {
    const int64_t stride = st.g.C;
    float A[PB], Bv[PB];
    if (n >= PB) {
#pragma unroll
        for (int i = 0; i < PB; ++i) A[i] = ip[i * stride];
        ip += PB * stride;
        n -= PB;
        // two batches per iteration, ping-pong between the register sets (no copies)
        while (n >= 2 * PB) {
#pragma unroll
            for (int i = 0; i < PB; ++i) Bv[i] = ip[i * stride];
            ip += PB * stride;
            process_batch<S, HAS_OUT, SPARSE, PB>(st, s, A, op, sp, stride);
#pragma unroll
            for (int i = 0; i < PB; ++i) A[i] = ip[i * stride];
            ip += PB * stride;
            process_batch<S, HAS_OUT, SPARSE, PB>(st, s, Bv, op, sp, stride);
            n -= 2 * PB;
        }
        if (n >= PB) {
#pragma unroll
            for (int i = 0; i < PB; ++i) Bv[i] = ip[i * stride];
            ip += PB * stride;
            n -= PB;
            process_batch<S, HAS_OUT, SPARSE, PB>(st, s, A, op, sp, stride);
            process_batch<S, HAS_OUT, SPARSE, PB>(st, s, Bv, op, sp, stride);
        } else {
            process_batch<S, HAS_OUT, SPARSE, PB>(st, s, A, op, sp, stride);
        }
    }
    for (; n > 0; --n) {
        float o = st.compute(s, *ip);
        ip += stride;
        if (HAS_OUT) { *op = o; op += stride; }
        if (SPARSE) st.sparse_step(sp, s);
    }
}

__device__ __forceinline__ ArStage::State ArStage::guess(int64_t clip, int c, int64_t u) const {
    int64_t t0 = u - Wc;
    Coarse cs{g, fa, fr, sa, sr};
    Coarse::State s;
    if (t0 <= 0) {
        t0 = 0;
        s.z[0] = s.z[1] = floor_db;  // the true initial state
    } else {
        s.z[0] = s.z[1] = *in_ptr(clip, c, t0);
    }
    if (u > t0) run_affine_impl<Coarse, false, false, 64>(cs, s, in_ptr(clip, c, t0), nullptr, Coarse::Sparse{-1}, u - t0);
    return State{{s.z[0], s.z[1]}};
}

template <class S, bool OUT>
__device__ __forceinline__ void run_affine(const S& st, typename S::State& s, int64_t clip, int c,
                                           int64_t t0, int64_t t1) {
    const float* ip = st.in_ptr(clip, c, t0);
    float* op = (OUT && S::DENSE_OUT) ? st.out_ptr(clip, c, t0) : nullptr;
    typename S::Sparse sp = st.sparse_begin(clip, c, t0);
    const bool sparse = OUT && !S::DENSE_OUT && sp.rem >= 0;
    const int64_t n = t1 - t0;
    if (S::DENSE_OUT) {
        if (op) run_affine_impl<S, true, false>(st, s, ip, op, sp, n);
        else run_affine_impl<S, false, false>(st, s, ip, op, sp, n);
    } else {
        if (sparse) run_affine_impl<S, false, true>(st, s, ip, op, sp, n);
        else run_affine_impl<S, false, false>(st, s, ip, op, sp, n);
    }
}

template <class S, bool OUT>
__device__ __forceinline__ void run_span(const S& st, typename S::State& s, int64_t clip, int c,
                                         int64_t t0, int64_t t1) {
    int64_t a = t0;
#pragma unroll
    for (int i = 0; i <= S::NBRK; ++i) {
        int64_t b = t1;
        if (i < S::NBRK) b = min(max(st.brk(i), a), t1);
        if (b > a) run_affine<S, OUT>(st, s, clip, c, a, b);
        a = b;
    }
}

// speculative warm-up over [ws, start): the exact step by default
template <class S>
__device__ __forceinline__ void warm_up(const S& st, typename S::State& s, int64_t clip, int c, int64_t ws,
                                        int64_t start) {
    run_span<S, false>(st, s, clip, c, ws, start);
}
template <>
__device__ __forceinline__ void warm_up<MmStage>(const MmStage& st, MmStage::State& s, int64_t clip, int c,
                                                 int64_t ws, int64_t start) {
    const int64_t mid = max(ws, start - MmStage::WARM_FULL);
    if (mid > ws && ws > 0) {  // (ws == 0 starts from the true state: run the full step throughout)
        MmStage::MaxOnly mo{st.g, st.alpha_max, st.ialpha_max};
        MmStage::MaxOnly::State ms{{s.z[1]}};
        run_affine_impl<MmStage::MaxOnly, false, false, 64>(mo, ms, st.in_ptr(clip, c, ws), nullptr,
                                                            MmStage::MaxOnly::Sparse{-1}, mid - ws);
        s.z[1] = ms.z[0];
        ws = mid;
    }
    run_span<MmStage, false>(st, s, clip, c, ws, start);
}

// One chunk-Jacobi pass.  Thread = (clip, chunk, channel), channel fastest.
// State words are compared and stored as raw bits (NaN-safe).
template <class S>
__global__ __launch_bounds__(64) void k_jacobi(S st, int pass, int64_t n_threads,
                                               const uint32_t* __restrict__ end_prev,
                                               uint32_t* __restrict__ end_next,
                                               uint32_t* __restrict__ used, int* changed) {
    int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_threads) return;
    const int C = st.g.C;
    const int c = (int)(id % C);
    const int64_t r = id / C;
    const int64_t k = r % st.n_chunks;
    const int64_t clip = r / st.n_chunks;
    const int64_t start = k * st.L;
    const int64_t end = min(start + st.L, st.len());
    const int64_t sidx = ((clip * st.n_chunks + k) * C + c) * S::NS;
    typename S::State s;
    if (pass == 0) {
        int64_t ws = start - st.W;
        if (ws <= 0) {
            ws = 0;
            s = st.init(clip, c);
        } else {
            s = st.guess(clip, c, ws);
        }
        warm_up(st, s, clip, c, ws, start);
#pragma unroll
        for (int i = 0; i < S::NS; ++i) used[sidx + i] = ofp_f2u(s.z[i]);
        run_span<S, true>(st, s, clip, c, start, end);
#pragma unroll
        for (int i = 0; i < S::NS; ++i) end_next[sidx + i] = ofp_f2u(s.z[i]);
        return;
    }
    if (k == 0) {
#pragma unroll
        for (int i = 0; i < S::NS; ++i) end_next[sidx + i] = end_prev[sidx + i];
        return;
    }
    const int64_t pidx = sidx - (int64_t)C * S::NS;  // chunk k-1, same clip and channel
    bool same = true;
    uint32_t in[S::NS];
#pragma unroll
    for (int i = 0; i < S::NS; ++i) {
        in[i] = end_prev[pidx + i];
        same &= (in[i] == used[sidx + i]);
    }
    if (same) {
#pragma unroll
        for (int i = 0; i < S::NS; ++i) end_next[sidx + i] = end_prev[sidx + i];
        return;
    }
#pragma unroll
    for (int i = 0; i < S::NS; ++i) {
        s.z[i] = ofp_u2f(in[i]);
        used[sidx + i] = in[i];
    }
    run_span<S, true>(st, s, clip, c, start, end);
#pragma unroll
    for (int i = 0; i < S::NS; ++i) end_next[sidx + i] = ofp_f2u(s.z[i]);
    atomicAdd(changed, 1);
}

// ---- 4-lane form of the IIR stage ---------------------------------------------------
// The IIR is the stage whose speculation rarely verifies (section 3 of DESIGN.md), so
// its cost is (samples walked sequentially) x (time per step), and a lone wave issues one
// instruction per 4 cycles.  Here FOUR lanes cooperate on one chain: lane k of a quad
// owns delay z_k and the taps (b_{k+1}, a_{k+1}); z_0 and z_{k+1} arrive through DPP
// quad permutes, so a step is ~9 instructions instead of ~22.  Lane k also loads
// sample 4m+k and stores output 4m+k, i.e. one load and one store per four steps.
// The operations and their order are those of ofp_df2t4_step (the last tap adds
// -0.0f, exact for every addend), so results are bit-identical.
template <int CTRL>
__device__ __forceinline__ float qperm(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

struct Hp4 {
    float z, Bk, Ak, b0;
    bool last;  // lane 3 of the quad
};

__device__ __forceinline__ float hp4_step(float x, Hp4& h) {
    const float bx = h.Bk * x;
    const float y = qperm<0x00>(h.z) + h.b0 * x;  // z0 of the quad
    float zn = qperm<0xF9>(h.z);                   // z_{k+1}
    zn = h.last ? -0.0f : zn;
    const float t = zn + bx;
    h.z = t - h.Ak * y;
    return y;
}

template <bool HAS_OUT>
__device__ __forceinline__ void hp4_quad_block(float xr, Hp4& h, int k4, float* opl) {
    const float y0 = hp4_step(qperm<0x00>(xr), h);
    const float y1 = hp4_step(qperm<0x55>(xr), h);
    const float y2 = hp4_step(qperm<0xAA>(xr), h);
    const float y3 = hp4_step(qperm<0xFF>(xr), h);
    if (HAS_OUT) {
        float yk = k4 == 0 ? y0 : y1;
        yk = k4 == 2 ? y2 : yk;
        yk = k4 == 3 ? y3 : yk;
        *opl = yk;
    }
}

// positions [t0, t1) of an affine span (addresses affine in the position)
template <bool HAS_OUT>
__device__ __forceinline__ void hp4_affine(Hp4& h, int k4, const float* ip, float* op, int64_t stride, int64_t n) {
    constexpr int PB = 8;  // registers of prefetch = 4*PB steps
    const float* ipl = ip + k4 * stride;
    float* opl = HAS_OUT ? op + k4 * stride : nullptr;
    const int64_t s4 = 4 * stride;
    int64_t nb = n >> 2;  // blocks of four steps
    float A[PB], Bv[PB];
    if (nb >= PB) {
#pragma unroll
        for (int i = 0; i < PB; ++i) A[i] = ipl[i * s4];
        ipl += PB * s4;
        nb -= PB;
        while (nb >= 2 * PB) {
#pragma unroll
            for (int i = 0; i < PB; ++i) Bv[i] = ipl[i * s4];
            ipl += PB * s4;
#pragma unroll
            for (int i = 0; i < PB; ++i) hp4_quad_block<HAS_OUT>(A[i], h, k4, HAS_OUT ? opl + i * s4 : nullptr);
            if (HAS_OUT) opl += PB * s4;
#pragma unroll
            for (int i = 0; i < PB; ++i) A[i] = ipl[i * s4];
            ipl += PB * s4;
#pragma unroll
            for (int i = 0; i < PB; ++i) hp4_quad_block<HAS_OUT>(Bv[i], h, k4, HAS_OUT ? opl + i * s4 : nullptr);
            if (HAS_OUT) opl += PB * s4;
            nb -= 2 * PB;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) hp4_quad_block<HAS_OUT>(A[i], h, k4, HAS_OUT ? opl + i * s4 : nullptr);
        if (HAS_OUT) opl += PB * s4;
    }
    for (; nb > 0; --nb) {
        hp4_quad_block<HAS_OUT>(*ipl, h, k4, opl);
        ipl += s4;
        if (HAS_OUT) opl += s4;
    }
    // fewer than four steps left: every lane reads the same sample, lane 0 stores
    const float* ipr = ipl - k4 * stride;
    float* opr = HAS_OUT ? opl - k4 * stride : nullptr;
    for (int r = (int)(n & 3); r > 0; --r) {
        const float y = hp4_step(*ipr, h);
        ipr += stride;
        if (HAS_OUT) {
            if (k4 == 0) *opr = y;
            opr += stride;
        }
    }
}

template <bool OUT>
__device__ __forceinline__ void hp4_span(const HpStage& st, Hp4& h, int k4, int64_t clip, int c, int64_t t0,
                                         int64_t t1) {
    int64_t a = t0;
#pragma unroll
    for (int i = 0; i <= HpStage::NBRK; ++i) {
        int64_t b = t1;
        if (i < HpStage::NBRK) b = min(max(st.brk(i), a), t1);
        if (b > a) {
            const float* ip = st.in_ptr(clip, c, a);
            float* op = OUT ? st.out_ptr(clip, c, a) : nullptr;
            if (op) hp4_affine<true>(h, k4, ip, op, st.g.C, b - a);
            else hp4_affine<false>(h, k4, ip, nullptr, st.g.C, b - a);
        }
        a = b;
    }
}

// chunk-Jacobi pass, thread = (clip, chunk, channel, delay k4); same protocol as k_jacobi
__global__ __launch_bounds__(64) void k_jacobi_hp4(HpStage st, int pass, int64_t n_threads,
                                                   const uint32_t* __restrict__ end_prev,
                                                   uint32_t* __restrict__ end_next, uint32_t* __restrict__ used,
                                                   int* changed) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_threads) return;  // n_threads is a multiple of 4: quads are never split
    const int k4 = (int)(id & 3);
    const int64_t q = id >> 2;
    const int C = st.g.C;
    const int c = (int)(q % C);
    const int64_t r = q / C;
    const int64_t k = r % st.n_chunks;
    const int64_t clip = r / st.n_chunks;
    const int64_t start = k * st.L;
    const int64_t end = min(start + st.L, st.len());
    const int64_t sidx = ((clip * st.n_chunks + k) * C + c) * 4 + k4;
    Hp4 h;
    h.Bk = st.b[k4 + 1];
    h.Ak = st.a[k4 + 1];
    h.b0 = st.b[0];
    h.last = k4 == 3;
    if (pass == 0) {
        int64_t ws = max<int64_t>(start - st.W, 0);
        h.z = 0.0f;  // true initial state at 0, guess elsewhere
        hp4_span<false>(st, h, k4, clip, c, ws, start);
        used[sidx] = ofp_f2u(h.z);
        hp4_span<true>(st, h, k4, clip, c, start, end);
        end_next[sidx] = ofp_f2u(h.z);
        return;
    }
    if (k == 0) {
        end_next[sidx] = end_prev[sidx];
        return;
    }
    const uint32_t in = end_prev[sidx - (int64_t)C * 4];
    const bool diff = in != used[sidx];
    // the four lanes of a quad decide together (any delay differs -> re-run the chain)
    unsigned long long m = __ballot(diff);
    const int lane = threadIdx.x & 63;
    const bool rerun = ((m >> (lane & ~3)) & 0xfull) != 0ull;
    if (!rerun) {
        end_next[sidx] = end_prev[sidx];
        return;
    }
    h.z = ofp_u2f(in);
    used[sidx] = in;
    hp4_span<true>(st, h, k4, clip, c, start, end);
    end_next[sidx] = ofp_f2u(h.z);
    if (k4 == 0) atomicAdd(changed, 1);
}

// ---- IIR stage, multi-candidate speculation ------------------------------------------
// A single speculative warm-up coalesces with the true trajectory only at loud events
// and only with probability ~1/2 per event, so chunk-Jacobi on the IIR walks long
// stretches sequentially.  Instead every chunk start gets R candidate states, from R
// speculative runs that begin at different offsets (different rounding histories = R
// independent chances to coalesce), each continued to the chunk end:
//     U[k][r] = candidate state at the start of chunk k,  E[k][r] = state at its end.
// Resolution then walks each chain: the true start of chunk k is E[k-1][sel[k-1]];
// if some U[k][r] equals it bitwise, sel[k] = r and the walk continues for free,
// otherwise chunk k is run once from the true state (slot R) and the walk resumes.
// Finally every chunk is run from its verified start state and writes the output.
// Only bitwise-verified states are ever used, so the result is the sequential one.
constexpr int HP_MAXR = 16;

struct HpCand {
    HpStage st;
    int R;            // candidates per chunk (slots 0..R-1; slot R = exact re-run)
    int64_t delta;    // offset between candidate starts
    uint32_t* U;      // [clips][chunks][C][R+1][4]
    uint32_t* E;      // same shape
    int8_t* sel;      // [clips][chunks][C] chosen slot, -1 unknown
    uint8_t* done;    // [clips][chunks][C] output written
    uint8_t* nxt;     // [clips][chunks][C][R+1] slot of chunk k matching E[k-1][r], 255 none
    int* counters;    // [0] chains with an unresolved chunk after the last resolve
    int32_t* pos;     // [clips][C] first chunk not yet resolved (resume point of the walk)
    __device__ __host__ int64_t slot(int64_t clip, int64_t k, int c, int r) const {
        return ((((clip * st.n_chunks + k) * st.g.C + c) * (R + 1)) + r) * 4;
    }
};

// pass A: thread = (clip, chunk, channel, candidate, delay)
__global__ __launch_bounds__(64) void k_hp_candidates(HpCand a, int64_t n_threads) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_threads) return;
    const HpStage& st = a.st;
    const int k4 = (int)(id & 3);
    int64_t q = id >> 2;
    const int r = (int)(q % a.R);
    q /= a.R;
    const int C = st.g.C;
    const int c = (int)(q % C);
    q /= C;
    const int64_t k = q % st.n_chunks;
    const int64_t clip = q / st.n_chunks;
    const int64_t start = k * st.L;
    const int64_t end = min(start + st.L, st.len());
    const int64_t si = a.slot(clip, k, c, r) + k4;
    if (k4 == 0 && r == 0) {
        a.sel[(clip * st.n_chunks + k) * C + c] = (k == 0) ? 0 : -1;
        a.done[(clip * st.n_chunks + k) * C + c] = 0;
        a.U[a.slot(clip, k, c, a.R)] = 0x7fc00001u;  // slot R empty: a NaN pattern no state can equal
    }
    Hp4 h;
    h.Bk = st.b[k4 + 1];
    h.Ak = st.a[k4 + 1];
    h.b0 = st.b[0];
    h.last = k4 == 3;
    h.z = 0.0f;
    const int64_t ws = max<int64_t>(start - st.W - (int64_t)r * a.delta, 0);
    hp4_span<false>(st, h, k4, clip, c, ws, start);
    a.U[si] = ofp_f2u(h.z);
    hp4_span<false>(st, h, k4, clip, c, start, end);
    a.E[si] = ofp_f2u(h.z);
}

// pass A, one lane per candidate (packed-fp32 step of HpStage::compute): four times
// fewer lanes than the quad form, which matters here because EVERY candidate of every
// chunk runs (thousands of waves); the quad form is kept for the sparse re-runs.
__global__ __launch_bounds__(64) void k_hp_candidates1(HpCand a, int64_t n_threads) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_threads) return;
    const HpStage& st = a.st;
    int64_t q = id;
    const int C = st.g.C;
    const int c = (int)(q % C);
    q /= C;
    const int r = (int)(q % a.R);   // candidates of one chunk sit in different waves' lanes: same trip count per lane group
    q /= a.R;
    const int64_t k = q % st.n_chunks;
    const int64_t clip = q / st.n_chunks;
    const int64_t start = k * st.L;
    const int64_t end = min(start + st.L, st.len());
    const int64_t si = a.slot(clip, k, c, r);
    if (r == 0) {
        a.sel[(clip * st.n_chunks + k) * C + c] = (k == 0) ? 0 : -1;
        a.done[(clip * st.n_chunks + k) * C + c] = 0;
        a.U[a.slot(clip, k, c, a.R)] = 0x7fc00001u;  // slot R empty
    }
    HpStage::State s = st.init(clip, c);
    const int64_t ws = max<int64_t>(start - st.W - (int64_t)r * a.delta, 0);
    run_span<HpStage, false>(st, s, clip, c, ws, start);
#pragma unroll
    for (int i = 0; i < 4; ++i) a.U[si + i] = ofp_f2u(s.z[i]);
    run_span<HpStage, false>(st, s, clip, c, start, end);
#pragma unroll
    for (int i = 0; i < 4; ++i) a.E[si + i] = ofp_f2u(s.z[i]);
}

// pass B1: nxt[k][c][r_prev] for every chunk k >= 1 (parallel)
__global__ __launch_bounds__(256) void k_hp_match(HpCand a, int64_t n_items) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_items) return;
    const int R1 = a.R + 1;
    const int rp = (int)(id % R1);
    int64_t q = id / R1;
    const int C = a.st.g.C;
    const int c = (int)(q % C);
    q /= C;
    const int64_t k = q % a.st.n_chunks;
    const int64_t clip = q / a.st.n_chunks;
    uint8_t res = 255;
    if (k > 0) {
        const uint32_t* e = a.E + a.slot(clip, k - 1, c, rp);
        const uint32_t e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3];
        for (int r = 0; r < R1; ++r) {
            const uint32_t* u = a.U + a.slot(clip, k, c, r);
            if (u[0] == e0 && u[1] == e1 && u[2] == e2 && u[3] == e3) {
                res = (uint8_t)r;
                break;
            }
        }
    }
    a.nxt[id] = res;
}

// pass B2: one wave per chain walks the match table (staged through LDS in tiles)
__global__ __launch_bounds__(64) void k_hp_resolve(HpCand a) {
    __shared__ uint8_t tile[64 * (HP_MAXR + 1)];
    const int C = a.st.g.C, R1 = a.R + 1;
    const int64_t chain = blockIdx.x;  // clip*C + c
    const int c = (int)(chain % C);
    const int64_t clip = chain / C;
    const int64_t nk = a.st.n_chunks;
    const int lane = threadIdx.x;
    // resume where the previous round stopped: chunks before pos[] are resolved
    const int64_t kstart = max<int64_t>(1, a.pos[chain]);
    int cur = a.sel[(clip * nk + kstart - 1) * C + c];  // slot chosen for the previous chunk
    bool stuck = false;
    int64_t reached = nk;
    for (int64_t k0 = kstart; k0 < nk && !stuck; k0 += 64) {
        const int nblk = (int)min<int64_t>(64, nk - k0);
        __syncthreads();
        for (int i = lane; i < nblk * R1; i += 64) {
            const int bk = i / R1, r = i % R1;
            tile[i] = a.nxt[(((clip * nk + k0 + bk) * C + c) * R1) + r];
        }
        __syncthreads();
        if (lane == 0) {
            for (int bk = 0; bk < nblk; ++bk) {
                int8_t* sp = a.sel + (clip * nk + k0 + bk) * C + c;
                int s = *sp;
                if (s < 0) {
                    const uint8_t m = tile[bk * R1 + cur];
                    if (m == 255) { stuck = true; reached = k0 + bk; break; }
                    s = m;
                    *sp = (int8_t)s;
                }
                cur = s;
            }
        }
        stuck = __shfl(stuck ? 1 : 0, 0) != 0;
    }
    if (lane == 0) {
        a.pos[chain] = (int32_t)reached;
        if (stuck) atomicAdd(a.counters, 1);
    }
}

// pass C: run every chunk whose true start state is known and that has not produced its
// output yet, from that state; a chunk without a matching candidate fills slot R.
__global__ __launch_bounds__(64) void k_hp_run(HpCand a, int64_t n_threads) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_threads) return;
    const HpStage& st = a.st;
    const int k4 = (int)(id & 3);
    int64_t q = id >> 2;
    const int C = st.g.C;
    const int c = (int)(q % C);
    q /= C;
    const int64_t k = q % st.n_chunks;
    const int64_t clip = q / st.n_chunks;
    const int64_t ci = (clip * st.n_chunks + k) * C + c;
    if (a.done[ci]) return;
    int sp = 0;
    if (k > 0) {
        sp = a.sel[ci - C];
        if (sp < 0) return;  // predecessor not resolved yet
    }
    Hp4 h;
    h.Bk = st.b[k4 + 1];
    h.Ak = st.a[k4 + 1];
    h.b0 = st.b[0];
    h.last = k4 == 3;
    const uint32_t xin = (k == 0) ? 0u : a.E[a.slot(clip, k - 1, c, sp) + k4];
    h.z = ofp_u2f(xin);
    const int64_t start = k * st.L;
    const int64_t end = min(start + st.L, st.len());
    hp4_span<true>(st, h, k4, clip, c, start, end);
    const bool unresolved = a.sel[ci] < 0;
    if (unresolved) {
        a.U[a.slot(clip, k, c, a.R) + k4] = xin;
        a.E[a.slot(clip, k, c, a.R) + k4] = ofp_f2u(h.z);
    }
    // sel[ci] is NOT written here: a successor chunk running in this same launch must not
    // see a slot that is still being filled.  The next k_hp_match finds slot R (its U equals
    // the predecessor's end state by construction) and k_hp_resolve then selects it.
    if (k4 == 0) a.done[ci] = 1;  // the four lanes read done[] above, before this write (one wave, in order)
}

// ---- elementwise stages ----------------------------------------------------
// rectified dB (detection.py:747-748).  from_x: no high-pass, read the audio
// through the stream mapping; else in place on the filtered buffer.
__global__ __launch_bounds__(256) void k_rect_db(Geom g, const float* __restrict__ x,
                                                 float* __restrict__ buf, int64_t n_clips,
                                                 int from_x, float floor_db) {
    const int64_t per_clip = g.U * g.C;
    const int64_t total = n_clips * per_clip;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        float v;
        if (from_x) {
            int64_t clip = i / per_clip, rem = i % per_clip;
            int64_t u = rem / g.C;
            int c = (int)(rem % g.C);
            v = x[(clip * g.N + u_src(g, u)) * g.C + c];
        } else {
            v = buf[i];
        }
        buf[i] = ofp_rect_db(v, floor_db);
    }
}

// back to linear (detection.py:753-754), in place; main part also to the caller's rel
__global__ __launch_bounds__(256) void k_rel_linear(Geom g, float* __restrict__ buf,
                                                    float* __restrict__ rel_out, int64_t n_clips,
                                                    float floor_db) {
    const int64_t per_clip = g.U * g.C;
    const int64_t total = n_clips * per_clip;
    const int64_t warm = g.n_wb * g.C;
    const int64_t main = g.Nm * g.C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        float v = ofp_rel_linear(buf[i], floor_db);
        buf[i] = v;
        if (rel_out) {
            int64_t clip = i / per_clip, rem = i % per_clip;
            if (rem >= warm) rel_out[clip * main + (rem - warm)] = v;
        }
    }
}

// ---- block scan: first upward crossing and last sample below `off`, per
// (clip, main block, channel) -- everything of detection.py:759-770,784-790 that
// does not depend on the hysteresis state.
struct ScanArgs {
    Geom g;
    const float* rel;  // [clips][U][C]
    const float* thr_mn;
    const float* thr_mx;  // [clips][nb][C] (relative mode)
    const float* on_f;
    const float* off_f;
    const double* on_d;
    int manual;
    int64_t nb, n_clips;
    int32_t* first_cross;  // [clips][nb][C]: index or -1
    int32_t* last_below;   // [clips][nb][C]: index or -1
};

__global__ __launch_bounds__(256) void k_block_scan(ScanArgs a) {
    const int C = a.g.C, B = a.g.B;
    const int64_t total = a.n_clips * a.nb * C;
    int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int c = (int)(id % C);
    const int64_t j = (id / C) % a.nb;
    const int64_t clip = id / ((int64_t)C * a.nb);
    float on, off;
    double on0;
    if (a.manual) {
        on = a.on_f[c];
        on0 = a.on_d[c];
        off = a.off_f[c];
    } else {
        float mn = a.thr_mn[id], mx = a.thr_mx[id];
        float t1 = mx * a.on_f[c];
        on = t1 + mn;  // detection.py:763
        on0 = (double)on;
        float t2 = mx * a.off_f[c];
        off = t2 + mn;  // detection.py:787
    }
    const float* r = a.rel + (clip * a.g.U + a.g.n_wb + j * B) * C + c;
    // detection.py:769: row 0 compares prev_values (float64 copy of the previous
    // block's last row; zeros before the first main block) with the threshold
    float prev = (j == 0) ? 0.0f : r[-C];
    bool below_before = (double)prev < on0;
    int first = -1, last = -1;
    for (int t = 0; t < B; ++t) {
        float v = r[(int64_t)t * C];
        if (first < 0 && v > on && below_before) first = t;
        if (v < off) last = t;
        below_before = v < on;
    }
    a.first_cross[id] = first;
    a.last_below[id] = last;
}

// ---- hysteresis / cooldown state machine over the blocks of one clip
// (detection.py:764-797).  One 64-lane wave per clip, lanes over channels.
struct SmArgs {
    Geom g;
    int64_t nb, n_clips, cap, cooldown;
    const int32_t* first_cross;
    const int32_t* last_below;
    ofp_onset* records;  // [clips][cap]
    int64_t* counts;     // [clips]
    int32_t clip_base;   // added to record.clip
};

// Event-driven: the crossing tables are staged through LDS a tile of blocks at a
// time (coalesced, one tile ahead); while no channel is latched the wave jumps
// straight to the next block that holds a candidate crossing and advances the
// cooldown counters in closed form over the skipped blocks.
constexpr int SM_NPL = 16;  // table entries per lane per tile (TB*C <= 64*SM_NPL)

__global__ __launch_bounds__(64) void k_state_machine(SmArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int C = a.g.C, B = a.g.B;
    const int64_t clip = blockIdx.x;
    const int lane = threadIdx.x;
    const int TB = max(1, min(64, (64 * SM_NPL) / C));  // blocks per tile
    int64_t* deb = reinterpret_cast<int64_t*>(smem);          // [C]
    int32_t* onidx = reinterpret_cast<int32_t*>(deb + C);     // [C]
    int32_t* t_fc = onidx + C;                                // [TB*C]
    int32_t* t_lb = t_fc + TB * C;                            // [TB*C]
    uint8_t* state = reinterpret_cast<uint8_t*>(t_lb + TB * C);  // [C]
    uint8_t* onflag = state + C;                              // [C]
    for (int c = lane; c < C; c += 64) {
        deb[c] = 0;
        state[c] = 0;
    }
    int64_t count = 0;
    ofp_onset* rec = a.records + clip * a.cap;
    const int32_t* fc_g = a.first_cross + clip * a.nb * C;
    const int32_t* lb_g = a.last_below + clip * a.nb * C;
    int32_t rf[SM_NPL], rl[SM_NPL];
    auto load_tile = [&](int64_t j0) {
        const int64_t n = min<int64_t>(TB, a.nb - j0) * C;
#pragma unroll
        for (int i = 0; i < SM_NPL; ++i) {
            const int64_t e = lane + 64 * i;
            rf[i] = e < n ? fc_g[j0 * C + e] : -1;
            rl[i] = e < n ? lb_g[j0 * C + e] : -1;
        }
    };
    auto put_tile = [&]() {
#pragma unroll
        for (int i = 0; i < SM_NPL; ++i) {
            const int e = lane + 64 * i;
            if (e < TB * C) {
                t_fc[e] = rf[i];
                t_lb[e] = rl[i];
            }
        }
    };
    bool any_latched = false;  // wave-uniform: some channel has state == 1
    if (a.nb > 0) load_tile(0);
    for (int64_t j0 = 0; j0 < a.nb; j0 += TB) {
        __syncthreads();
        put_tile();
        __syncthreads();
        if (j0 + TB < a.nb) load_tile(j0 + TB);
        const int nblk = (int)min<int64_t>(TB, a.nb - j0);
        // candidate blocks of this tile: any channel with an upward crossing
        bool cand = false;
        if (lane < nblk)
            for (int c = 0; c < C; ++c) cand |= t_fc[lane * C + c] >= 0;
        const unsigned long long cmask = __ballot(cand);
        int bi = 0;
        while (bi < nblk) {
            if (!any_latched) {
                const unsigned long long rest = cmask >> bi;
                const int skip = rest ? __builtin_ctzll(rest) : (nblk - bi);
                if (skip > 0) {  // nothing can fire: only the cooldown counters move (:780)
                    for (int c = lane; c < C; c += 64) {
                        int64_t d = deb[c];
                        if (d > 0) deb[c] = d - (int64_t)B * min<int64_t>(skip, (d + B - 1) / B);
                    }
                    bi += skip;
                    if (bi >= nblk) break;
                }
            }
            const int64_t j = j0 + bi;
            const int32_t* fc = t_fc + bi * C;
            const int32_t* lb = t_lb + bi * C;
            int mx = 0;
            for (int c = lane; c < C; c += 64) {
                int f = fc[c];
                bool gate = !state[c] && deb[c] < 1;      // :764-768 (block-start values)
                bool on = gate && f >= 0;
                onflag[c] = on;
                int oi = on ? f : 0;                      // :774 argmax of an all-False column is 0
                onidx[c] = oi;
                mx = max(mx, oi);
            }
            // :790 on_indices.max() over ALL channels
            for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
            bool latched = false;
            for (int c0 = 0; c0 < C; c0 += 64) {
                int c = c0 + lane;
                bool on = false;
                if (c < C) {
                    on = onflag[c];
                    if (on) {                              // :778-779
                        state[c] = 1;
                        deb[c] = a.cooldown;
                    }
                    if (deb[c] > 0) deb[c] -= B;           // :780
                    if (lb[c] >= mx) state[c] = 0;         // :784-791 (any row >= mx below off)
                    latched |= state[c] != 0;
                }
                unsigned long long m = __ballot(on);
                if (on) {
                    int64_t pos = count + __popcll(m & ((1ull << lane) - 1ull));
                    if (pos < a.cap) {
                        rec[pos].clip = (int32_t)clip + a.clip_base;
                        rec[pos].channel = c;
                        rec[pos].sample = j * B + onidx[c];  // detection.py:80
                    }
                }
                count += __popcll(m);
            }
            any_latched = __ballot(latched) != 0ull;
            ++bi;
        }
    }
    if (lane == 0) a.counts[clip] = count;
}

// ---- backtracking (detection.py:800-825 == envelope_follower.c:59-85), one
// thread per onset.  The ring buffer of the reference (last N rows after writing
// the current block) is a window of the main relative envelope; rows before the
// stream start read as zero.
struct BtArgs {
    Geom g;
    const float* rel;  // [clips][U][C]
    ofp_onset* records;
    const int64_t* counts;
    int64_t cap, n_clips, N;  // N = backtrack_buffer_size
    float alpha, tol;
    int32_t clip_base;
};

__device__ __forceinline__ float bt_at(const BtArgs& a, int64_t clip, int64_t block_end, int64_t i, int c) {
    int64_t m = block_end - i;  // buffer[-i]
    if (m < 0 || i > a.N) return 0.0f;  // outside the N-row window / before the stream
    return a.rel[(clip * a.g.U + a.g.n_wb + m) * a.g.C + c];
}

__global__ __launch_bounds__(64) void k_backtrack(BtArgs a) {
    int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t clip = id / a.cap, k = id % a.cap;
    if (clip >= a.n_clips) return;
    int64_t n = min(a.counts[clip], a.cap);
    if (k >= n) return;
    ofp_onset& r = a.records[clip * a.cap + k];
    const int B = a.g.B;
    const int c = r.channel;
    int64_t j = r.sample / B;
    int64_t delta = r.sample % B;
    int64_t block_end = (j + 1) * B;
    float omba = (float)(1.0 - (double)a.alpha);
    int64_t i = B - delta;
    float cur = bt_at(a, clip, block_end, i, c);
    i += 1;
    float prev = bt_at(a, clip, block_end, i, c);
    float ps = a.alpha * prev + omba * cur;
    while (cur > ps && fabsf(ps - prev) > a.tol && (i + 1 < a.N)) {
        delta -= 1;
        i += 1;
        cur = ps;
        prev = bt_at(a, clip, block_end, i, c);
        ps = a.alpha * prev + omba * cur;
    }
    r.sample = j * B + delta;
}

// ---------------------------------------------------------------------------
// host side
struct Layout {
    Geom g;
    int64_t nb;
    int64_t hp_L, hp_W, hp_chunks, hp_delta;
    int hp_R;
    int64_t o_hp_U, o_hp_E, o_hp_sel, o_hp_done, o_hp_nxt, o_hp_pos;
    int64_t ar_L, ar_W, ar_Wc, ar_chunks;
    int64_t mm_L, mm_W, mm_chunks;
    // byte offsets
    int64_t o_xdb, o_dif, o_hp_state, o_ar_state, o_mm_state, o_thr_mn, o_thr_mx, o_first,
        o_last, o_flags, total;
};

int64_t pick(int64_t user, int64_t dflt) { return user > 0 ? user : dflt; }

Layout make_layout(const ofp_detector* d, int64_t n_clips, int64_t N, int64_t warm) {
    Layout l;
    const auto& p = d->p;
    Geom& g = l.g;
    g.C = p.n_channels;
    g.B = p.block_size;
    g.N = N;
    g.Nm = (N / g.B) * g.B;
    g.n_w = std::max<int64_t>(0, std::min(warm, N));
    g.n_wb = (g.n_w / g.B) * g.B;
    g.U = g.n_wb + g.Nm;
    g.V = g.n_w + g.Nm;
    l.nb = g.Nm / g.B;
    // longest follower time constant in samples (coefficient = 1/samples)
    float cmin = std::min(std::min(p.fast_attack, p.fast_release), std::min(p.slow_attack, p.slow_release));
    double tau = cmin > 0 ? 1.0 / cmin : 1.0;
    int64_t ar_w_default = align_up((int64_t)std::min(10.0 * tau, 4.0e6), 1024);
    l.ar_Wc = d->t.ar_coarse_warm > 0 ? d->t.ar_coarse_warm
                                     : (d->t.ar_coarse_warm < 0 ? 0 : align_up((int64_t)std::min(14.0 * tau, 8.0e6), 1024));
    l.hp_L = pick(d->t.hp_chunk, 8192);
    l.hp_W = d->t.hp_warm > 0 ? d->t.hp_warm : (d->t.hp_warm < 0 ? 0 : 49152);
    l.hp_R = (int)std::max<int64_t>(1, std::min<int64_t>(HP_MAXR, pick(d->t.hp_candidates, 8)));
    l.hp_delta = pick(d->t.hp_candidate_offset, 1021);
    l.ar_L = pick(d->t.ar_chunk, 4096);
    l.ar_W = d->t.ar_warm > 0 ? d->t.ar_warm : (d->t.ar_warm < 0 ? 0 : ar_w_default);
    l.mm_L = pick(d->t.mm_chunk, 8192);
    l.mm_W = d->t.mm_warm > 0 ? d->t.mm_warm : (d->t.mm_warm < 0 ? 0 : 49152);
    l.hp_chunks = std::max<int64_t>(1, cdiv(g.V, l.hp_L));
    l.ar_chunks = std::max<int64_t>(1, cdiv(g.U, l.ar_L));
    l.mm_chunks = std::max<int64_t>(1, cdiv(g.U, l.mm_L));
    int64_t o = 0;
    auto take = [&](int64_t bytes) {
        int64_t r = o;
        o += align_up(bytes, 256);
        return r;
    };
    const int64_t stream = n_clips * g.U * g.C * 4;
    l.o_xdb = take(stream);
    l.o_dif = take(stream);
    l.o_hp_state = take(256);
    {
        const int64_t cc = n_clips * l.hp_chunks * g.C;
        l.o_hp_U = take(cc * (l.hp_R + 1) * 16);
        l.o_hp_E = take(cc * (l.hp_R + 1) * 16);
        l.o_hp_sel = take(cc);
        l.o_hp_done = take(cc);
        l.o_hp_nxt = take(cc * (l.hp_R + 1));
        l.o_hp_pos = take(n_clips * g.C * 4);
    }
    l.o_ar_state = take(3 * n_clips * l.ar_chunks * g.C * 2 * 4);
    l.o_mm_state = take(3 * n_clips * l.mm_chunks * g.C * 2 * 4);
    l.o_thr_mn = take(n_clips * l.nb * g.C * 4);
    l.o_thr_mx = take(n_clips * l.nb * g.C * 4);
    l.o_first = take(n_clips * l.nb * g.C * 4);
    l.o_last = take(n_clips * l.nb * g.C * 4);
    l.o_flags = take(256);
    l.total = o;
    return l;
}

// chunk-Jacobi driver for one stage; returns OFP_OK or an error
template <class S>
int launch_pass(const char* name, const S& st, int pass, int64_t n_threads, const uint32_t* prev, uint32_t* next,
                uint32_t* used, int* d_changed, hipStream_t stream) {
    const unsigned grid = (unsigned)cdiv(n_threads, 64);
    hipLaunchKernelGGL(k_jacobi<S>, dim3(grid), dim3(64), 0, stream, st, pass, n_threads, prev, next, used,
                       d_changed);
    OFP_LAUNCH_CHECK(name);
    return OFP_OK;
}

template <>
int launch_pass<HpStage>(const char* name, const HpStage& st, int pass, int64_t n_threads, const uint32_t* prev,
                         uint32_t* next, uint32_t* used, int* d_changed, hipStream_t stream) {
    const int64_t n4 = n_threads * 4;  // four lanes per chain
    hipLaunchKernelGGL(k_jacobi_hp4, dim3((unsigned)cdiv(n4, 64)), dim3(64), 0, stream, st, pass, n4, prev, next,
                       used, d_changed);
    OFP_LAUNCH_CHECK(name);
    return OFP_OK;
}

// d_changed: int[OFP_MAX_GROUP]; passes are launched in groups of `group` between host
// synchronisations (a converged stage makes the surplus passes of a group no-ops).
constexpr int OFP_MAX_GROUP = 16;

template <class S>
int run_stage(const char* name, S st, int64_t n_clips, unsigned char* ws, int64_t o_state,
              int* d_changed, int group, int max_passes, hipStream_t stream, int64_t* passes,
              int64_t* repaired) {
    const int64_t n_threads = n_clips * st.n_chunks * st.g.C;
    const int64_t words = n_threads * S::NS;
    uint32_t* used = reinterpret_cast<uint32_t*>(ws + o_state);
    uint32_t* endA = used + words;
    uint32_t* endB = endA + words;
    int rc = launch_pass(name, st, 0, n_threads, (const uint32_t*)endB, endA, used, d_changed, stream);
    if (rc != OFP_OK) return rc;
    *passes = 1;
    if (st.n_chunks == 1) return OFP_OK;  // a single chunk starts from the true state: exact
    uint32_t* prev = endA;
    uint32_t* next = endB;
    group = std::max(1, std::min(group, OFP_MAX_GROUP));
    for (int pass = 1;;) {
        OFP_HIP(hipMemsetAsync(d_changed, 0, sizeof(int) * OFP_MAX_GROUP, stream));
        for (int gidx = 0; gidx < group; ++gidx, ++pass) {
            rc = launch_pass(name, st, pass, n_threads, (const uint32_t*)prev, next, used,
                             d_changed + gidx, stream);
            if (rc != OFP_OK) return rc;
            std::swap(prev, next);
            *passes += 1;
        }
        int changed[OFP_MAX_GROUP];
        OFP_HIP(hipMemcpyAsync(changed, d_changed, sizeof(int) * group, hipMemcpyDeviceToHost, stream));
        OFP_HIP(hipStreamSynchronize(stream));
        for (int gidx = 0; gidx < group; ++gidx) *repaired += changed[gidx];
        if (changed[group - 1] == 0) break;
        if (max_passes > 0 && pass > max_passes)
            return ofp::fail(OFP_ERR_NOCONVERGE, "%s: %d chunks still changing after %d passes", name,
                             changed[group - 1], pass - 1);
    }
    return OFP_OK;
}

}  // namespace

// ---------------------------------------------------------------------------
extern "C" {

int ofp_detector_create(const ofp_detector_params* p, const double* on_threshold,
                        const double* off_threshold, ofp_detector** out) {
    OFP_REQUIRE(p && on_threshold && off_threshold && out, "ofp_detector_create: NULL argument");
    OFP_REQUIRE(p->n_channels >= 1 && p->n_channels <= 4096, "n_channels %d out of range [1,4096]",
                p->n_channels);
    OFP_REQUIRE(p->block_size >= 1 && p->block_size <= (1 << 20), "block_size %d out of range",
                p->block_size);
    OFP_REQUIRE(!p->backtrack || p->backtrack_buffer_size >= p->block_size,
                "backtrack_buffer_size should be at least block_size!");
    OFP_REQUIRE(!p->hp_enabled || p->hp_a[0] != 0.0f, "hp_a[0] must be non-zero");
    ofp_detector* d = new (std::nothrow) ofp_detector();
    if (!d) return ofp::fail(OFP_ERR_INVALID, "out of host memory");
    d->p = *p;
    std::memset(&d->t, 0, sizeof(d->t));
    for (int k = 0; k < 5; ++k) {
        d->b[k] = p->hp_enabled ? p->hp_b[k] / p->hp_a[0] : 0.0f;
        d->a[k] = p->hp_enabled ? p->hp_a[k] / p->hp_a[0] : 0.0f;
    }
    d->ialpha_min = ofp_ialpha(p->alpha_min);
    d->ialpha_max = ofp_ialpha(p->alpha_max);
    const int C = p->n_channels;
    d->on.assign(on_threshold, on_threshold + C);
    d->off.assign(off_threshold, off_threshold + C);
    std::vector<float> onf(C), offf(C);
    for (int c = 0; c < C; ++c) {
        onf[c] = (float)on_threshold[c];
        offf[c] = (float)off_threshold[c];
    }
    hipError_t e = hipMalloc(&d->d_on_f, C * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&d->d_off_f, C * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&d->d_on_d, C * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(d->d_on_f, onf.data(), C * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d->d_off_f, offf.data(), C * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d->d_on_d, d->on.data(), C * sizeof(double), hipMemcpyHostToDevice);
    for (int k = 0; k < 8 && e == hipSuccess; ++k) e = hipEventCreate(&d->ev[k]);
    if (e != hipSuccess) {
        ofp_detector_destroy(d);
        return ofp::fail(OFP_ERR_HIP, "ofp_detector_create: %s", hipGetErrorString(e));
    }
    *out = d;
    return OFP_OK;
}

int ofp_detector_destroy(ofp_detector* d) {
    if (!d) return OFP_OK;
    if (d->d_on_f) (void)hipFree(d->d_on_f);
    if (d->d_off_f) (void)hipFree(d->d_off_f);
    if (d->d_on_d) (void)hipFree(d->d_on_d);
    for (int k = 0; k < 8; ++k)
        if (d->ev[k]) (void)hipEventDestroy(d->ev[k]);
    delete d;
    return OFP_OK;
}

int ofp_detector_set_tuning(ofp_detector* d, const ofp_detect_tuning* t) {
    OFP_REQUIRE(d && t, "ofp_detector_set_tuning: NULL argument");
    d->t = *t;
    return OFP_OK;
}

int64_t ofp_detect_workspace_bytes(const ofp_detector* d, int64_t n_clips, int64_t n_samples,
                                   int64_t warm) {
    if (!d || n_clips < 0 || n_samples < 0) return -1;
    return make_layout(d, n_clips, n_samples, warm).total;
}

int ofp_detect_offline(ofp_detector* d, const float* d_x, int64_t n_clips, int64_t N, int64_t warm,
                       float* d_rel, ofp_onset* d_records, int64_t cap, int64_t* d_counts,
                       void* d_ws, int64_t ws_bytes, int64_t* h_info, void* stream_) {
    OFP_REQUIRE(d && d_counts && d_ws, "ofp_detect_offline: NULL argument");
    OFP_REQUIRE(n_clips >= 1 && N >= 0 && cap >= 0, "ofp_detect_offline: bad sizes");
    OFP_REQUIRE(d_x || N == 0, "ofp_detect_offline: d_x is NULL");
    OFP_REQUIRE(d_records || cap == 0, "ofp_detect_offline: d_records is NULL");
    hipStream_t stream = (hipStream_t)stream_;
    const Layout l = make_layout(d, n_clips, N, warm);
    if (ws_bytes < l.total)
        return ofp::fail(OFP_ERR_WORKSPACE, "work space %lld < required %lld bytes", (long long)ws_bytes,
                         (long long)l.total);
    const Geom& g = l.g;
    const auto& p = d->p;
    unsigned char* ws = static_cast<unsigned char*>(d_ws);
    float* xdb = reinterpret_cast<float*>(ws + l.o_xdb);
    float* dif = reinterpret_cast<float*>(ws + l.o_dif);
    int* d_changed = reinterpret_cast<int*>(ws + l.o_flags);
    int64_t info[OFP_DETECT_INFO_LEN] = {0};
    hipEvent_t* ev = d->ev;
    OFP_HIP(hipEventRecord(ev[0], stream));
    if (l.nb == 0) {  // fewer samples than one block: nothing is processed (detection.py:74-75)
        OFP_HIP(hipMemsetAsync(d_counts, 0, n_clips * sizeof(int64_t), stream));
        OFP_HIP(hipStreamSynchronize(stream));
        if (h_info) std::memcpy(h_info, info, sizeof(info));
        return OFP_OK;
    }
    const int64_t n_elem = n_clips * g.U * g.C;
    const unsigned ew_grid = (unsigned)std::min<int64_t>(cdiv(n_elem, 256), 256 * 16);

    // --- hp + dB
    if (p.hp_enabled) {
        HpStage st;
        st.g = g;
        st.x = d_x;
        st.out = xdb;
        std::memcpy(st.b, d->b, sizeof(st.b));
        std::memcpy(st.a, d->a, sizeof(st.a));
        st.L = l.hp_L;
        st.W = l.hp_W;
        st.n_chunks = l.hp_chunks;
        HpCand hc;
        hc.st = st;
        hc.R = l.hp_R;
        hc.delta = l.hp_delta;
        hc.U = reinterpret_cast<uint32_t*>(ws + l.o_hp_U);
        hc.E = reinterpret_cast<uint32_t*>(ws + l.o_hp_E);
        hc.sel = reinterpret_cast<int8_t*>(ws + l.o_hp_sel);
        hc.done = reinterpret_cast<uint8_t*>(ws + l.o_hp_done);
        hc.nxt = reinterpret_cast<uint8_t*>(ws + l.o_hp_nxt);
        hc.counters = d_changed;
        hc.pos = reinterpret_cast<int32_t*>(ws + l.o_hp_pos);
        OFP_HIP(hipMemsetAsync(hc.pos, 0, n_clips * g.C * 4, stream));
        const int64_t chains = n_clips * g.C;
        const int64_t nA = n_clips * l.hp_chunks * g.C * hc.R * 4;
        const int64_t nM = n_clips * l.hp_chunks * g.C * (hc.R + 1);
        const int64_t nC = n_clips * l.hp_chunks * g.C * 4;
        hipLaunchKernelGGL(k_hp_candidates1, dim3((unsigned)cdiv(nA / 4, 64)), dim3(64), 0, stream, hc, nA / 4);
        OFP_LAUNCH_CHECK("k_hp_candidates1");
        for (int it = 0;; ++it) {
            OFP_HIP(hipMemsetAsync(d_changed, 0, sizeof(int), stream));
            hipLaunchKernelGGL(k_hp_match, dim3((unsigned)cdiv(nM, 256)), dim3(256), 0, stream, hc, nM);
            OFP_LAUNCH_CHECK("k_hp_match");
            hipLaunchKernelGGL(k_hp_resolve, dim3((unsigned)chains), dim3(64), 0, stream, hc);
            OFP_LAUNCH_CHECK("k_hp_resolve");
            int stuck = 0;
            OFP_HIP(hipMemcpyAsync(&stuck, d_changed, sizeof(int), hipMemcpyDeviceToHost, stream));
            hipLaunchKernelGGL(k_hp_run, dim3((unsigned)cdiv(nC, 64)), dim3(64), 0, stream, hc, nC);
            OFP_LAUNCH_CHECK("k_hp_run");
            OFP_HIP(hipStreamSynchronize(stream));
            info[0] += 1;
            info[3] += stuck;
            if (stuck == 0) break;
            if (d->t.max_passes > 0 && it >= d->t.max_passes)
                return ofp::fail(OFP_ERR_NOCONVERGE, "hp stage: %d chains still unresolved after %d rounds", stuck, it);
        }
    }
    OFP_HIP(hipEventRecord(ev[1], stream));
    hipLaunchKernelGGL(k_rect_db, dim3(ew_grid), dim3(256), 0, stream, g, d_x, xdb, n_clips,
                       p.hp_enabled ? 0 : 1, p.floor_db);
    OFP_LAUNCH_CHECK("k_rect_db");
    OFP_HIP(hipEventRecord(ev[2], stream));

    // --- followers
    {
        ArStage st;
        st.g = g;
        st.xdb = xdb;
        st.dif = dif;
        st.fa = p.fast_attack;
        st.fr = p.fast_release;
        st.sa = p.slow_attack;
        st.sr = p.slow_release;
        st.floor_db = p.floor_db;
        st.L = l.ar_L;
        st.W = l.ar_W;
        st.Wc = l.ar_Wc;
        st.n_chunks = l.ar_chunks;
        int rc = run_stage("follower stage", st, n_clips, ws, l.o_ar_state, d_changed, 1,
                           d->t.max_passes, stream, &info[1], &info[3]);
        if (rc != OFP_OK) return rc;
    }
    OFP_HIP(hipEventRecord(ev[3], stream));
    hipLaunchKernelGGL(k_rel_linear, dim3(ew_grid), dim3(256), 0, stream, g, dif, d_rel, n_clips,
                       p.floor_db);
    OFP_LAUNCH_CHECK("k_rel_linear");
    OFP_HIP(hipEventRecord(ev[4], stream));
    const float* rel = dif;

    // --- tracker (relative thresholds only; in manual mode its state is never read)
    float* thr_mn = reinterpret_cast<float*>(ws + l.o_thr_mn);
    float* thr_mx = reinterpret_cast<float*>(ws + l.o_thr_mx);
    if (!p.manual) {
        MmStage st;
        st.g = g;
        st.rel = rel;
        st.thr_mn = thr_mn;
        st.thr_mx = thr_mx;
        st.alpha_min = p.alpha_min;
        st.alpha_max = p.alpha_max;
        st.ialpha_min = d->ialpha_min;
        st.ialpha_max = d->ialpha_max;
        st.minmin = p.minmin;
        st.min0 = p.min0;
        st.max0 = p.max0;
        st.nb = l.nb;
        st.L = l.mm_L;
        st.W = l.mm_W;
        st.n_chunks = l.mm_chunks;
        int rc = run_stage("tracker stage", st, n_clips, ws, l.o_mm_state, d_changed, 1,
                           d->t.max_passes, stream, &info[2], &info[3]);
        if (rc != OFP_OK) return rc;
    }

    OFP_HIP(hipEventRecord(ev[5], stream));
    // --- crossings per block, then the hysteresis state machine
    ScanArgs sa;
    sa.g = g;
    sa.rel = rel;
    sa.thr_mn = thr_mn;
    sa.thr_mx = thr_mx;
    sa.on_f = d->d_on_f;
    sa.off_f = d->d_off_f;
    sa.on_d = d->d_on_d;
    sa.manual = p.manual;
    sa.nb = l.nb;
    sa.n_clips = n_clips;
    sa.first_cross = reinterpret_cast<int32_t*>(ws + l.o_first);
    sa.last_below = reinterpret_cast<int32_t*>(ws + l.o_last);
    {
        int64_t total = n_clips * l.nb * g.C;
        hipLaunchKernelGGL(k_block_scan, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, sa);
        OFP_LAUNCH_CHECK("k_block_scan");
    }
    SmArgs sm;
    sm.g = g;
    sm.nb = l.nb;
    sm.n_clips = n_clips;
    sm.cap = cap;
    sm.cooldown = p.cooldown;
    sm.first_cross = sa.first_cross;
    sm.last_below = sa.last_below;
    sm.records = d_records;
    sm.counts = d_counts;
    sm.clip_base = 0;
    {
        const int tb = std::max(1, std::min(64, (64 * SM_NPL) / g.C));
        size_t lds = (size_t)g.C * (8 + 4 + 1 + 1) + (size_t)2 * tb * g.C * 4 + 16;
        if (lds > 65536)
            OFP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_state_machine),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_state_machine, dim3((unsigned)n_clips), dim3(64), lds, stream, sm);
        OFP_LAUNCH_CHECK("k_state_machine");
    }
    if (p.backtrack && cap > 0) {
        BtArgs bt;
        bt.g = g;
        bt.rel = rel;
        bt.records = d_records;
        bt.counts = d_counts;
        bt.cap = cap;
        bt.n_clips = n_clips;
        bt.N = p.backtrack_buffer_size;
        bt.alpha = p.backtrack_alpha;
        bt.tol = p.backtrack_tol;
        bt.clip_base = 0;
        hipLaunchKernelGGL(k_backtrack, dim3((unsigned)cdiv(n_clips * cap, 64)), dim3(64), 0, stream, bt);
        OFP_LAUNCH_CHECK("k_backtrack");
    }
    OFP_HIP(hipEventRecord(ev[6], stream));
    OFP_HIP(hipStreamSynchronize(stream));
    // stage durations in nanoseconds (HIP events on the launch stream)
    for (int k = 0; k < 6; ++k) {
        float ms = 0.0f;
        OFP_HIP(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
        info[4 + k] = (int64_t)(ms * 1.0e6);
    }
    {
        float ms = 0.0f;
        OFP_HIP(hipEventElapsedTime(&ms, ev[0], ev[6]));
        info[10] = (int64_t)(ms * 1.0e6);
    }
    if (h_info) std::memcpy(h_info, info, sizeof(info));
    return OFP_OK;
}

}  // extern "C"
