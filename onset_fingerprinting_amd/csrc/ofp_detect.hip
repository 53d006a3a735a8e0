// Amplitude onset detector on gfx950 -- offline (batched, time-parallel) form.
//
// Reference behaviour: detection.py:19-86 (driver), :595-888 (detector),
// envelope_follower.c (followers / tracker / backtracking).  The arithmetic of
// every step is the single definition in include/ofp_math.h.
//
// The detector is a chain of sequential recurrences along time; a lone wave retires one
// dependent VALU operation per ~9 cycles, so one chain costs 11-40 ns per sample and the
// parallelism has to come from cutting time into chunks WITHOUT changing a bit:
//
//  * chunk-Jacobi (followers, tracker): pass 0 runs every chunk from a guessed state
//    after a speculative warm-up and records the state it USED at the chunk start and
//    the state it ENDED with; pass j >= 1 re-runs exactly the chunks whose used state
//    differs bitwise from the end state of the preceding chunk.  Chunk 0 starts from
//    the true state, so the fixed point IS the sequential result; the host stops when
//    a pass changes nothing.  Worst case: sequential speed, never a wrong answer.
//  * multi-candidate speculation (the IIR, whose trajectories coalesce only at loud
//    events): see the block comment above k_hp_candidates.
//
// Data layout: every internal stream is PLANAR, one contiguous fp32 series per chain
// ([clip][channel][time]), so a lane walks its own series with 16-byte loads and the
// register prefetch covers 64+ steps of memory latency.  The caller's interleaved
// [time][channel] audio is transposed once on entry (k_transpose_in) and the relative
// envelope once on exit (k_rel_out).
//
// Stages, in stream order:
//   hp : 4th-order high-pass IIR (detection.py:743-744)      state z[4]
//   db : rectified dB with floor (detection.py:747-748)      elementwise
//   ar : fast & slow attack/release followers (:751)         state (yf, ys)
//   rel: back to linear + clip (:753-754)                    elementwise
//   mm : EMA min/max tracker (:762) -> thresholds per block  state (mn, mx)
//   scan + state machine: threshold crossings, hysteresis, cooldown (:764-797)
//   backtrack (:800-825)
#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/ofp_math.h"
#include "ofp_common.h"

#include "ofp_detector.h"

// The speculative IIR kernel keeps one VALU-bound wave on most SIMDs for ~1.7 ms; with several
// steps in flight the short, dependent-latency-bound kernels of the other steps land on the same
// SIMDs.  They raise their wave priority so that their (sparse) instructions issue first and the
// long kernel fills the gaps, instead of both halving.
#define OFP_LATENCY_BOUND_KERNEL() __builtin_amdgcn_s_setprio(3)

namespace {

using ofp::align_up;
using ofp::cdiv;

typedef float v2f __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------
// stream geometry of one clip (all in samples)
struct Geom {
    int64_t N;     // samples per clip
    int64_t Nm;    // floor(N/B)*B main samples
    int64_t n_w;   // warm samples through the high-pass (detection.py:70,828-829)
    int64_t n_wb;  // warm samples through followers/tracker: full blocks only (:832-834)
    int64_t U;     // follower stream length n_wb + Nm
    int64_t V;     // high-pass stream length n_w + Nm
    int64_t Nv;    // floats per series of the planar input copy: x[0:n_w] ++ x[0:N] (rounded up to 4)
    int32_t C, B;
};
// The reference processes x[0:n_w] (init_minmax_tracker) and then restarts at x[0] with all
// filter/follower/tracker state kept.  hp stream position v -> audio index / follower index:
__host__ __device__ inline int64_t hp_src(const Geom& g, int64_t v) { return v < g.n_w ? v : v - g.n_w; }
__host__ __device__ inline int64_t hp_dst(const Geom& g, int64_t v) {
    return v < g.n_wb ? v : (v >= g.n_w ? v - g.n_w + g.n_wb : -1);
}
__host__ __device__ inline int64_t u_src(const Geom& g, int64_t u) { return u < g.n_wb ? u : u - g.n_wb; }
// The planar copy of the input holds the high-pass stream itself, x[0:n_w] ++ x[0:N], contiguous per
// series: a lane that walks the stream walks memory, whatever side of the restart it is on (with the
// restart as a jump in the source, the lanes of a wave that cross it at different steps serialised:
// the waves holding chunks 3-9 of every series ran 1.6 x longer than all others and set the launch's
// duration).  Follower-stream index u -> index in that copy:
__host__ __device__ inline int64_t u_src_planar(const Geom& g, int64_t u) { return u < g.n_wb ? u : u - g.n_wb + g.n_w; }

// ---------------------------------------------------------------------------
// walk: the one inner loop.  A lane walks n consecutive floats of ITS OWN series,
// applying a step functor; inputs arrive through 16-byte loads issued a whole batch
// (4*PB4 steps) ahead into a ping-pong pair of register sets, because their addresses do
// not depend on the recurrence.  OUT: 0 none, 1 scalar stores, 4 16-byte stores (op must
// then be congruent to ip modulo 16 bytes).  EV: the functor wants a per-step event
// (`rem` counts steps to the next event; rem < 0 disables) -- used by the tracker to
// emit its state at block ends without a per-step branch in the common batch.
template <int PB4, int OUT, bool EV, bool DEEP = false, class F>
__device__ __forceinline__ void walk(const float* ip, float* op, int64_t n, int& rem, F& f) {
    auto one = [&](float x) {
        float o = f(x);
        if (OUT) *op++ = o;
        if (EV && rem >= 0) {
            if (rem == 0) f.event();
            rem -= 1;
        }
    };
    while (n > 0 && (reinterpret_cast<uintptr_t>(ip) & 15u)) {
        one(*ip++);
        --n;
    }
    const float4* q = reinterpret_cast<const float4*>(ip);
    int64_t nb = n >> 2;
    auto batch = [&](const float4 (&v)[PB4]) {
        if (EV && rem >= 0 && rem < 4 * PB4) {  // an event falls inside this batch (rare)
#pragma unroll
            for (int i = 0; i < PB4; ++i) {
                one(v[i].x); one(v[i].y); one(v[i].z); one(v[i].w);
            }
        } else {
#pragma unroll
            for (int i = 0; i < PB4; ++i) {
                float4 o;
                o.x = f(v[i].x); o.y = f(v[i].y); o.z = f(v[i].z); o.w = f(v[i].w);
                if (OUT == 4) reinterpret_cast<float4*>(op)[i] = o;
                if (OUT == 1) { op[4 * i] = o.x; op[4 * i + 1] = o.y; op[4 * i + 2] = o.z; op[4 * i + 3] = o.w; }
            }
            if (OUT) op += 4 * PB4;
            if (EV && rem >= 0) rem -= 4 * PB4;
        }
    };
    auto load = [&](float4 (&v)[PB4]) {
#pragma unroll
        for (int i = 0; i < PB4; ++i) v[i] = q[i];
        q += PB4;
        // keep the whole group issued here: hipcc otherwise sinks each load next to its use
        // and exposes one memory latency per few steps instead of one per batch
        __builtin_amdgcn_sched_barrier(0);
    };
    auto run = [&](const float4 (&v)[PB4]) {
        batch(v);
        __builtin_amdgcn_sched_barrier(0);
    };
    if (DEEP) {
        // three register sets: two batches of loads are in flight while the third is consumed
        // (for the light steps, whose batch is shorter than one HBM round trip)
        float4 A[PB4], Bv[PB4], Cv[PB4];
        if (nb >= 2 * PB4) {
            load(A);
            load(Bv);
            nb -= 2 * PB4;
            while (nb >= 3 * PB4) {
                load(Cv); run(A);
                load(A); run(Bv);
                load(Bv); run(Cv);
                nb -= 3 * PB4;
            }
            run(A);
            run(Bv);
        }
    } else {
        float4 A[PB4], Bv[PB4];
        if (nb >= PB4) {
            load(A);
            nb -= PB4;
            while (nb >= 2 * PB4) {
                load(Bv); run(A);
                load(A); run(Bv);
                nb -= 2 * PB4;
            }
            run(A);
        }
    }
    while (nb > 0) {  // leftover whole float4s
        const float4 v = *q++;
        one(v.x); one(v.y); one(v.z); one(v.w);
        --nb;
    }
    ip = reinterpret_cast<const float*>(q);
    for (int r = (int)(n & 3); r > 0; --r) one(*ip++);
}

// ---------------------------------------------------------------------------
// walk_lines: walk<8, 4, false> for a WHOLE WAVE whose stores leave as complete 128-byte lines.  A lane's own 16-byte
// stores make every store instruction of the wave touch 64 different lines (16 B each; the lane completes its line with
// eight instructions): measured (tools/probes/storeprobe.hip, IIR-like step, one wave per SIMD) a walker that reads
// and writes its stream that way moves 2.5-2.9 TB/s, one whose wave hands each batch over through LDS and stores
// 8 complete lines per instruction (8 lanes x 16 B a line) 4.7-5.0 TB/s.  All 64 lanes of the wave call this together
// (their n may differ, 0 = nothing to do; each a multiple of 32 steps; ip / op 16-byte aligned); tile: 64 x WL_PITCH
// floats of LDS owned by the wave.
constexpr int WL_PITCH = 36;  // 32 + 4 floats: 16-byte rows, banks spread
template <class F, class D>
__device__ __forceinline__ void walk_lines_to(const float* ip, int64_t n, F& f, float* tile, D&& dst_of) {
    // dst_of(b): where the lane's batch b (32 steps) goes, or NULL for a batch without output
    const int lane = threadIdx.x & 63;
    const int64_t nb = n >> 5;
    int64_t nbmax = nb;
    for (int o = 32; o > 0; o >>= 1) nbmax = max(nbmax, __shfl_xor(nbmax, o));
    const float4* q = reinterpret_cast<const float4*>(ip);
    float4 A[8], Bv[8];
    auto load = [&](float4 (&v)[8], int64_t b) {
        if (b < nb) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = q[i];
            q += 8;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto run = [&](const float4 (&v)[8], int64_t b) {
        const bool have = b < nb;
        if (have) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float4 o;
                o.x = f(v[i].x); o.y = f(v[i].y); o.z = f(v[i].z); o.w = f(v[i].w);
                *reinterpret_cast<float4*>(&tile[lane * WL_PITCH + 4 * i]) = o;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        float* const mine = have ? dst_of(b) : nullptr;
        const unsigned long long hv = __ballot(mine != nullptr);
        const int64_t my = reinterpret_cast<int64_t>(mine);
        // store j: the lines of lanes 8 j .. 8 j + 7; this lane writes piece lane & 7 of lane 8 j + (lane >> 3)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int owner = 8 * j + (lane >> 3);
            const int lo = __shfl((int)my, owner), hi = __shfl((int)(my >> 32), owner);
            if ((hv >> owner) & 1ull) {
                float* dst = reinterpret_cast<float*>(((int64_t)hi << 32) | (uint32_t)lo) + 4 * (lane & 7);
                *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(&tile[owner * WL_PITCH + 4 * (lane & 7)]);
            }
        }
        __builtin_amdgcn_wave_barrier();  // (the tile is free for the next batch)
        __builtin_amdgcn_sched_barrier(0);
    };
    load(A, 0);
    int64_t b = 0;
    for (; b + 1 < nbmax; b += 2) {
        load(Bv, b + 1); run(A, b);
        load(A, b + 2); run(Bv, b + 1);
    }
    if (b < nbmax) run(A, b);
}
template <class F>
__device__ __forceinline__ void walk_lines(const float* ip, float* op, int64_t n, F& f, float* tile) {
    walk_lines_to(ip, n, f, tile, [op](int64_t b) { return op + 32 * b; });
}

// ---------------------------------------------------------------------------
// walk_il: the same inner loop over an INTERLEAVED series ([time][channel] rows -- the caller's audio or `rel` output): a
// lane's samples are CH floats apart and the lanes of a row's channels are neighbours, so a wave's 4-byte load covers
// whole rows (CH = 8: eight 32-byte rows of eight chunks).  No output; EV as in walk().  CH is a template parameter so
// that the PB loads of a batch are ONE address register pair and PB immediate offsets.
template <int CH, int PB, bool EV, class F>
__device__ __forceinline__ void walk_il(const float* ip, int64_t n, int& rem, F& f) {
    auto one = [&](float x) {
        f(x);
        if (EV && rem >= 0) {
            if (rem == 0) f.event();
            rem -= 1;
        }
    };
    int64_t nb = n / PB;
    int tail = (int)(n - nb * PB);
    auto load = [&](float (&v)[PB]) {
#pragma unroll
        for (int i = 0; i < PB; ++i) v[i] = ip[i * CH];
        ip += PB * CH;
        __builtin_amdgcn_sched_barrier(0);
    };
    auto run = [&](const float (&v)[PB]) {
        if (EV && rem >= 0 && rem < PB) {  // an event falls inside this batch (rare)
#pragma unroll
            for (int i = 0; i < PB; ++i) one(v[i]);
        } else {
#pragma unroll
            for (int i = 0; i < PB; ++i) f(v[i]);
            if (EV && rem >= 0) rem -= PB;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    float A[PB], Bv[PB];
    if (nb >= 1) {
        load(A);
        nb -= 1;
        while (nb >= 2) {
            load(Bv); run(A);
            load(A); run(Bv);
            nb -= 2;
        }
        if (nb == 1) {
            load(Bv); run(A);
            run(Bv);
        } else {
            run(A);
        }
    }
    for (; tail > 0; --tail) {
        one(*ip);
        ip += CH;
    }
}
// A range of a series that comes in two pieces (n0 samples from p0, then n1 from p1: the warm-up rows of `rel` live in a
// buffer of their own; the audio restarts at row 0 after the warm-up).  The pieces run through ONE instantiation of the
// walk; a range that lies in one piece takes the first trip whichever piece it is, so only the lanes that hold the
// joint make a wave take both.  (A walk that switched pieces inside its batch loads made the compiler carry a
// selected address per load: five vector instructions per sample more.)
template <int CH, int PB, bool EV, class F>
__device__ __forceinline__ void walk_il2(const float* p0, int64_t n0, const float* p1, int64_t n1, int& rem, F& f) {
    if (n0 == 0) {
        p0 = p1;
        n0 = n1;
        n1 = 0;
    }
#pragma unroll 1
    for (int piece = 0; piece < 2; ++piece) {
        const float* p = piece ? p1 : p0;
        const int64_t n = piece ? n1 : n0;
        if (n > 0) walk_il<CH, PB, EV>(p, n, rem, f);
    }
}

// walk_il with a PLANAR output (the IIR stage on the caller's interleaved audio): PB 4-byte loads per batch, the
// outputs of four steps leave as one 16-byte store (scalar steps until `op` is 16-byte aligned, and for the tail).
template <int CH, int PB, class F>
__device__ __forceinline__ void walk_il_out(const float* ip, float* op, int64_t n, F& f) {
    while (n > 0 && (reinterpret_cast<uintptr_t>(op) & 15u)) {
        *op++ = f(*ip);
        ip += CH;
        --n;
    }
    int64_t nb = n / PB;
    int tail = (int)(n - nb * PB);
    auto load = [&](float (&v)[PB]) {
#pragma unroll
        for (int i = 0; i < PB; ++i) v[i] = ip[i * CH];
        ip += PB * CH;
        __builtin_amdgcn_sched_barrier(0);
    };
    auto run = [&](const float (&v)[PB]) {
#pragma unroll
        for (int i = 0; i < PB / 4; ++i) {
            float4 o;
            o.x = f(v[4 * i]); o.y = f(v[4 * i + 1]); o.z = f(v[4 * i + 2]); o.w = f(v[4 * i + 3]);
            reinterpret_cast<float4*>(op)[i] = o;
        }
        op += PB;
        __builtin_amdgcn_sched_barrier(0);
    };
    float A[PB], Bv[PB];
    if (nb >= 1) {
        load(A);
        nb -= 1;
        while (nb >= 2) {
            load(Bv); run(A);
            load(A); run(Bv);
            nb -= 2;
        }
        if (nb == 1) {
            load(Bv); run(A);
            run(Bv);
        } else {
            run(A);
        }
    }
    for (; tail > 0; --tail) {
        *op++ = f(*ip);
        ip += CH;
    }
}

// ---------------------------------------------------------------------------
// step functors (state by value inside; include/ofp_math.h is the definition)

// ofp_df2t4_step with the same operations in the same order, arranged as 2-wide vectors so
// the compiler can use packed fp32 (v_pk_mul_f32 / v_pk_add_f32 are per-lane IEEE fp32):
// pairs (z0,z2) and (z1,z3); the last tap adds -0.0f, exact for every addend.
struct HpStep {
    float z[4];
    float b0;
    v2f B13, B24, A13, A24;
    __device__ void coeffs(const float* b, const float* a) {
        b0 = b[0];
        B13 = v2f{b[1], b[3]}; B24 = v2f{b[2], b[4]};
        A13 = v2f{a[1], a[3]}; A24 = v2f{a[2], a[4]};
    }
    __device__ float operator()(float xv) {
        v2f O = {z[1], z[3]};
        v2f Wz = {z[2], -0.0f};
        const float y = z[0] + b0 * xv;
        const v2f xx = {xv, xv}, yy = {y, y};
        const v2f E2 = (O + B13 * xx) - A13 * yy;   // (z0', z2')
        const v2f O2 = (Wz + B24 * xx) - A24 * yy;  // (z1', z3')
        z[0] = E2.x; z[2] = E2.y; z[1] = O2.x; z[3] = O2.y;
        return y;
    }
    __device__ void event() {}
};

// ofp_ar_step for both followers (two independent chains: their operations interleave)
struct ArStep {
    float yf, ys, fa, fr, sa, sr;
    __device__ float operator()(float xv) {
        yf = ofp_ar_step(xv, yf, fa, fr);
        ys = ofp_ar_step(xv, ys, sa, sr);
        return yf - ys;
    }
    __device__ void event() {}
};

// one follower alone (the two do not involve each other): half the instructions of ArStep per step
struct ArOne {
    float y, att, rel;
    __device__ float operator()(float xv) {
        y = ofp_ar_step(xv, y, att, rel);
        return 0.0f;
    }
    __device__ void event() {}
};

// the same followers in plain fp32 fma arithmetic: only a GUESS generator for the exact warm-up
struct ArCoarse {
    float yf, ys, fa, fr, sa, sr;
    __device__ float operator()(float xv) {
        const float d0 = xv - yf, d1 = xv - ys;
        yf = fmaf(d0 > 0.0f ? fa : fr, d0, yf);
        ys = fmaf(d1 > 0.0f ? sa : sr, d1, ys);
        return 0.0f;
    }
    __device__ void event() {}
};

// ofp_min_step / ofp_max_step, the two EMAs as one 2-wide vector; event() stores the
// tracker state after a MAIN block (thresholds are taken from the post-block state, :762-763)
struct MmStep {
    float mn, mx, minmin;
    v2f IA, AL;
    float* pmn;
    float* pmx;
    int stride, B;
    int* rem;
    __device__ float operator()(float xv) {
        const v2f m = {mn, mx}, xx = {xv, xv};
        const v2f e = m * IA + xx * AL;
        float a = xv < mn ? xv : e.x;
        a = xv < minmin ? minmin : a;
        const float b = xv > mx ? xv : e.y;
        mn = a;
        mx = b;
        return 0.0f;
    }
    __device__ void event() {
        *pmn = mn;
        *pmx = mx;
        pmn += stride;
        pmx += stride;
        *rem = B;  // walk() decrements right after: B-1 further steps to the next block end
    }
};

struct MaxStep {  // the max alone: the long part of the tracker's speculative warm-up
    float mx, ia, al;
    __device__ float operator()(float xv) {
        mx = ofp_max_step(xv, mx, ia, al);
        return 0.0f;
    }
    __device__ void event() {}
};

struct MinStep {  // the min alone (it does not involve the max either)
    float mn, ia, al, minmin;
    __device__ float operator()(float xv) {
        mn = ofp_min_step(xv, mn, ia, al, minmin);
        return 0.0f;
    }
    __device__ void event() {}
};

// the same one-word steps with the per-block output of the chunk pass: event() stores the state at a
// block end of the main part (thr_mn / thr_mx, [clips][nb][C])
struct MinStepEv {
    float mn, ia, al, minmin;
    float* p;
    int stride, B;
    int* rem;
    __device__ float operator()(float xv) {
        mn = ofp_min_step(xv, mn, ia, al, minmin);
        return 0.0f;
    }
    __device__ void event() {
        *p = mn;
        p += stride;
        *rem = B;  // walk() decrements right after: B-1 further steps to the next block end
    }
};
struct MaxStepEv {
    float mx, ia, al;
    float* p;
    int stride, B;
    int* rem;
    __device__ float operator()(float xv) {
        mx = ofp_max_step(xv, mx, ia, al);
        return 0.0f;
    }
    __device__ void event() {
        *p = mx;
        p += stride;
        *rem = B;
    }
};

// ---------------------------------------------------------------------------
// Zero fill of the call's counters and flags by a kernel of our own.  (hipMemsetAsync captured into a hipGraph did
// not survive replays here: after other launches had gone through the stream between two replays, the captured
// memset node filled the region with a stale 16-byte pattern -- device pointers of later kernel arguments -- instead
// of zeros; found by tests/test_gpu_detect.py::test_the_whole_call_is_one_hipgraph..., whose third replay came back
// with garbage counters.  A kernel node carries its arguments by value.)
__global__ __launch_bounds__(256) void k_zero(uint4* __restrict__ p, int64_t n16) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

__global__ __launch_bounds__(256) void k_zero_i64(int64_t* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}

// ---------------------------------------------------------------------------
// k_transpose_in: caller audio [clip][N][C] -> planar [clip][C][N]
// Tile rows are TU + 4 floats apart (16-byte aligned rows).  Fast path (full tiles, C and TU powers of two, everything
// 16-byte aligned: every shape of the BASELINE configs): 16-byte global accesses on both sides and shifts instead of
// the two integer divisions per element of the general path -- round 3's instruction counters showed this pure data
// movement issuing as many vector instructions per step (376 M wave-instructions at 16 clips) as the dB pass.
__device__ __forceinline__ void transpose_in_tile(const float* __restrict__ x, float* __restrict__ xt, int64_t N, int C, int TU,
                                                  int64_t n_w, int64_t Nv, int64_t tile_x, float* tile) {
    // tile: [C][TU+4]
    const int S = TU + 4;
    const int64_t clip = blockIdx.y;
    const int64_t t0 = tile_x * TU;
    const int nt = (int)min<int64_t>(TU, N - t0);
    const float* src = x + (clip * N + t0) * C;
    const int total = nt * C;
    const bool pow2 = (C & (C - 1)) == 0 && (TU & (TU - 1)) == 0 && TU >= 4;
    const bool fast = pow2 && nt == TU && ((N * C) & 3) == 0 && (n_w & 3) == 0 && (Nv & 3) == 0 &&
                      (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
    if (fast) {
        const int lc = 31 - __clz(C), lq = 31 - __clz(TU) - 2;  // log2 C, log2 (TU / 4)
        const float4* s4 = reinterpret_cast<const float4*>(src);
        const int total4 = total >> 2;
        for (int i4 = threadIdx.x; i4 < total4; i4 += 256) {
            const float4 v = s4[i4];
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = 4 * i4 + k;
                tile[(idx & (C - 1)) * S + (idx >> lc)] = e[k];
            }
        }
        __syncthreads();
        for (int i4 = threadIdx.x; i4 < total4; i4 += 256) {
            const int c = i4 >> lq, t = (i4 & ((1 << lq) - 1)) << 2;
            const float4 v = *reinterpret_cast<const float4*>(&tile[c * S + t]);
            float* row = xt + (clip * C + c) * Nv;
            *reinterpret_cast<float4*>(row + n_w + t0 + t) = v;
            if (t0 + t < n_w) *reinterpret_cast<float4*>(row + t0 + t) = v;  // the warm-up part of the stream (detection.py:70; n_w % 4 == 0)
        }
        return;
    }
    for (int i = threadIdx.x; i < total; i += 256) {
        const int t = i / C, c = i - t * C;
        tile[c * S + t] = src[i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < total; i += 256) {
        const int c = i / nt, t = i - c * nt;
        float* row = xt + (clip * C + c) * Nv;
        const float v = tile[c * S + t];
        row[n_w + t0 + t] = v;
        if (t0 + t < n_w) row[t0 + t] = v;  // the warm-up part of the stream (detection.py:70)
    }
}

// (several tiles per workgroup: see k_rel_out)
__global__ __launch_bounds__(256) void k_transpose_in(const float* __restrict__ x, float* __restrict__ xt,
                                                      int64_t N, int C, int TU, int64_t n_w, int64_t Nv, int64_t n_tiles) {
    extern __shared__ float tile[];
    for (int64_t tx = blockIdx.x; tx < n_tiles; tx += gridDim.x) {
        transpose_in_tile(x, xt, N, C, TU, n_w, Nv, tx, tile);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// follower stage (chunk-Jacobi).  thread = (clip, chunk, channel)
struct ArArgs {
    Geom g;
    const float* xdb;  // planar [clip*C + c][U]
    float* dif;        // planar
    float fa, fr, sa, sr, floor_db;
    int64_t L, W, Wc, Wf, n_chunks;
    int64_t S;  // span of k_ar_warm2: chunks one speculative run walks through after its warm-up
    int lines;    // the output walks of k_ar_chunk / k_ar_warm_both store complete lines (walk_lines: everything a multiple
                  // of 32 steps, the host checks)
    int through;  // k_ar_warm_both wrote the differences and the end states of the chunks it walked through (all but the
                  // last of every group of S): pass 0 of k_ar_chunk runs the others only
};
// chunk k was walked through by its group's speculative run (ArArgs::through, MmArgs::through)
__device__ __forceinline__ bool chunk_walked(int64_t k, int64_t S, int64_t n_chunks) {
    return k + 1 < min((k / S) * S + S, n_chunks);
}

// The stage is split into LEAN kernels, one walk instantiation each: measured on gfx950, the
// very same loop runs 3-4x slower inside a kernel that also carries the other (heavily
// unrolled) walk instantiations than alone (11 ns/step vs 35-50 for the max tracker), so
// code size per kernel is kept small.
//   k_ar_coarse: guess at the start of the exact warm-up, from the followers run in fp32-fma
//                arithmetic over the preceding Wc samples (a third of the instructions)
//   k_ar_warm  : exact speculative warm-up -> used[k], the state chunk k starts from
//   k_ar_chunk : pass 0 runs every chunk from used[k]; pass j >= 1 re-runs the chunks whose
//                used[k] differs bitwise from end[k-1], from end[k-1]
__global__ __launch_bounds__(64) void k_ar_coarse(ArArgs a, int64_t n_threads, uint32_t* __restrict__ used) {
    OFP_LATENCY_BOUND_KERNEL();
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_threads) return;
    const int64_t k = id % a.n_chunks;       // chunk fastest: neighbouring lanes do equal work
    const int64_t chain = id / a.n_chunks;   // clip*C + c
    const int64_t ws = max<int64_t>(k * a.L - a.W, 0);
    const int64_t sidx = (chain * a.n_chunks + k) * 2;
    const float* xs = a.xdb + chain * a.g.U;
    ArCoarse cs{a.floor_db, a.floor_db, a.fa, a.fr, a.sa, a.sr};  // true initial state (:697-702)
    if (ws > 0) {
        const int64_t t0 = max<int64_t>(ws - a.Wc, 0);
        if (t0 > 0) cs.yf = cs.ys = xs[t0];
        int norem = -1;
        walk<16, 0, false>(xs + t0, nullptr, ws - t0, norem, cs);
    }
    used[sidx] = ofp_f2u(cs.yf);
    used[sidx + 1] = ofp_f2u(cs.ys);
}

// Guess for a SYMMETRIC slow follower (attack == release, the reference's default 2205/2205):
// the recurrence is then a plain exponential average apart from its roundings, so its state at
// a chunk boundary b is  S_b = q^L S_{b-1} + P_{b-1},  P_j = sum_i c q^(L-1-i) x[jL+i],  q = 1-c,
// S_0 = floor.  k_ar_sym_local evaluates every P_j with one wave (fp64), k_ar_sym_combine chains
// them per channel: microseconds instead of a sequential pass over Wc samples per chunk.  The
// exact warm-up of chunk k starts at boundary k - W/L (make_layout makes W a multiple of L); the
// fast follower starts from the floor, and the warm-up is long enough to forget that.
// Only a guess: the chunk passes verify every state bit for bit.
__global__ __launch_bounds__(64) void k_ar_sym_local(ArArgs a, int64_t n_waves, double* __restrict__ P) {
    OFP_LATENCY_BOUND_KERNEL();
    const int64_t id = blockIdx.x;  // one wave per (chain, chunk)
    if (id >= n_waves) return;
    const int lane = threadIdx.x;
    const int64_t j = id % a.n_chunks;
    const int64_t chain = id / a.n_chunks;
    const int64_t base = j * a.L;
    const int64_t len = min(a.L, a.g.U - base);
    const float* xs = a.xdb + chain * a.g.U + base;
    const double c = (double)a.sa, q = 1.0 - c;
    double q64 = 1.0, w = c;  // q^64 and this lane's first weight c q^lane
    for (int i = 0; i < 64; ++i) {
        q64 *= q;
        if (i < lane) w *= q;
    }
    double acc0 = 0.0, acc1 = 0.0;
    const double q128 = q64 * q64;
    double w1 = w * q64;
    int64_t t = lane;
    for (; t + 64 < len; t += 128) {  // sample len-1-t carries weight c q^t
        acc0 += w * (double)xs[len - 1 - t];
        acc1 += w1 * (double)xs[len - 1 - t - 64];
        w *= q128;
        w1 *= q128;
    }
    if (t < len) acc0 += w * (double)xs[len - 1 - t];
    double acc = acc0 + acc1;
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) P[id] = acc;
}

__global__ __launch_bounds__(64) void k_ar_sym_combine(ArArgs a, const double* __restrict__ P,
                                                       uint32_t* __restrict__ used) {
    OFP_LATENCY_BOUND_KERNEL();
    __shared__ double sp[1024];
    __shared__ float ss[1024];
    const int64_t chain = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t shift = a.W / a.L;  // boundaries between a chunk and the start of its warm-up
    const double qL = exp((double)a.L * log(1.0 - (double)a.sa));
    double S = (double)a.floor_db;  // state at boundary b0
    for (int64_t b0 = 0; b0 < a.n_chunks; b0 += 1024) {
        const int nb = (int)min<int64_t>(1024, a.n_chunks - b0);
        __syncthreads();
        for (int i = lane; i < nb; i += 64) sp[i] = P[chain * a.n_chunks + b0 + i];
        __syncthreads();
        if (lane == 0) {
            for (int i = 0; i < nb; ++i) {
                ss[i] = (float)S;  // state at boundary b0 + i
                S = qL * S + sp[i];
            }
        }
        __syncthreads();
        for (int i = lane; i < nb; i += 64) {
            const int64_t k = b0 + i + shift;  // the chunk whose warm-up starts at this boundary
            if (k < a.n_chunks) {
                const int64_t sidx = (chain * a.n_chunks + k) * 2;
                used[sidx] = ofp_f2u(a.floor_db);
                used[sidx + 1] = ofp_f2u(b0 + i > 0 ? ss[i] : a.floor_db);
            }
        }
    }
    for (int64_t k = lane; k < min(shift, a.n_chunks); k += 64) {  // warm-up starts at sample 0: true state
        const int64_t sidx = (chain * a.n_chunks + k) * 2;
        used[sidx] = ofp_f2u(a.floor_db);
        used[sidx + 1] = ofp_f2u(a.floor_db);
    }
}

__global__ __launch_bounds__(64) void k_ar_warm(ArArgs a, int64_t n_threads, uint32_t* __restrict__ used) {
    OFP_LATENCY_BOUND_KERNEL();
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_threads) return;
    const int64_t k = id % a.n_chunks;
    const int64_t chain = id / a.n_chunks;
    const int64_t start = k * a.L;
    const int64_t ws = max<int64_t>(start - a.W, 0);
    const int64_t sidx = (chain * a.n_chunks + k) * 2;
    const float* xs = a.xdb + chain * a.g.U;
    ArStep s{ofp_u2f(used[sidx]), ofp_u2f(used[sidx + 1]), a.fa, a.fr, a.sa, a.sr};
    int norem = -1;
    walk<8, 0, false>(xs + ws, nullptr, start - ws, norem, s);
    used[sidx] = ofp_f2u(s.yf);
    used[sidx + 1] = ofp_f2u(s.ys);
}

// The same warm-up with the followers as separate lanes (symmetric-guess path): the slow one walks
// its whole window from the closed-form guess (first half of the grid) WHILE the fast one, which
// starts from the floor and forgets it within ~24 of its own time constants, walks the last Wf
// samples only (second half), each with half the instructions per step.
__global__ __launch_bounds__(64) void k_ar_warm2(ArArgs a, int64_t n_threads, uint32_t* __restrict__ used) {
    // SPAN: one lane = one run that warms up before chunk k0 = g*S and then walks on through the S-1
    // following chunks, leaving the state it passes every boundary with as that chunk's start guess (a
    // guess with a LONGER warm-up than its own run would have had).  Per chunk the launch reads and
    // steps (W + (S-1) L) / S samples instead of W: the overlapping windows of neighbouring chunks were
    // most of this stage's memory traffic.  n_threads = chains * ceil(n_chunks / S).
    OFP_LATENCY_BOUND_KERNEL();
    const int64_t half = (n_threads + 63) / 64;
    const bool fast = blockIdx.x >= half;
    const int64_t id = ((int64_t)blockIdx.x - (fast ? half : 0)) * blockDim.x + threadIdx.x;
    if (id >= n_threads) return;
    const int64_t n_groups = cdiv(a.n_chunks, a.S);
    const int64_t g = id % n_groups;
    const int64_t chain = id / n_groups;
    const int64_t k0 = g * a.S;
    const int64_t start = k0 * a.L;
    const float* xs = a.xdb + chain * a.g.U;
    int norem = -1;
    if (!fast) {
        const int64_t ws = max<int64_t>(start - a.W, 0);
        ArOne s{ofp_u2f(used[(chain * a.n_chunks + k0) * 2 + 1]), a.sa, a.sr};  // closed-form guess at the run's start
        walk<8, 0, false>(xs + ws, nullptr, start - ws, norem, s);
        used[(chain * a.n_chunks + k0) * 2 + 1] = ofp_f2u(s.y);
        for (int64_t k = k0 + 1; k < min(k0 + a.S, a.n_chunks); ++k) {
            walk<8, 0, false>(xs + (k - 1) * a.L, nullptr, a.L, norem, s);
            used[(chain * a.n_chunks + k) * 2 + 1] = ofp_f2u(s.y);
        }
    } else {
        const int64_t ws = max<int64_t>(start - min(a.W, a.Wf), 0);
        ArOne s{a.floor_db, a.fa, a.fr};  // the true state at sample 0 (:697-702), a guess elsewhere
        walk<8, 0, false>(xs + ws, nullptr, start - ws, norem, s);
        used[(chain * a.n_chunks + k0) * 2] = ofp_f2u(s.y);
        for (int64_t k = k0 + 1; k < min(k0 + a.S, a.n_chunks); ++k) {
            walk<8, 0, false>(xs + (k - 1) * a.L, nullptr, a.L, norem, s);
            used[(chain * a.n_chunks + k) * 2] = ofp_f2u(s.y);
        }
    }
}

// k_ar_warm2 with both followers in ONE lane (throughput setting): a saturated launch pays for bytes, not for the
// length of a lane's dependent chain, and the two lanes of a chunk read the same samples at different times (the
// slow one W ahead of the chunk, the fast one Wf), so each line of the stream came from HBM twice.  Here the
// fast follower simply starts with the slow one (a longer warm-up than it needs).
template <bool LINES>
__global__ __launch_bounds__(64) void k_ar_warm_both(ArArgs a, int64_t n_threads, uint32_t* __restrict__ used,
                                                     uint32_t* __restrict__ end0) {
    // a.through: the chunks the run walks through after its warm-up ARE their pass 0 -- it writes their differences and
    // their end states (end0 = the end array pass 0 writes) exactly as k_ar_chunk would from the same start state: one
    // pass over the dB stream less for those chunks.  (A wrong start guess is repaired by the verifying passes as before.)
    // LINES (with a.through): those outputs leave as complete lines (walk_lines: the whole wave stays together).
    OFP_LATENCY_BOUND_KERNEL();
    __shared__ float tile[LINES ? 64 * WL_PITCH : 1];
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = id < n_threads;
    if (!LINES && !live) return;
    const int64_t n_groups = cdiv(a.n_chunks, a.S);
    const int64_t g = live ? id % n_groups : 0;
    const int64_t chain = live ? id / n_groups : 0;
    const int64_t k0 = g * a.S;
    const int64_t start = k0 * a.L;
    const float* xs = a.xdb + chain * a.g.U;
    int norem = -1;
    const int64_t ws = max<int64_t>(start - a.W, 0);
    const int64_t s0 = (chain * a.n_chunks + k0) * 2;
    ArStep s{a.floor_db, live ? ofp_u2f(used[s0 + 1]) : 0.0f, a.fa, a.fr, a.sa, a.sr};  // floor / closed-form guess at the run's start
    if (live) {
        walk<8, 0, false>(xs + ws, nullptr, start - ws, norem, s);
        used[s0] = ofp_f2u(s.yf);
        used[s0 + 1] = ofp_f2u(s.ys);
    }
    float* os = a.dif + chain * a.g.U;
    if (LINES) {
        for (int64_t i = 1; i < a.S; ++i) {  // (the same trip count for the whole wave)
            const int64_t k = k0 + i;
            const bool on = live && k < a.n_chunks;
            walk_lines(xs + (on ? (k - 1) * a.L : 0), os + (on ? (k - 1) * a.L : 0), on ? a.L : 0, s, tile);
            if (on) {
                used[(chain * a.n_chunks + k) * 2] = ofp_f2u(s.yf);
                used[(chain * a.n_chunks + k) * 2 + 1] = ofp_f2u(s.ys);
                end0[(chain * a.n_chunks + k - 1) * 2] = ofp_f2u(s.yf);
                end0[(chain * a.n_chunks + k - 1) * 2 + 1] = ofp_f2u(s.ys);
            }
        }
        return;
    }
    for (int64_t k = k0 + 1; k < min(k0 + a.S, a.n_chunks); ++k) {
        if (a.through) walk<8, 4, false>(xs + (k - 1) * a.L, os + (k - 1) * a.L, a.L, norem, s);
        else walk<8, 0, false>(xs + (k - 1) * a.L, nullptr, a.L, norem, s);
        used[(chain * a.n_chunks + k) * 2] = ofp_f2u(s.yf);
        used[(chain * a.n_chunks + k) * 2 + 1] = ofp_f2u(s.ys);
        if (a.through) {
            end0[(chain * a.n_chunks + k - 1) * 2] = ofp_f2u(s.yf);
            end0[(chain * a.n_chunks + k - 1) * 2 + 1] = ofp_f2u(s.ys);
        }
    }
}

template <bool LINES>
__global__ __launch_bounds__(64) void k_ar_chunk(ArArgs a, int pass, int64_t n_threads,
                                                 const uint32_t* __restrict__ end_prev,
                                                 uint32_t* __restrict__ end_next, uint32_t* __restrict__ used,
                                                 int* changed, const int* gate) {
    OFP_LATENCY_BOUND_KERNEL();
    // gate: the change counter of the pass this one follows (passes enqueued ahead of the host's knowledge): zero =
    // that pass repaired nothing, so it left both end arrays identical and this pass has nothing to do
    if (gate && *gate == 0) return;
    __shared__ float tile[LINES ? 64 * WL_PITCH : 1];
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool act = id < n_threads;
    const int64_t k = act ? id % a.n_chunks : 0;
    const int64_t chain = act ? id / a.n_chunks : 0;
    const int64_t start = k * a.L;
    const int64_t end = min(start + a.L, a.g.U);
    const int64_t sidx = (chain * a.n_chunks + k) * 2;
    if (act && pass == 0 && a.through && chunk_walked(k, a.S, a.n_chunks)) act = false;  // k_ar_warm_both has been through it
    uint32_t i0 = 0, i1 = 0;
    if (act) {
        i0 = used[sidx];
        i1 = used[sidx + 1];
        if (pass > 0) {
            if (k == 0) {
                end_next[sidx] = end_prev[sidx];
                end_next[sidx + 1] = end_prev[sidx + 1];
                act = false;
            } else {
                const uint32_t p0 = end_prev[sidx - 2], p1 = end_prev[sidx - 1];
                if (p0 == i0 && p1 == i1) {
                    end_next[sidx] = end_prev[sidx];
                    end_next[sidx + 1] = end_prev[sidx + 1];
                    act = false;
                } else {
                    i0 = p0;
                    i1 = p1;
                    used[sidx] = i0;
                    used[sidx + 1] = i1;
                    atomicAdd(changed, 1);
                }
            }
        }
    }
    ArStep s{ofp_u2f(i0), ofp_u2f(i1), a.fa, a.fr, a.sa, a.sr};
    if (LINES) {  // the whole wave stays together (walk_lines); lanes with nothing to do lend their stores
        if (!__any(act)) return;
        walk_lines(a.xdb + chain * a.g.U + start, a.dif + chain * a.g.U + start, act ? end - start : 0, s, tile);
    } else {
        if (!act) return;
        int norem = -1;
        walk<8, 4, false>(a.xdb + chain * a.g.U + start, a.dif + chain * a.g.U + start, end - start, norem, s);
    }
    if (act) {
        end_next[sidx] = ofp_f2u(s.yf);
        end_next[sidx + 1] = ofp_f2u(s.ys);
    }
}

// ---------------------------------------------------------------------------
// tracker stage (chunk-Jacobi)
struct MmArgs {
    Geom g;
    const float* rel;  // planar [clip*C + c][U]
    const float* rel_il;    // the same values as the caller gets them, interleaved [clip][Nm][C] (main part) ...
    const float* rel_warm;  // ... and [clip][n_wb][C] (warm-up part): what the *_il kernels read instead
    float* thr_mn;     // [clips][nb][C] tracker state after each MAIN block
    float* thr_mx;
    float alpha_min, alpha_max, ialpha_min, ialpha_max, minmin, min0, max0;
    int64_t nb, L, W, n_chunks;
    uint8_t* dirty;  // [chains][n_chunks] chunk must be run again although its start matches (k_mm_sweep)
    int64_t n_chains;
    int64_t S;  // span of k_mm_warm2: chunks one speculative run walks through after its warm-up
    int through;  // k_mm_warm_il left the per-block outputs and the end states of the chunks it walked through (all but
                  // the last of every group of S): pass 0 of k_mm_chunk_il runs the others only
};
// The min and the max are two independent recurrences and are treated as such: every launch of
// this stage has one lane per (chain, chunk) for the max (first half of the grid) and one for the
// min (second half), each walking a one-word step (11-15 ns instead of 25-35 ns for the pair).
//   k_mm_warm2: speculative warm-up.  The max is guessed from BELOW (0): it coalesces with the true
//               max at the first sample that resets the true one, which needs the long window (and
//               is exact for chunks whose window reaches the stream start).  The min is guessed
//               from ABOVE (+inf): it coalesces at the first sample that resets the true min,
//               which is frequent, so its window is the last MM_WARM_FULL samples only.
//   k_mm_chunk: chunk with threshold output at the block ends of the MAIN part; pass 0 from
//               used[k], pass j >= 1 re-runs chunks whose used[k] != end[k-1] bitwise.
constexpr int64_t MM_WARM_FULL = 12288;

// Speculative warm-up: the two recurrences are independent, so the max runs its whole window alone
// (first half of the grid) WHILE the min runs its short one alone (second half), each at the speed
// of a one-word step.
__global__ __launch_bounds__(64) void k_mm_warm2(MmArgs a, int64_t n_threads, uint32_t* __restrict__ used) {
    // SPAN as in k_ar_warm2: a run warms up before chunk g*S and walks on through the S-1 following
    // chunks, leaving a start guess at every boundary it passes.  n_threads = chains * ceil(n_chunks / S).
    OFP_LATENCY_BOUND_KERNEL();
    const int64_t half = (n_threads + 63) / 64;  // blocks per half
    const bool is_min = blockIdx.x >= half;
    const int64_t id = ((int64_t)blockIdx.x - (is_min ? half : 0)) * blockDim.x + threadIdx.x;
    if (id >= n_threads) return;
    const int64_t n_groups = cdiv(a.n_chunks, a.S);
    const int64_t g = id % n_groups;
    const int64_t chain = id / n_groups;
    const int64_t k0 = g * a.S;
    const int64_t start = k0 * a.L;
    const float* rs = a.rel + chain * a.g.U;
    int norem = -1;
    if (!is_min) {
        const int64_t ws = max<int64_t>(start - a.W, 0);
        MaxStep mo{ws > 0 ? 0.0f : a.max0, a.ialpha_max, a.alpha_max};  // guessed from below
        walk<16, 0, false>(rs + ws, nullptr, start - ws, norem, mo);
        used[(chain * a.n_chunks + k0) * 2 + 1] = ofp_f2u(mo.mx);
        for (int64_t k = k0 + 1; k < min(k0 + a.S, a.n_chunks); ++k) {
            walk<16, 0, false>(rs + (k - 1) * a.L, nullptr, a.L, norem, mo);
            used[(chain * a.n_chunks + k) * 2 + 1] = ofp_f2u(mo.mx);
        }
    } else {
        const int64_t ws = max(max<int64_t>(start - a.W, 0), start - MM_WARM_FULL);
        MinStep mi{ws > 0 ? __builtin_inff() : a.min0, a.ialpha_min, a.alpha_min, a.minmin};  // from above
        walk<16, 0, false>(rs + ws, nullptr, start - ws, norem, mi);
        used[(chain * a.n_chunks + k0) * 2] = ofp_f2u(mi.mn);
        for (int64_t k = k0 + 1; k < min(k0 + a.S, a.n_chunks); ++k) {
            walk<16, 0, false>(rs + (k - 1) * a.L, nullptr, a.L, norem, mi);
            used[(chain * a.n_chunks + k) * 2] = ofp_f2u(mi.mn);
        }
    }
}

__global__ __launch_bounds__(64) void k_mm_chunk(MmArgs a, int pass, int64_t n_threads,
                                                 const uint32_t* __restrict__ end_prev,
                                                 uint32_t* __restrict__ end_next, uint32_t* __restrict__ used,
                                                 int* changed, const int* gate) {
    OFP_LATENCY_BOUND_KERNEL();
    if (gate && *gate == 0) return;  // (see k_ar_chunk)
    // launched with 2 * ceil(nt / 64) workgroups: the first half carries the max, the second the min
    const int64_t nt = a.n_chains * a.n_chunks;
    const int64_t half = (nt + 63) / 64;
    const bool is_min = blockIdx.x >= half;
    const int64_t id = ((int64_t)blockIdx.x - (is_min ? half : 0)) * blockDim.x + threadIdx.x;
    if (id >= nt) return;
    const int64_t k = id % a.n_chunks;
    const int64_t chain = id / a.n_chunks;
    const int64_t start = k * a.L;
    const int64_t end = min(start + a.L, a.g.U);
    const int64_t sidx = (chain * a.n_chunks + k) * 2 + (is_min ? 0 : 1);  // this lane's word of the pair
    uint32_t i0 = used[sidx];
    if (pass > 0) {
        if (k == 0) {
            end_next[sidx] = end_prev[sidx];
            return;
        }
        const uint32_t p0 = end_prev[sidx - 2];
        const bool redo = !is_min && a.dirty[id] != 0;  // a light pass changed this chunk's starting max
        if (p0 == i0 && !redo) {
            end_next[sidx] = end_prev[sidx];
            return;
        }
        if (!is_min) a.dirty[id] = 0;
        i0 = p0;
        used[sidx] = i0;
        atomicAdd(changed, 1);
    }
    const int C = a.g.C;
    const int64_t clip = chain / C;
    const int c = (int)(chain % C);
    // steps to the first block end of the MAIN part at or after `start` (events are disabled
    // while rem < 0, i.e. never here: the count simply runs through the warm part)
    int rem;
    const int64_t m = start - a.g.n_wb;  // position in the main part (negative: still warm part)
    int64_t j = 0;
    if (m >= 0) {
        j = m / a.g.B;
        rem = (int)(a.g.B - 1 - (m - j * a.g.B));
    } else {
        rem = (int)min<int64_t>(-m + a.g.B - 1, 0x7fffffff);
    }
    const float* rs = a.rel + chain * a.g.U + start;
    const int64_t oi = (clip * a.nb + j) * C + c;
    if (is_min) {
        MinStepEv s{ofp_u2f(i0), a.ialpha_min, a.alpha_min, a.minmin, a.thr_mn + oi, C, a.g.B, &rem};
        walk<8, 0, true>(rs, nullptr, end - start, rem, s);
        end_next[sidx] = ofp_f2u(s.mn);
    } else {
        MaxStepEv s{ofp_u2f(i0), a.ialpha_max, a.alpha_max, a.thr_mx + oi, C, a.g.B, &rem};
        walk<8, 0, true>(rs, nullptr, end - start, rem, s);
        end_next[sidx] = ofp_f2u(s.mx);
    }
}

// The same two kernels with the min and the max in ONE lane (MmStep, the 2-wide packed step): the throughput
// setting.  A saturated launch pays for the bytes it moves, and the separate lanes read every line of the stream
// twice, at different times; the pair costs no more instructions than the two one-word walks together.
__global__ __launch_bounds__(64) void k_mm_warm_both(MmArgs a, int64_t n_threads, uint32_t* __restrict__ used) {
    OFP_LATENCY_BOUND_KERNEL();
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_threads) return;
    const int64_t n_groups = cdiv(a.n_chunks, a.S);
    const int64_t g = id % n_groups;
    const int64_t chain = id / n_groups;
    const int64_t k0 = g * a.S;
    const int64_t start = k0 * a.L;
    const float* rs = a.rel + chain * a.g.U;
    int norem = -1;
    const int64_t ws = max<int64_t>(start - a.W, 0);
    // max guessed from below, min from above (see k_mm_warm2); exact when the window reaches the stream start
    MmStep s{ws > 0 ? __builtin_inff() : a.min0, ws > 0 ? 0.0f : a.max0, a.minmin, v2f{a.ialpha_min, a.ialpha_max},
             v2f{a.alpha_min, a.alpha_max}, nullptr, nullptr, 0, 0, nullptr};
    walk<8, 0, false>(rs + ws, nullptr, start - ws, norem, s);
    used[(chain * a.n_chunks + k0) * 2] = ofp_f2u(s.mn);
    used[(chain * a.n_chunks + k0) * 2 + 1] = ofp_f2u(s.mx);
    for (int64_t k = k0 + 1; k < min(k0 + a.S, a.n_chunks); ++k) {
        walk<8, 0, false>(rs + (k - 1) * a.L, nullptr, a.L, norem, s);
        used[(chain * a.n_chunks + k) * 2] = ofp_f2u(s.mn);
        used[(chain * a.n_chunks + k) * 2 + 1] = ofp_f2u(s.mx);
    }
}

__global__ __launch_bounds__(64) void k_mm_chunk_both(MmArgs a, int pass, int64_t n_threads,
                                                      const uint32_t* __restrict__ end_prev,
                                                      uint32_t* __restrict__ end_next, uint32_t* __restrict__ used,
                                                      int* changed, const int* gate) {
    OFP_LATENCY_BOUND_KERNEL();
    if (gate && *gate == 0) return;  // (see k_ar_chunk)
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_threads) return;
    const int64_t k = id % a.n_chunks;
    const int64_t chain = id / a.n_chunks;
    const int64_t start = k * a.L;
    const int64_t end = min(start + a.L, a.g.U);
    const int64_t sidx = (chain * a.n_chunks + k) * 2;
    uint32_t i0 = used[sidx], i1 = used[sidx + 1];
    if (pass > 0) {
        if (k == 0) {
            end_next[sidx] = end_prev[sidx];
            end_next[sidx + 1] = end_prev[sidx + 1];
            return;
        }
        const uint32_t p0 = end_prev[sidx - 2], p1 = end_prev[sidx - 1];
        const bool redo = a.dirty[id] != 0;  // a light pass changed this chunk's starting max
        if (p0 == i0 && p1 == i1 && !redo) {
            end_next[sidx] = end_prev[sidx];
            end_next[sidx + 1] = end_prev[sidx + 1];
            return;
        }
        a.dirty[id] = 0;
        i0 = p0;
        i1 = p1;
        used[sidx] = i0;
        used[sidx + 1] = i1;
        atomicAdd(changed, 1);
    }
    const int C = a.g.C;
    const int64_t clip = chain / C;
    const int c = (int)(chain % C);
    int rem;
    const int64_t m = start - a.g.n_wb;  // position in the main part (negative: still warm part)
    int64_t j = 0;
    if (m >= 0) {
        j = m / a.g.B;
        rem = (int)(a.g.B - 1 - (m - j * a.g.B));
    } else {
        rem = (int)min<int64_t>(-m + a.g.B - 1, 0x7fffffff);
    }
    const int64_t oi = (clip * a.nb + j) * C + c;
    MmStep s{ofp_u2f(i0), ofp_u2f(i1), a.minmin, v2f{a.ialpha_min, a.ialpha_max}, v2f{a.alpha_min, a.alpha_max},
             a.thr_mn + oi, a.thr_mx + oi, C, a.g.B, &rem};
    walk<8, 0, true>(a.rel + chain * a.g.U + start, nullptr, end - start, rem, s);
    end_next[sidx] = ofp_f2u(s.mn);
    end_next[sidx + 1] = ofp_f2u(s.mx);
}

// The max coalesces only where the true max is reset, so a stretch without a reset (a loud hit
// followed by seconds of quieter ones) leaves every chunk start inside it wrong after the
// speculative warm-up, and the repair then advances ONE chunk per pass.  The max recurrence does
// not involve the min, so once a pass has found something to repair the host switches to LIGHT
// passes: the same verify-and-repair rule applied to the max alone (11 ns per step instead of the
// full step, eight passes per host round trip), which marks the chunks it corrects dirty; the full
// passes that follow run those again and remain the verification.
__global__ __launch_bounds__(64) void k_mm_maxpass(MmArgs a, int64_t n_threads, const uint32_t* __restrict__ end_prev,
                                                   uint32_t* __restrict__ end_next, uint32_t* __restrict__ used,
                                                   int* changed, const int* gate) {
    OFP_LATENCY_BOUND_KERNEL();
    if (gate && *gate == 0) return;  // (see k_ar_chunk: a group of light passes runs only after a full pass that repaired)
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_threads) return;
    const int64_t k = id % a.n_chunks;
    const int64_t chain = id / a.n_chunks;
    const int64_t sidx = (chain * a.n_chunks + k) * 2;
    end_next[sidx] = end_prev[sidx];  // the min is not touched here
    if (k == 0 || used[sidx + 1] == end_prev[sidx - 1]) {
        end_next[sidx + 1] = end_prev[sidx + 1];
        return;
    }
    const uint32_t t = end_prev[sidx - 1];
    used[sidx + 1] = t;
    MaxStep mo{ofp_u2f(t), a.ialpha_max, a.alpha_max};
    int norem = -1;
    const int64_t start = k * a.L;
    walk<16, 0, false>(a.rel + chain * a.g.U + start, nullptr, min(start + a.L, a.g.U) - start, norem, mo);
    end_next[sidx + 1] = ofp_f2u(mo.mx);
    a.dirty[id] = 1;
    atomicAdd(changed, 1);
}

// ---- the tracker on the INTERLEAVED envelope (throughput setting, C = 4 or 8 with a `rel` output): the three kernels
// above with lane = (clip, chunk, channel), channel fastest, reading the rows the caller gets (walk_il) -- the planar
// copy of `rel` is then never written (one pass over the stream less per call).
struct IlSrc {
    const float* p0;
    int64_t n0;
    const float* p1;
    int64_t n1;
};
// stream positions [u0, u1) of (clip, c): warm-up rows first, main rows after n_wb
__device__ __forceinline__ IlSrc mm_il_range(const MmArgs& a, int64_t clip, int c, int64_t u0, int64_t u1) {
    const int64_t nw = a.g.n_wb, C = a.g.C;
    IlSrc s;
    s.n0 = max<int64_t>(min(u1, nw) - u0, 0);
    s.p0 = a.rel_warm + (clip * nw + min(u0, nw)) * C + c;
    s.n1 = (u1 - u0) - s.n0;
    s.p1 = a.rel_il + (clip * a.g.Nm + (max(u0, nw) - nw)) * C + c;
    return s;
}

template <int CH>
__global__ __launch_bounds__(64) void k_mm_warm_il(MmArgs a, int64_t n_threads, uint32_t* __restrict__ used,
                                                   uint32_t* __restrict__ end0) {
    // a.through: the chunks the run walks through after its warm-up ARE their pass 0 -- the run stores the tracker state
    // at their block ends and their end states (end0 = the end array pass 0 writes), exactly as k_mm_chunk_il would from
    // the same start state; one pass over the stream less.  A start guess that turns out wrong is repaired by the
    // verifying passes as before (they compare used[k] with the end of chunk k - 1, whoever wrote it).
    OFP_LATENCY_BOUND_KERNEL();
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_threads) return;
    const int64_t n_groups = cdiv(a.n_chunks, a.S);
    const int c = (int)(id % CH);
    const int64_t r = id / CH;
    const int64_t g = r % n_groups;
    const int64_t clip = r / n_groups;
    const int64_t chain = clip * CH + c;
    const int64_t k0 = g * a.S;
    const int64_t start = k0 * a.L;
    int rem = -1;  // (no events during the warm-up)
    const int64_t ws = max<int64_t>(start - a.W, 0);
    MmStep s{ws > 0 ? __builtin_inff() : a.min0, ws > 0 ? 0.0f : a.max0, a.minmin, v2f{a.ialpha_min, a.ialpha_max},
             v2f{a.alpha_min, a.alpha_max}, nullptr, nullptr, CH, a.g.B, &rem};
    IlSrc q = mm_il_range(a, clip, c, ws, start);
    walk_il2<CH, 32, true>(q.p0, q.n0, q.p1, q.n1, rem, s);
    used[(chain * a.n_chunks + k0) * 2] = ofp_f2u(s.mn);
    used[(chain * a.n_chunks + k0) * 2 + 1] = ofp_f2u(s.mx);
    if (a.through) {  // block-end output from chunk k0 on (as in k_mm_chunk_il)
        const int64_t m = start - a.g.n_wb;
        int64_t j = 0;
        if (m >= 0) {
            j = m / a.g.B;
            rem = (int)(a.g.B - 1 - (m - j * a.g.B));
        } else {
            rem = (int)min<int64_t>(-m + a.g.B - 1, 0x7fffffff);
        }
        const int64_t oi = (clip * a.nb + j) * CH + c;
        s.pmn = a.thr_mn + oi;
        s.pmx = a.thr_mx + oi;
    }
    for (int64_t k = k0 + 1; k < min(k0 + a.S, a.n_chunks); ++k) {
        q = mm_il_range(a, clip, c, (k - 1) * a.L, k * a.L);
        walk_il2<CH, 32, true>(q.p0, q.n0, q.p1, q.n1, rem, s);
        used[(chain * a.n_chunks + k) * 2] = ofp_f2u(s.mn);
        used[(chain * a.n_chunks + k) * 2 + 1] = ofp_f2u(s.mx);
        if (a.through) {
            end0[(chain * a.n_chunks + k - 1) * 2] = ofp_f2u(s.mn);
            end0[(chain * a.n_chunks + k - 1) * 2 + 1] = ofp_f2u(s.mx);
        }
    }
}

template <int CH>
__global__ __launch_bounds__(64) void k_mm_chunk_il(MmArgs a, int pass, int64_t n_threads,
                                                    const uint32_t* __restrict__ end_prev,
                                                    uint32_t* __restrict__ end_next, uint32_t* __restrict__ used,
                                                    int* changed, const int* gate) {
    OFP_LATENCY_BOUND_KERNEL();
    if (gate && *gate == 0) return;  // (see k_ar_chunk)
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_threads) return;
    const int c = (int)(id % CH);
    const int64_t r = id / CH;
    const int64_t k = r % a.n_chunks;
    const int64_t clip = r / a.n_chunks;
    const int64_t chain = clip * CH + c;
    const int64_t did = chain * a.n_chunks + k;  // (the index the planar kernels call `id`)
    const int64_t start = k * a.L;
    const int64_t end = min(start + a.L, a.g.U);
    const int64_t sidx = did * 2;
    if (pass == 0 && a.through && chunk_walked(k, a.S, a.n_chunks)) return;  // k_mm_warm_il has been through it
    uint32_t i0 = used[sidx], i1 = used[sidx + 1];
    if (pass > 0) {
        if (k == 0) {
            end_next[sidx] = end_prev[sidx];
            end_next[sidx + 1] = end_prev[sidx + 1];
            return;
        }
        const uint32_t p0 = end_prev[sidx - 2], p1 = end_prev[sidx - 1];
        const bool redo = a.dirty[did] != 0;  // a light pass changed this chunk's starting max
        if (p0 == i0 && p1 == i1 && !redo) {
            end_next[sidx] = end_prev[sidx];
            end_next[sidx + 1] = end_prev[sidx + 1];
            return;
        }
        a.dirty[did] = 0;
        i0 = p0;
        i1 = p1;
        used[sidx] = i0;
        used[sidx + 1] = i1;
        atomicAdd(changed, 1);
    }
    int rem;
    const int64_t m = start - a.g.n_wb;  // position in the main part (negative: still warm part)
    int64_t j = 0;
    if (m >= 0) {
        j = m / a.g.B;
        rem = (int)(a.g.B - 1 - (m - j * a.g.B));
    } else {
        rem = (int)min<int64_t>(-m + a.g.B - 1, 0x7fffffff);
    }
    const int64_t oi = (clip * a.nb + j) * CH + c;
    MmStep s{ofp_u2f(i0), ofp_u2f(i1), a.minmin, v2f{a.ialpha_min, a.ialpha_max}, v2f{a.alpha_min, a.alpha_max},
             a.thr_mn + oi, a.thr_mx + oi, CH, a.g.B, &rem};
    const IlSrc q = mm_il_range(a, clip, c, start, end);
    walk_il2<CH, 32, true>(q.p0, q.n0, q.p1, q.n1, rem, s);
    end_next[sidx] = ofp_f2u(s.mn);
    end_next[sidx + 1] = ofp_f2u(s.mx);
}

template <int CH>
__global__ __launch_bounds__(64) void k_mm_maxpass_il(MmArgs a, int64_t n_threads, const uint32_t* __restrict__ end_prev,
                                                      uint32_t* __restrict__ end_next, uint32_t* __restrict__ used,
                                                      int* changed, const int* gate) {
    OFP_LATENCY_BOUND_KERNEL();
    if (gate && *gate == 0) return;
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_threads) return;
    const int c = (int)(id % CH);
    const int64_t r = id / CH;
    const int64_t k = r % a.n_chunks;
    const int64_t clip = r / a.n_chunks;
    const int64_t did = (clip * CH + c) * a.n_chunks + k;
    const int64_t sidx = did * 2;
    end_next[sidx] = end_prev[sidx];  // the min is not touched here
    if (k == 0 || used[sidx + 1] == end_prev[sidx - 1]) {
        end_next[sidx + 1] = end_prev[sidx + 1];
        return;
    }
    const uint32_t t = end_prev[sidx - 1];
    used[sidx + 1] = t;
    MaxStep mo{ofp_u2f(t), a.ialpha_max, a.alpha_max};
    int norem = -1;
    const int64_t start = k * a.L;
    const IlSrc q = mm_il_range(a, clip, c, start, min(start + a.L, a.g.U));
    walk_il2<CH, 32, false>(q.p0, q.n0, q.p1, q.n1, norem, mo);
    end_next[sidx + 1] = ofp_f2u(mo.mx);
    a.dirty[did] = 1;
    atomicAdd(changed, 1);
}

// ---------------------------------------------------------------------------
// IIR stage
struct HpArgs {
    Geom g;
    const float* xt;  // planar audio [clip*C + c][N]
    const float* x_il;  // or the caller's interleaved audio [clip][N][C] (kernels instantiated with CH = C: no planar copy)
    float* out;       // planar [clip*C + c][U]
    float b[5], a[5];
    int64_t L, W, n_chunks;
};

// positions [t0, t1) of the hp stream; the stream<->memory mapping is affine between the
// breaks n_wb and n_w (detection.py:828-834: the tail of the warm-up passes the filter only)
template <bool OUT, int CH = 0, class F>
__device__ __forceinline__ void hp_span(const HpArgs& a, F& f, int64_t chain, int64_t t0, int64_t t1) {
    int norem = -1;
    if constexpr (CH > 0) {
        // the caller's interleaved audio: stream position v is row v (v < n_w) or row v - n_w of the clip
        constexpr int CHD = CH > 0 ? CH : 1;
        const int64_t clip = chain / CHD;
        const float* xs = a.x_il + clip * a.g.N * CH + (chain - clip * CH);
        if (!OUT) {
            if (t1 > t0) {
                const int64_t n0 = max<int64_t>(min(t1, a.g.n_w) - t0, 0);
                walk_il2<CH, 32, false>(xs + t0 * CH, n0, xs + (max(t0, a.g.n_w) - a.g.n_w) * CH, (t1 - t0) - n0, norem, f);
            }
            return;
        }
        float* os = a.out + chain * a.g.U;
        int64_t p = t0;
#pragma unroll 1
        for (int i = 0; i < 3; ++i) {  // (pieces as below; each lies on one side of the restart at n_w)
            const int64_t br = i == 0 ? a.g.n_wb : (i == 1 ? a.g.n_w : t1);
            const int64_t e = min(max(br, p), t1);
            if (e > p) {
                const int64_t u = hp_dst(a.g, p);
                const float* src = xs + hp_src(a.g, p) * CH;
                if (u >= 0) walk_il_out<CH, 32>(src, os + u, e - p, f);
                else walk_il<CH, 32, false>(src, e - p, norem, f);
            }
            p = e;
        }
        return;
    }
    const float* xs = a.xt + chain * a.g.Nv;  // the stream itself (see u_src_planar)
    if (!OUT) {
        if (t1 > t0) walk<8, 0, false>(xs + t0, nullptr, t1 - t0, norem, f);
        return;
    }
    float* os = a.out + chain * a.g.U;
    int64_t p = t0;
#pragma unroll 1
    for (int i = 0; i < 3; ++i) {  // the OUTPUT has the breaks: rows n_wb..n_w pass the filter only
        const int64_t br = i == 0 ? a.g.n_wb : (i == 1 ? a.g.n_w : t1);
        const int64_t e = min(max(br, p), t1);
        if (e > p) {
            const int64_t u = hp_dst(a.g, p);
            // 16-byte stores whenever the output is congruent to the input modulo 16 bytes (it is for every block
            // size and warm-up length that are multiples of 4 samples): a quarter of the store instructions
            if (u >= 0 && ((reinterpret_cast<uintptr_t>(xs + p) ^ reinterpret_cast<uintptr_t>(os + u)) & 15u) == 0)
                walk<8, 4, false>(xs + p, os + u, e - p, norem, f);
            else if (u >= 0) walk<8, 1, false>(xs + p, os + u, e - p, norem, f);
            else walk<8, 0, false>(xs + p, nullptr, e - p, norem, f);
        }
        p = e;
    }
}

constexpr int HP_MAXR = 16;

struct HpCand {
    HpArgs st;
    int R;            // candidates per chunk (slots 0..R-1; slot R = exact re-run)
    int64_t delta;    // offset between candidate starts
    int span;         // chunks one run walks through (R % span == 0)
    uint32_t* U;      // [clips][chunks][C][R+1][4]
    uint32_t* E;      // same shape
    int8_t* sel;      // [clips][chunks][C] chosen slot, -1 unknown
    uint8_t* done;    // [clips][chunks][C][S] output of sub-chunk written
    int S;            // sub-chunks per chunk: every candidate also records its state at the S-1
                      // inner boundaries, so a chunk whose start matched a candidate is run by S
                      // lanes at once (its candidate's trajectory is exact all the way through)
    uint32_t* M;      // [clips][chunks][C][R][S-1][4] those states
    uint8_t* nxt;     // [clips][C][chunks][R+1] slot of chunk k matching E[k-1][r], 255 none
    uint8_t* guessed; // [clips][chunks][C] sel[] is an unverified plurality guess (see k_hp_resolve)
    int8_t* gs;       // [clips][C][chunks] plurality slot of each chunk's end states (k_hp_plurality), -1 none
    int8_t* ran;      // [clips][chunks][C] slot whose trajectory produced the chunk's output (R: whole run
                      // from the true start state): lets a re-walk keep outputs whose start did not change
    int* counters;    // [0] chains with an unresolved chunk after the last resolve, [1] chains with
                      // unverified guesses
    const int* prev;  // the counters of the preceding round, or NULL: both zero = the stage had converged,
                      // this round (enqueued ahead of the host's check) has nothing to do
    int32_t* pos;     // [clips][C] first chunk not yet resolved (resume point of the walk)
    int8_t* mrg;      // [clips][chunks][C] early stop of a whole run: the candidate whose recorded trajectory the run from
                      // the true start state joined at a sub-chunk boundary (the sub-chunks after it are written from
                      // that candidate's inner states by the next round's lanes), -1 none
    int early;        // whole runs stop at the first sub-chunk boundary where they have joined a candidate
    unsigned long long* probe;  // diagnostics (OFP_HP_PROBE): per wave {start, end (s_memtime), HW_ID, XCC_ID}; else NULL
    __device__ __host__ int64_t slot(int64_t clip, int64_t k, int c, int r) const {
        return ((((clip * st.n_chunks + k) * st.g.C + c) * (R + 1)) + r) * 4;
    }
    __device__ __host__ int64_t mslot(int64_t clip, int64_t k, int c, int r, int sb) const {
        return (((((clip * st.n_chunks + k) * st.g.C + c) * R + r) * (S - 1)) + sb) * 4;
    }
};


// pass A: thread = (clip, channel, chunk, candidate), candidate fastest.
// SPREAD: the launch has no more workgroups than the chip has CUs (C2: 704 waves for 1024 SIMDs).
// Each wave is a long dependent chain, so two waves sharing a SIMD while other SIMDs idle cost
// ~30 % (measured: the same launch took 2.3 or 3.1 ms depending on where the dispatcher happened
// to put the waves).  A 4-wave workgroup that claims 100 of the CU's 160 KiB of LDS is alone on
// its CU, one wave per SIMD -- placement becomes deterministic.
constexpr int HP_CAND_THREADS = 256;
template <bool SPREAD>
__global__ __launch_bounds__(HP_CAND_THREADS) void k_hp_candidates(HpCand a, int64_t n_threads) {
    if (SPREAD) {
        __shared__ char claim[100 * 1024];
        if (n_threads < 0) claim[threadIdx.x] = 1;  // never taken: keeps the allocation alive
    }
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long t_probe = 0;
    if (a.probe) t_probe = __builtin_amdgcn_s_memtime();
    if (id >= n_threads) return;
    const HpArgs& st = a.st;
    // One lane = one RUN: it warms up over W samples before chunk j and then walks through `span`
    // consecutive chunks, leaving a candidate at every boundary it passes (slot i*Rm + r for chunk
    // j + i).  span = 1: every candidate has its own warm-up (shortest kernel).  span = 4: the 16
    // candidates of a chunk come from 4 runs started at each of the 4 preceding boundaries, the
    // warm-up is shared by 4 chunks and the launch does 2.7x less work in a quarter of the waves,
    // at the price of 3 more chunks of latency -- the throughput setting when several steps are
    // in flight.  Lane order: the Rm runs of one boundary are neighbours (they read the same
    // window shifted by r*delta and share cache lines).
    const int Rm = a.R / a.span;
    int64_t q = id;
    const int r = (int)(q % Rm);
    q /= Rm;
    const int64_t j = q % st.n_chunks;
    const int64_t chain = q / st.n_chunks;  // clip*C + c
    const int C = st.g.C;
    const int64_t clip = chain / C;
    const int c = (int)(chain % C);
    if (r == 0) {
        const int64_t ci = (clip * st.n_chunks + j) * C + c;
        a.sel[ci] = (j == 0) ? 0 : -1;
        for (int sb = 0; sb < a.S; ++sb) a.done[ci * a.S + sb] = 0;
        a.guessed[ci] = 0;
        a.ran[ci] = -2;
        a.mrg[ci] = -1;
        a.U[a.slot(clip, j, c, a.R)] = 0x7fc00001u;  // slot R empty: a NaN pattern no state can equal
        // slots of runs that would have started before the stream: never match, never agree
        for (int rr = (int)min<int64_t>((j + 1) * Rm, a.R); rr < a.R; ++rr) {
            a.U[a.slot(clip, j, c, rr)] = 0x7fc00001u;
            a.E[a.slot(clip, j, c, rr)] = 0x7fc00100u + (uint32_t)rr;
        }
    }
    HpStep s;
    s.coeffs(st.b, st.a);
    s.z[0] = s.z[1] = s.z[2] = s.z[3] = 0.0f;  // true state at 0 (detection.py:497), guess elsewhere
    // The candidates of a chunk must reach the loud events of the window with DIFFERENT rounding
    // histories (independent chances to coalesce with the true trajectory): staggered starts,
    // delta samples apart (default 8).  delta == 0 (tuning < 0): all start at the same sample from
    // slightly different states (r/1024 in z[0]; the difference decays below an ulp within ~200
    // samples, the histories stay distinct).
    if (a.delta == 0) s.z[0] = (float)r * 0.0009765625f;
    const int64_t ws = max<int64_t>(j * st.L - st.W - (int64_t)r * a.delta, 0);
    hp_span<false>(st, s, chain, ws, j * st.L);
    const int64_t Ls = st.L / a.S;
#pragma unroll 1
    for (int i = 0; i < a.span; ++i) {
        const int64_t k = j + i;
        if (k >= st.n_chunks) break;
        const int slot = i * Rm + r;
        const int64_t start = k * st.L;
        const int64_t end = min(start + st.L, st.g.V);
        const int64_t si = a.slot(clip, k, c, slot);
#pragma unroll
        for (int w = 0; w < 4; ++w) a.U[si + w] = ofp_f2u(s.z[w]);
#pragma unroll 1
        for (int sb = 0; sb < a.S; ++sb) {
            const int64_t t0 = min(start + sb * Ls, end);
            const int64_t t1 = sb == a.S - 1 ? end : min(start + (sb + 1) * Ls, end);
            hp_span<false>(st, s, chain, t0, t1);
            uint32_t* dst = sb == a.S - 1 ? a.E + si : a.M + a.mslot(clip, k, c, slot, sb);
#pragma unroll
            for (int w = 0; w < 4; ++w) dst[w] = ofp_f2u(s.z[w]);
        }
    }
    if (a.probe && (threadIdx.x & 63) == 0) {
        unsigned long long* q = a.probe + 4 * (id >> 6);
        q[0] = t_probe;
        q[1] = __builtin_amdgcn_s_memtime();
        q[2] = (unsigned)__builtin_amdgcn_s_getreg(63492);  // HW_REG_HW_ID: wave, simd, cu, sh, se ...
        q[3] = (unsigned)__builtin_amdgcn_s_getreg(63508);  // HW_REG_XCC_ID
    }
}

// ---- the same candidates in STAGES with duplicates removed between them (throughput setting).
// The R speculative runs of a chunk merge with EACH OTHER long before the chunk starts (measured on C2,
// tools/cand_merge_stats.py: of 8 runs 4.8 are distinct after 8 192 steps, 3.0 after 16 384, 1.6 at the chunk
// start), and two runs in the same state at the same position stay the same run for ever.  The warm-up is
// therefore cut into segments; after each, k_hp_dedupe keeps one run per distinct state of a group (= the
// candidates of one chunk) with the set of candidate slots it stands for, and compacts the work list, so that the
// next segment's launch has a lane per DISTINCT run only.  The last stage walks the chunk itself and writes the
// U / M / E records of every slot its run stands for: everything downstream sees exactly what k_hp_candidates
// (span 1) would have written, with a third of the steps.
struct HpRuns {
    int32_t* grp;    // [n] group = chain * n_chunks + j
    uint32_t* mask;  // [n] candidate slots of the group this run stands for
    float4* z;       // [n] filter state at the common position of the stage
};

// stage 0: lane = (chain, chunk j, candidate r), r fastest: from its staggered start to window offset p1.
// CH > 0 (all staged kernels and k_hp_run): the lanes read the caller's INTERLEAVED audio (hp_span<.., CH>), so the
// channel is the fastest lane coordinate -- lane = (clip, chunk j, candidate r, channel) here, and the groups are
// numbered (clip, j, channel) so that the work lists keep the channels of a chunk side by side: a wave's 4-byte loads
// then cover whole 4 CH-byte rows.
template <int CH>
__global__ __launch_bounds__(HP_CAND_THREADS) void k_hp_seg0(HpCand a, HpRuns out, int32_t* __restrict__ off,
                                                              int32_t* __restrict__ cnt, int64_t p1, int64_t n_threads) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_threads) return;
    const HpArgs& st = a.st;
    int64_t q = id;
    int r, c;
    int64_t grp, j, chain, clip;
    const int C = st.g.C;
    constexpr int CHD = CH > 0 ? CH : 1;  // (divisor in the branches that are dead for CH = 0)
    if (CH > 0) {
        c = (int)(q % CHD);
        q /= CHD;
        r = (int)(q % a.R);
        q /= a.R;
        j = q % st.n_chunks;
        clip = q / st.n_chunks;
        chain = clip * CH + c;
        grp = (clip * st.n_chunks + j) * CH + c;
    } else {
        r = (int)(q % a.R);
        q /= a.R;
        grp = q;
        j = q % st.n_chunks;
        chain = q / st.n_chunks;
        clip = chain / C;
        c = (int)(chain % C);
    }
    if (r == 0) {
        const int64_t ci = (clip * st.n_chunks + j) * C + c;
        a.sel[ci] = (j == 0) ? 0 : -1;
        for (int sb = 0; sb < a.S; ++sb) a.done[ci * a.S + sb] = 0;
        a.guessed[ci] = 0;
        a.ran[ci] = -2;
        a.mrg[ci] = -1;
        a.U[a.slot(clip, j, c, a.R)] = 0x7fc00001u;  // slot R empty: a NaN pattern no state can equal
        off[grp] = (int32_t)(grp * a.R);
        cnt[grp] = a.R;
    }
    HpStep s;
    s.coeffs(st.b, st.a);
    s.z[0] = s.z[1] = s.z[2] = s.z[3] = 0.0f;
    if (a.delta == 0) s.z[0] = (float)r * 0.0009765625f;
    const int64_t w0 = j * st.L - st.W;
    const int64_t ws = max<int64_t>(w0 - (int64_t)r * a.delta, 0);
    hp_span<false, CH>(st, s, chain, ws, max<int64_t>(w0 + p1, 0));
    const int64_t li = CH > 0 ? grp * a.R + r : id;  // the runs of a group are neighbours in the list (k_hp_dedupe)
    out.grp[li] = (int32_t)grp;
    out.mask[li] = 1u << r;
    out.z[li] = make_float4(s.z[0], s.z[1], s.z[2], s.z[3]);
}

// between stages: 16 lanes per group, lane i holds run i of the group.  A run whose state equals (bitwise) that of
// an earlier run of the group is dropped and its slot set joins that run's; the survivors are appended to the next
// work list (a wave reserves its range with one atomic; a group's runs stay neighbours: they read the same samples)
__global__ __launch_bounds__(256) void k_hp_dedupe(HpRuns in, HpRuns out, int32_t* __restrict__ off,
                                                   int32_t* __restrict__ cnt, int64_t n_groups, int* __restrict__ n_out) {
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int i = threadIdx.x & 15;
    const int lane = threadIdx.x & 63;
    const bool live = g < n_groups;
    const int o = live ? off[g] : 0, k = live ? cnt[g] : 0;
    const bool have = i < k;
    uint32_t z0 = 0, z1 = 0, z2 = 0, z3 = 0, mk = 0;
    if (have) {
        const float4 v = in.z[o + i];
        z0 = ofp_f2u(v.x), z1 = ofp_f2u(v.y), z2 = ofp_f2u(v.z), z3 = ofp_f2u(v.w);
        mk = in.mask[o + i];
    }
    int leader = i;
    uint32_t all = 0;
    for (int q = 0; q < 16; ++q) {  // (uniform trip count: every lane takes part in the shuffles)
        const bool same = __shfl(z0, q, 16) == z0 && __shfl(z1, q, 16) == z1 && __shfl(z2, q, 16) == z2 &&
                          __shfl(z3, q, 16) == z3 && q < k;
        const uint32_t mq = __shfl(mk, q, 16);
        if (same && have) {
            all |= mq;
            leader = min(leader, q);
        }
    }
    const bool keep = have && leader == i;
    const unsigned long long kept = __ballot(keep);
    const int before = __popcll(kept & ((1ull << lane) - 1));  // survivors before me in the wave: groups stay in order
    const int total = __popcll(kept);
    int base = 0;
    if (lane == 0 && total > 0) base = atomicAdd(n_out, total);
    base = __shfl(base, 0);
    const int gfirst = __popcll(kept & ((1ull << (lane & 48)) - 1));
    if (live && i == 0) {
        off[g] = base + gfirst;
        cnt[g] = __popcll((kept >> (lane & 48)) & 0xffffull);
    }
    if (keep) {
        out.grp[base + before] = (int32_t)g;
        out.mask[base + before] = all;
        out.z[base + before] = make_float4(ofp_u2f(z0), ofp_u2f(z1), ofp_u2f(z2), ofp_u2f(z3));
    }
}

// a middle stage: lane = one distinct run, window offsets [p0, p1)
template <int CH>
__global__ __launch_bounds__(HP_CAND_THREADS) void k_hp_seg(HpCand a, HpRuns runs, const int* __restrict__ n_runs,
                                                             int64_t p0, int64_t p1) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= *n_runs) return;
    const HpArgs& st = a.st;
    const int64_t grp = runs.grp[id];
    constexpr int CHD = CH > 0 ? CH : 1;  // (divisor in the branch that is dead for CH = 0)
    const int64_t j = CH > 0 ? (grp / CHD) % st.n_chunks : grp % st.n_chunks;
    const int64_t chain = CH > 0 ? grp / (CHD * st.n_chunks) * CH + grp % CHD : grp / st.n_chunks;
    HpStep s;
    s.coeffs(st.b, st.a);
    const float4 v = runs.z[id];
    s.z[0] = v.x; s.z[1] = v.y; s.z[2] = v.z; s.z[3] = v.w;
    const int64_t w0 = j * st.L - st.W;
    hp_span<false, CH>(st, s, chain, max<int64_t>(w0 + p0, 0), max<int64_t>(w0 + p1, 0));
    runs.z[id] = make_float4(s.z[0], s.z[1], s.z[2], s.z[3]);
}

// the last stage: the chunk itself, with the records of every slot the run stands for
template <int CH>
__global__ __launch_bounds__(HP_CAND_THREADS) void k_hp_seg_chunk(HpCand a, HpRuns runs, const int* __restrict__ n_runs) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= *n_runs) return;
    const HpArgs& st = a.st;
    const int64_t grp = runs.grp[id];
    const uint32_t mask = runs.mask[id];
    constexpr int CHD = CH > 0 ? CH : 1;  // (divisor in the branch that is dead for CH = 0)
    const int64_t k = CH > 0 ? (grp / CHD) % st.n_chunks : grp % st.n_chunks;
    const int64_t chain = CH > 0 ? grp / (CHD * st.n_chunks) * CH + grp % CHD : grp / st.n_chunks;
    const int C = st.g.C;
    const int64_t clip = chain / C;
    const int c = (int)(chain % C);
    HpStep s;
    s.coeffs(st.b, st.a);
    const float4 v = runs.z[id];
    s.z[0] = v.x; s.z[1] = v.y; s.z[2] = v.z; s.z[3] = v.w;
    const int64_t Ls = st.L / a.S;
    const int64_t start = k * st.L;
    const int64_t end = min(start + st.L, st.g.V);
    auto record = [&](int sb) {  // sb < 0: U, sb in [0, S-1): M, sb == S-1: E
        for (uint32_t mm = mask; mm; mm &= mm - 1) {
            const int slot = __ffs((int)mm) - 1;
            uint32_t* dst = sb < 0 ? a.U + a.slot(clip, k, c, slot)
                                   : (sb == a.S - 1 ? a.E + a.slot(clip, k, c, slot) : a.M + a.mslot(clip, k, c, slot, sb));
#pragma unroll
            for (int w = 0; w < 4; ++w) dst[w] = ofp_f2u(s.z[w]);
        }
    };
    record(-1);
#pragma unroll 1
    for (int sb = 0; sb < a.S; ++sb) {
        const int64_t t0 = min(start + sb * Ls, end);
        const int64_t t1 = sb == a.S - 1 ? end : min(start + (sb + 1) * Ls, end);
        hp_span<false, CH>(st, s, chain, t0, t1);
        record(sb);
    }
}

// ---- verification of the candidates.  Three steps per round (k_hp_match / k_hp_resolve / k_hp_run), each step
// written as a device function of ONE chain.  A round whose predecessor left nothing open returns at once
// (HpCand::prev), so the host enqueues a fixed number of rounds ahead and reads the last round's counters with the
// call's final synchronisation (tuning host_verify: a group of rounds per host round trip instead).
// (A workgroup per chain running all rounds between barriers was measured as well: equal for big batches, but the
// lone clip's stage takes 3.3 instead of 2.0 ms -- the run step's lanes then sit on 8 CUs instead of all.)

// wave-scope synchronisation: the resolve step is the work of ONE wave, whose lanes exchange data through LDS and
// global memory (inside a multi-wave workgroup a workgroup barrier would be wrong there)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// step B1: nxt[k][r_prev] = the slot of chunk k whose start state equals E[k-1][r_prev] (255: none)
__device__ __forceinline__ void hp_match_item(const HpCand& a, int64_t clip, int c, int64_t k, int rp) {
    const int R1 = a.R + 1;
    const int C = a.st.g.C;
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
    // chain-major layout [clip][c][k][r]: the rows k0..k0+63 of one chain that the resolve step stages
    // are contiguous bytes
    a.nxt[(((clip * C + c) * a.st.n_chunks + k) * R1) + rp] = res;
}

__global__ __launch_bounds__(256) void k_hp_match(HpCand a, int64_t n_items) {
    OFP_LATENCY_BOUND_KERNEL();
    if (a.prev && a.prev[0] + a.prev[1] == 0) return;
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_items) return;
    const int R1 = a.R + 1;
    const int rp = (int)(id % R1);
    int64_t q = id / R1;
    const int C = a.st.g.C;
    const int c = (int)(q % C);
    q /= C;
    hp_match_item(a, q / a.st.n_chunks, c, q % a.st.n_chunks, rp);
}

// step A2 (once per call): gs[k] = the slot of chunk k whose END state is shared by the most
// candidates (at least two), -1 if all end states differ.  Candidates that agree with each other
// have merged, and then almost surely with the true trajectory as well: the guess the resolve step
// continues from at a break (and verifies afterwards).  16 lanes per chunk, lane r holds candidate r
// (every lane of a 16-lane group calls this, `live` or not: shuffles).
__device__ __forceinline__ void hp_plurality_group(const HpCand& a, int64_t chain, int64_t k, bool live, int r) {
    const int64_t nk = a.st.n_chunks;
    const int C = a.st.g.C;
    const int64_t clip = chain / C;
    const int c = (int)(chain % C);
    uint32_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;
    if (live && r < a.R) {
        const uint32_t* e = a.E + a.slot(clip, k, c, r);
        e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3];
    }
    int cnt = 0;
    for (int q = 0; q < a.R; ++q)
        cnt += (__shfl(e0, q, 16) == e0 && __shfl(e1, q, 16) == e1 && __shfl(e2, q, 16) == e2 &&
                __shfl(e3, q, 16) == e3) ? 1 : 0;
    // most votes, lowest slot on ties; key = cnt * 16 + (15 - r)
    int key = (r < a.R && cnt >= 2) ? cnt * 16 + (15 - r) : -1;
    for (int o = 8; o > 0; o >>= 1) key = max(key, __shfl_xor(key, o, 16));
    if (live && r == 0) a.gs[chain * nk + k] = (int8_t)(key < 0 ? -1 : 15 - (key & 15));
}

__global__ __launch_bounds__(256) void k_hp_plurality(HpCand a, int64_t n_items) {
    OFP_LATENCY_BOUND_KERNEL();
    const int64_t id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;  // (chain, k), k fastest
    const bool live = id < n_items;
    const int64_t nk = a.st.n_chunks;
    hp_plurality_group(a, live ? id / nk : 0, live ? id % nk : 0, live, threadIdx.x & 15);
}

// step B2: one wave per chain resolves which slot every chunk starts from.
//
// sel[k] = nxt[k][sel[k-1]]: a chain of table look-ups, i.e. a composition of maps
// f_k : slot -> slot, which is associative -- so instead of walking the chain, each lane composes
// the maps of its 8 consecutive chunks, a 6-step scan composes those across the wave, and every
// lane then knows the slot entering its segment and fills it in.
//
// A chunk k whose true start state E[k-1][sel] matches none of its candidates is a break: it has
// to be run from that state before its end state is known, which costs a round.  Instead of
// stopping there, f_k continues from the plurality end state of chunk k (gs[k]) and the chunk is
// marked `guessed`; the run step runs chunk k from its true start state in the same round (slot R),
// and the NEXT resolve compares that exact end state with the guess.  Right (the usual case):
// nothing else to do.  Wrong: slot R is selected, the later chunks of the chain are resolved again
// and those whose start state actually changed are run again.  Without a plurality the chain is
// stuck at k until the next round.  Results stay exact; only the number of rounds changes.
constexpr int HP_SEG = 8, HP_RT = 64 * HP_SEG;               // chunks per lane, per tile
constexpr int HP_NV = HP_MAXR + 2, HP_STUCK = HP_MAXR + 1;   // map domain: slots 0..R, STUCK
struct HpResolveLds {
    uint8_t tile[HP_RT * (HP_MAXR + 1)];
    int8_t stile[HP_RT], rtile[HP_RT], gtile[HP_RT];
    uint8_t maps[2][64][HP_NV];
};

// -> bit 0: the chain is stuck at a break without a plurality, bit 1: it has unverified guesses (wave-uniform)
__device__ __forceinline__ int hp_resolve_chain(const HpCand& a, int64_t chain, int lane, HpResolveLds& L) {
    constexpr int SEG = HP_SEG, RT = HP_RT, NV = HP_NV, STUCK = HP_STUCK;
    uint8_t* tile = L.tile;
    int8_t *stile = L.stile, *rtile = L.rtile, *gtile = L.gtile;
    const int C = a.st.g.C, R1 = a.R + 1;
    const int c = (int)(chain % C);
    const int64_t clip = chain / C;
    const int64_t nk = a.st.n_chunks;
    // --- verify the guesses of earlier rounds whose chunk has been run since
    int64_t wrong = nk;
    bool pending = false;
    for (int64_t k = lane; k < nk; k += 64) {
        const int64_t ci = (clip * nk + k) * C + c;
        if (!a.guessed[ci]) continue;
        if (!a.done[ci * a.S]) {  // a guessed chunk is run whole by its sub-chunk 0 lane
            pending = true;
            continue;
        }
        const uint32_t* ex = a.E + a.slot(clip, k, c, a.R);
        const uint32_t* eg = a.E + a.slot(clip, k, c, a.sel[ci]);
        if (ex[0] == eg[0] && ex[1] == eg[1] && ex[2] == eg[2] && ex[3] == eg[3]) a.guessed[ci] = 0;
        else wrong = min(wrong, k);
    }
    for (int o = 32; o > 0; o >>= 1) wrong = min(wrong, __shfl_xor(wrong, o));
    pending = __any(pending);
    if (wrong < nk) {  // first wrong guess: select the exact slot there, forget everything after it
        for (int64_t k = wrong + lane; k < nk; k += 64) {
            const int64_t ci = (clip * nk + k) * C + c;
            a.guessed[ci] = 0;
            if (k == wrong) {
                a.sel[ci] = (int8_t)a.R;
            } else {
                a.sel[ci] = -1;  // resolved again below; outputs are dropped only where the start changes
            }
        }
        if (lane == 0) a.pos[chain] = (int32_t)(wrong + 1);
        pending = false;  // guesses before `wrong` were verified or are still counted below
        for (int64_t k = lane; k < wrong; k += 64) pending = pending || a.guessed[(clip * nk + k) * C + c];
        pending = __any(pending);
    }
    wave_sync();
    // resume where the previous round stopped: chunks before pos[] are resolved
    const int64_t kstart = max<int64_t>(1, a.pos[chain]);
    int cur = a.sel[(clip * nk + kstart - 1) * C + c];  // slot chosen for the previous chunk
    int64_t reached = nk;  // first chunk left unresolved (uniform)
    // f_k(v): the slot chunk k+1 starts from when chunk k starts from v
    auto step = [&](int i, int v) -> int {
        if (stile[i] >= 0) return stile[i];  // resolved in an earlier round
        if (v == STUCK) return STUCK;
        const uint8_t m = tile[i * R1 + v];
        if (m != 255) return m;
        return gtile[i] >= 0 ? gtile[i] : STUCK;
    };
    for (int64_t k0 = kstart; k0 < nk && reached == nk; k0 += RT) {
        const int nblk = (int)min<int64_t>(RT, nk - k0);
        wave_sync();
        {
            const uint8_t* src = a.nxt + ((clip * C + c) * nk + k0) * R1;  // nblk * R1 contiguous bytes
            for (int i = lane; i < nblk * R1; i += 64) tile[i] = src[i];
        }
        for (int i = lane; i < nblk; i += 64) {
            stile[i] = a.sel[(clip * nk + k0 + i) * C + c];
            rtile[i] = a.ran[(clip * nk + k0 + i) * C + c];
            gtile[i] = a.gs[chain * nk + k0 + i];
        }
        wave_sync();
        // 1. the map of this lane's segment [s0, s1)
        const int s0 = min(lane * SEG, nblk), s1 = min(s0 + SEG, nblk);
        for (int v = 0; v < NV; ++v) {
            int w = (v <= a.R || v == STUCK) ? v : STUCK;
            for (int i = s0; i < s1; ++i) w = step(i, w);
            L.maps[0][lane][v] = (uint8_t)w;
        }
        wave_sync();
        // 2. inclusive scan over the lanes: maps[b][l] = segment 0 .. l composed
        int b = 0;
        for (int o = 1; o < 64; o <<= 1) {
            for (int v = 0; v < NV; ++v)
                L.maps[b ^ 1][lane][v] = lane >= o ? L.maps[b][lane][L.maps[b][lane - o][v]] : L.maps[b][lane][v];
            wave_sync();
            b ^= 1;
        }
        // 3. the slot entering this lane's segment, then the segment itself
        int v = lane == 0 ? cur : L.maps[b][lane - 1][cur];
        int first_stuck = RT;  // index in the tile of the chunk the chain is stuck at
        for (int i = s0; i < s1; ++i) {
            if (stile[i] >= 0) {
                v = stile[i];
                continue;
            }
            if (v == STUCK) break;  // an earlier chunk is stuck: this one stays unresolved
            const uint8_t m = tile[i * R1 + v];
            const int64_t ci = (clip * nk + k0 + i) * C + c;
            int sv;
            bool redo = false;
            if (m != 255) {
                sv = m;
                redo = rtile[i] != sv;  // its output (if any) came from another start
            } else {
                redo = true;  // a break is run whole from its (new) true start
                if (gtile[i] >= 0) {
                    sv = gtile[i];
                    a.guessed[ci] = 1;
                    pending = true;
                } else {
                    sv = -1;
                    first_stuck = i;
                }
            }
            if (redo) {
                for (int sb = 0; sb < a.S; ++sb) a.done[ci * a.S + sb] = 0;
                a.mrg[ci] = -1;
            }
            if (sv < 0) {
                v = STUCK;
                break;
            }
            a.sel[ci] = (int8_t)sv;
            v = sv;
        }
        for (int o = 32; o > 0; o >>= 1) first_stuck = min(first_stuck, __shfl_xor(first_stuck, o));
        if (first_stuck < RT) reached = k0 + first_stuck;
        cur = L.maps[b][63][cur];  // the slot entering the next tile (meaningless once stuck)
        pending = __any(pending);
    }
    if (lane == 0) a.pos[chain] = (int32_t)reached;
    return (reached < nk ? 1 : 0) | (pending ? 2 : 0);
}

__global__ __launch_bounds__(64) void k_hp_resolve(HpCand a) {
    OFP_LATENCY_BOUND_KERNEL();
    if (a.prev && a.prev[0] + a.prev[1] == 0) return;
    __shared__ HpResolveLds L;
    const int f = hp_resolve_chain(a, blockIdx.x, threadIdx.x, L);
    if (threadIdx.x == 0) {
        if (f & 1) atomicAdd(a.counters, 1);
        if (f & 2) atomicAdd(a.counters + 1, 1);
    }
}

// step C: run every chunk whose true start state is known and that has not produced its
// output yet, from that state; a chunk without a matching candidate fills slot R.
// item = (chain, chunk, sub-chunk).  A chunk whose start state matched candidate sel[] is run
// by its S lanes in parallel, lane sb > 0 starting from that candidate's recorded inner state;
// a chunk that starts from an unmatched state (a break, guessed or not) or whose own slot is the
// exact re-run has no such states and is run whole by lane 0.
// -> true if the item left sub-chunks open (an early stop): another round is needed
template <int CH>
__device__ __forceinline__ bool hp_run_item(const HpCand& a, int64_t chain, int64_t k, int sb) {
    const HpArgs& st = a.st;
    const int C = st.g.C;
    const int64_t clip = chain / C;
    const int c = (int)(chain % C);
    const int64_t ci = (clip * st.n_chunks + k) * C + c;
    // >= 0: an earlier whole run joined candidate mg; its remaining sub-chunks are open.  Read BEFORE done[]: a whole
    // run publishes done[] first and mrg last, so an item that sees the join also sees which sub-chunks it covered
    // (the items of one chunk normally sit in one wave and read all of this before any of them walks; with a
    // sub-chunk count that does not divide 64 they can straddle two)
    const int mg = a.mrg[ci];
    __threadfence();
    if (a.done[ci * a.S + sb]) return false;
    int sp = 0;
    if (k > 0) {
        sp = a.sel[ci - C];
        if (sp < 0) return false;  // predecessor not resolved yet
    }
    const int own = a.sel[ci];
    const bool whole = (own < 0 || own >= a.R || a.guessed[ci]) && mg < 0;  // no exact inner states
    if (whole && sb > 0) return false;
    HpStep s;
    s.coeffs(st.b, st.a);
    uint32_t xin[4] = {0u, 0u, 0u, 0u};
    if (sb > 0) {
        const uint32_t* m = a.M + a.mslot(clip, k, c, mg >= 0 ? mg : own, sb - 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) xin[i] = m[i];
    } else if (k > 0) {
        const uint32_t* e = a.E + a.slot(clip, k - 1, c, sp);
#pragma unroll
        for (int i = 0; i < 4; ++i) xin[i] = e[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s.z[i] = ofp_u2f(xin[i]);
    const int64_t start = k * st.L;
    const int64_t end = min(start + st.L, st.g.V);
    const int64_t Ls = st.L / a.S;
    // One walk call site for both kinds of item (a second instantiation of the unrolled loop doubles the kernel):
    // a verified sub-chunk is the single piece [sb], a whole run the pieces 0 .. S-1 (or ONE piece [start, end)
    // without `early`).  With `early` a whole run stops at the first sub-chunk boundary where its state equals
    // (bitwise) the state some candidate recorded there: from that point on it IS that candidate's trajectory, so
    // the chunk's end state is known at once (E of that candidate) and the sub-chunks behind the boundary are left
    // to the lanes of the next round, which start from the candidate's inner states.
    const bool pieces = !whole || a.early;
    const int q0 = whole ? 0 : sb, q1 = !whole ? sb + 1 : (a.early ? a.S : 1);
    int joined = -1, n_done = a.S;
#pragma unroll 1
    for (int q = q0; q < q1; ++q) {
        const int64_t t0 = pieces ? min(start + q * Ls, end) : start;
        const int64_t t1 = (!pieces || q == a.S - 1) ? end : min(start + (q + 1) * Ls, end);
        hp_span<true, CH>(st, s, chain, t0, t1);
        if (whole && q + 1 < q1) {
            const uint32_t z0 = ofp_f2u(s.z[0]), z1 = ofp_f2u(s.z[1]), z2 = ofp_f2u(s.z[2]), z3 = ofp_f2u(s.z[3]);
            for (int r = 0; r < a.R; ++r) {
                if (a.U[a.slot(clip, k, c, r)] == 0x7fc00001u) continue;  // a slot no run filled: no records
                const uint32_t* m = a.M + a.mslot(clip, k, c, r, q);
                if (m[0] == z0 && m[1] == z1 && m[2] == z2 && m[3] == z3) {
                    joined = r;
                    break;
                }
            }
            if (joined >= 0) {
                n_done = q + 1;
                break;
            }
        }
    }
    if (!whole) {
        if (sb == 0) a.ran[ci] = (int8_t)own;
        a.done[ci * a.S + sb] = 1;
        return false;
    }
    if (own < 0 || a.guessed[ci]) {
        // no candidate matched: this exact run becomes slot R.  sel[ci] itself is NOT written
        // here (a successor running in this same launch must not see a half-filled slot); the
        // next match step finds slot R and the resolve step selects it (or, for a guessed chunk,
        // compares it with the guess).
        uint32_t* u = a.U + a.slot(clip, k, c, a.R);
        uint32_t* e = a.E + a.slot(clip, k, c, a.R);
        const uint32_t* ej = joined >= 0 ? a.E + a.slot(clip, k, c, joined) : nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u[i] = xin[i];
            e[i] = ej ? ej[i] : ofp_f2u(s.z[i]);
        }
    }
    a.ran[ci] = (int8_t)a.R;
    for (int q = 0; q < n_done; ++q) a.done[ci * a.S + q] = 1;
    __threadfence();
    a.mrg[ci] = (int8_t)joined;
    return joined >= 0;  // open sub-chunks: another round
}

template <int CH>
__global__ __launch_bounds__(64) void k_hp_run(HpCand a, int64_t n_threads) {
    OFP_LATENCY_BOUND_KERNEL();
    if (a.prev && a.prev[0] + a.prev[1] == 0) return;
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_threads) return;
    if (CH > 0) {  // lane = (clip, chunk, sub-chunk, channel): see k_hp_seg0
        constexpr int CHD = CH > 0 ? CH : 1;
        const int c = (int)(id % CHD);
        const int64_t q = id / CHD;
        const int sb = (int)(q % a.S);
        const int64_t kc = q / a.S;
        if (hp_run_item<CH>(a, kc / a.st.n_chunks * CH + c, kc % a.st.n_chunks, sb)) atomicAdd(a.counters + 1, 1);
        return;
    }
    const int sb = (int)(id % a.S);
    const int64_t kc = id / a.S;
    if (hp_run_item<CH>(a, kc / a.st.n_chunks, kc % a.st.n_chunks, sb)) atomicAdd(a.counters + 1, 1);
}

// k_hp_run whose output leaves as complete lines (walk_lines; planar input; every position that matters a multiple of 32
// steps: the host checks).  The items' logic is hp_run_item's, arranged so that the wave stays together: every lane goes
// through the same number of piece trips, the lanes without work (or done with theirs) lend their stores.
__global__ __launch_bounds__(64) void k_hp_run_lines(HpCand a, int64_t n_threads) {
    OFP_LATENCY_BOUND_KERNEL();
    if (a.prev && a.prev[0] + a.prev[1] == 0) return;
    __shared__ float tile[64 * WL_PITCH];
    const HpArgs& st = a.st;
    const int C = st.g.C;
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool act = id < n_threads;
    const int sb = act ? (int)(id % a.S) : 0;
    const int64_t kc = act ? id / a.S : 0;
    const int64_t chain = kc / st.n_chunks, k = kc % st.n_chunks;
    const int64_t clip = chain / C;
    const int c = (int)(chain % C);
    const int64_t ci = (clip * st.n_chunks + k) * C + c;
    int mg = -1, sp = 0, own = -1;
    bool whole = false;
    if (act) {  // (hp_run_item's conditions, in its order)
        mg = a.mrg[ci];
        __threadfence();
        if (a.done[ci * a.S + sb]) act = false;
    }
    if (act && k > 0) {
        sp = a.sel[ci - C];
        if (sp < 0) act = false;  // predecessor not resolved yet
    }
    if (act) {
        own = a.sel[ci];
        whole = (own < 0 || own >= a.R || a.guessed[ci]) && mg < 0;  // no exact inner states
        if (whole && sb > 0) act = false;
    }
    if (!__any(act)) return;
    HpStep s;
    s.coeffs(st.b, st.a);
    uint32_t xin[4] = {0u, 0u, 0u, 0u};
    if (act) {
        if (sb > 0) {
            const uint32_t* m = a.M + a.mslot(clip, k, c, mg >= 0 ? mg : own, sb - 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) xin[i] = m[i];
        } else if (k > 0) {
            const uint32_t* e = a.E + a.slot(clip, k - 1, c, sp);
#pragma unroll
            for (int i = 0; i < 4; ++i) xin[i] = e[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s.z[i] = ofp_u2f(xin[i]);
    const int64_t start = k * st.L;
    const int64_t end = min(start + st.L, st.g.V);
    const int64_t Ls = st.L / a.S;
    const bool pieces = !whole || a.early;
    const int q0 = whole ? 0 : sb, q1 = !whole ? sb + 1 : (a.early ? a.S : 1);
    int joined = -1, n_done = a.S;
    int trips = act ? q1 - q0 : 0;
    for (int o = 32; o > 0; o >>= 1) trips = max(trips, __shfl_xor(trips, o));
    const float* xs = st.xt + chain * st.g.Nv;
    float* os = st.out + chain * st.g.U;
    const Geom g = st.g;
    for (int it = 0; it < trips; ++it) {
        const int q = q0 + it;
        const bool on = act && q < q1 && joined < 0;
        const int64_t t0 = on ? (pieces ? min(start + q * Ls, end) : start) : 0;
        const int64_t t1 = on ? ((!pieces || q == a.S - 1) ? end : min(start + (q + 1) * Ls, end)) : 0;
        walk_lines_to(xs + t0, t1 - t0, s, tile, [&](int64_t b) -> float* {
            const int64_t u = hp_dst(g, t0 + 32 * b);  // (a batch lies inside one of the three regions of the stream)
            return u >= 0 ? os + u : nullptr;
        });
        if (on && whole && q + 1 < q1) {
            const uint32_t z0 = ofp_f2u(s.z[0]), z1 = ofp_f2u(s.z[1]), z2 = ofp_f2u(s.z[2]), z3 = ofp_f2u(s.z[3]);
            for (int r = 0; r < a.R; ++r) {
                if (a.U[a.slot(clip, k, c, r)] == 0x7fc00001u) continue;  // a slot no run filled: no records
                const uint32_t* m = a.M + a.mslot(clip, k, c, r, q);
                if (m[0] == z0 && m[1] == z1 && m[2] == z2 && m[3] == z3) {
                    joined = r;
                    break;
                }
            }
            if (joined >= 0) n_done = q + 1;
        }
    }
    if (!act) return;
    if (!whole) {
        if (sb == 0) a.ran[ci] = (int8_t)own;
        a.done[ci * a.S + sb] = 1;
        return;
    }
    if (own < 0 || a.guessed[ci]) {  // (see hp_run_item)
        uint32_t* u = a.U + a.slot(clip, k, c, a.R);
        uint32_t* e = a.E + a.slot(clip, k, c, a.R);
        const uint32_t* ej = joined >= 0 ? a.E + a.slot(clip, k, c, joined) : nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u[i] = xin[i];
            e[i] = ej ? ej[i] : ofp_f2u(s.z[i]);
        }
    }
    a.ran[ci] = (int8_t)a.R;
    for (int q = 0; q < n_done; ++q) a.done[ci * a.S + q] = 1;
    __threadfence();
    a.mrg[ci] = (int8_t)joined;
    if (joined >= 0) atomicAdd(a.counters + 1, 1);  // open sub-chunks: another round
}

// ---- elementwise stages (planar, in place) -------------------------------------------
// rectified dB (detection.py:747-748).  from_x: no high-pass, read the audio through the
// stream mapping; else in place on the filtered buffer.
__global__ __launch_bounds__(256) void k_rect_db(Geom g, const float* __restrict__ xt, float* __restrict__ buf,
                                                 int64_t n_chains, int from_x, float floor_db) {
    const int64_t total = n_chains * g.U;
    const int64_t tid0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
    if (!from_x) {
        // in place on the filtered stream: 16 bytes per lane and access (the work space is 256-byte aligned)
        float4* b4 = reinterpret_cast<float4*>(buf);
        const int64_t n4 = total >> 2;
        for (int64_t i = tid0; i < n4; i += nthr) {
            float4 v = b4[i];
            v.x = ofp_rect_db(v.x, floor_db);
            v.y = ofp_rect_db(v.y, floor_db);
            v.z = ofp_rect_db(v.z, floor_db);
            v.w = ofp_rect_db(v.w, floor_db);
            b4[i] = v;
        }
        for (int64_t i = (n4 << 2) + tid0; i < total; i += nthr) buf[i] = ofp_rect_db(buf[i], floor_db);
        return;
    }
    for (int64_t i = tid0; i < total; i += nthr) {
        const int64_t chain = i / g.U, u = i - chain * g.U;
        buf[i] = ofp_rect_db(xt[chain * g.Nv + u_src_planar(g, u)], floor_db);
    }
}

// k_rect_db and k_ar_sym_local in one pass over the filtered stream (symmetric slow follower, U and the follower
// chunk multiples of 4): one wave per (chain, follower chunk) turns its chunk into rectified dB in place and,
// from the same registers, forms the chunk's weighted sum P = sum_t c q^t dB[len-1-t] for the closed-form guess
// (k_ar_sym_combine) -- the dB stream is not read a second time for it.  The chunk is walked from its end, so
// the weights only decrease (no overflow for any q in (0, 1)).  P is a guess generator: its summation order does
// not matter, the chunk passes verify every state bit for bit.
__global__ __launch_bounds__(64) void k_rect_db_sym(ArArgs a, float* __restrict__ buf, int64_t n_waves,
                                                    double* __restrict__ P) {
    const int64_t id = blockIdx.x;  // one wave per (chain, chunk)
    if (id >= n_waves) return;
    const int lane = threadIdx.x;
    const int64_t j = id % a.n_chunks;
    const int64_t chain = id / a.n_chunks;
    const int64_t base = j * a.L;
    const int64_t len = min(a.L, a.g.U - base);  // a multiple of 4 (the host checks)
    float4* xs = reinterpret_cast<float4*>(buf + chain * a.g.U + base);
    const int64_t ng = len >> 2;  // float4 groups; group ng-1-g holds t = 4 g .. 4 g + 3 (from the end)
    const double c = (double)a.sa, q = 1.0 - c;
    const double q2 = q * q, q4 = q2 * q2;
    double q256 = q4;  // q^256 = (q^4)^64
    for (int i = 0; i < 6; ++i) q256 *= q256;
    double w = c;      // c q^(4 lane): the weight of this lane's LAST element in its first group
    for (int i = 0; i < lane; ++i) w *= q4;
    double acc = 0.0;
    // Four groups per lane and trip, their loads issued together: one 16-byte load in flight per lane left the pass
    // latency-bound (a wave moved 1 KB per memory round trip + ~180 fp64 operations: 2.9 TB/s of traffic at full
    // occupancy on C3's 1.8 G samples, where a copy reaches 4.5+).
    auto one = [&](float4 v, int64_t g) {
        v.x = ofp_rect_db(v.x, a.floor_db);
        v.y = ofp_rect_db(v.y, a.floor_db);
        v.z = ofp_rect_db(v.z, a.floor_db);
        v.w = ofp_rect_db(v.w, a.floor_db);
        xs[ng - 1 - g] = v;
        // t = 4 g + 0 for .w, + 1 for .z, + 2 for .y, + 3 for .x
        acc += w * ((double)v.w + q * ((double)v.z + q * ((double)v.y + q * (double)v.x)));
        w *= q256;
    };
    int64_t g = lane;
    for (; g + 192 < ng; g += 256) {
        const float4 v0 = xs[ng - 1 - g], v1 = xs[ng - 1 - g - 64], v2 = xs[ng - 1 - g - 128], v3 = xs[ng - 1 - g - 192];
        __builtin_amdgcn_sched_barrier(0);
        one(v0, g);
        one(v1, g + 64);
        one(v2, g + 128);
        one(v3, g + 192);
    }
    for (; g < ng; g += 64) one(xs[ng - 1 - g], g);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) P[id] = acc;
}

// back to linear (detection.py:753-754), planar in place; the MAIN part is also written to
// the caller's interleaved [N'][C] array through an LDS tile transpose.
// With `sum` the pass also leaves, per (main block, channel), the largest and the smallest value of the block
// (sum_max / sum_minv [clips][nb][C], zeroed by the call's fill; the minimum as 0x7f800000 - bits so that zero is
// "none yet"): k_block_scan then reads a block's samples only if they can matter (see there).  Values are
// non-negative (clipped, :754), so their bit patterns order like the values; NaN is ignored on both sides, as a
// comparison with a threshold ignores it.
__device__ __forceinline__ void rel_out_tile(const Geom& g, float* __restrict__ buf, float* __restrict__ rel_out,
                                             float floor_db, int TU, uint32_t* __restrict__ sum_max,
                                             uint32_t* __restrict__ sum_minv, int64_t nb, float* __restrict__ rel_warm,
                                             int planar, int64_t tile_x, float* tile) {
    // rel_warm: the rows of the warm-up part in the output's interleaved order ([clip][n_wb][C]), for the tracker's
    // *_il kernels; planar = 0: the planar series are not written back (nothing reads them then)
    // tile: [C][TU+4], then the tile's summaries [2][C][nbt]
    const int C = g.C;
    const int S = TU + 4;
    const int64_t clip = blockIdx.y;
    const int64_t u0 = tile_x * TU;
    const int nt = (int)min<int64_t>(TU, g.U - u0);
    const int total = nt * C;
    // main blocks this tile touches: j0 .. j0 + nbt - 1 (host: summaries only with B a multiple of 4 and >= 32)
    const bool sum = sum_max != nullptr && u0 + nt > g.n_wb;
    const int64_t j0 = max<int64_t>(u0 - g.n_wb, 0) / g.B;
    const int nbt = TU / g.B + 2;
    uint32_t* s_max = reinterpret_cast<uint32_t*>(tile + (size_t)C * S);
    uint32_t* s_minv = s_max + C * nbt;
    const int lb = (g.B & (g.B - 1)) == 0 ? 31 - __clz(g.B) : -1;  // (a shift for power-of-two blocks)
    const int mbase = (int)(u0 - g.n_wb - j0 * g.B);  // (negative while the tile is still in the warm-up part: j0 = 0 then)
    if (sum) {
        for (int i = threadIdx.x; i < 2 * C * nbt; i += 256) s_max[i] = 0u;
        __syncthreads();
    }
    // 16-byte accesses on both sides whenever the geometry keeps them aligned (it does for every block size
    // that is a multiple of 4): the planar series in steps of 4 samples, the interleaved output in steps of
    // 4 floats of its [time][channel] order
    const bool vec = (g.U & 3) == 0 && (TU & 3) == 0 && (nt & 3) == 0 && ((g.n_wb * C) & 3) == 0 &&
                     (((int64_t)TU * C) & 3) == 0 && ((g.Nm * C) & 3) == 0;
    if (vec) {
        const int q = nt >> 2;  // float4 groups per channel row
        const int lq = (q & (q - 1)) == 0 ? 31 - __clz(q) : -1;  // (a shift when q is a power of two: every full tile)
        auto at = [&](int i) -> float4* {
            const int c = lq >= 0 ? i >> lq : i / q, t = (i - c * q) << 2;
            return reinterpret_cast<float4*>(buf + (clip * C + c) * g.U + u0 + t);
        };
        auto one = [&](int i, float4 v) {
            const int c = lq >= 0 ? i >> lq : i / q, t = (i - c * q) << 2;
            v.x = ofp_rel_linear(v.x, floor_db);
            v.y = ofp_rel_linear(v.y, floor_db);
            v.z = ofp_rel_linear(v.z, floor_db);
            v.w = ofp_rel_linear(v.w, floor_db);
            if (planar) *at(i) = v;
            *reinterpret_cast<float4*>(tile + c * S + t) = v;
            const int m = mbase + t;  // main row of the group's first value relative to block j0 (the four share a block)
            if (sum && m >= 0) {
                float hi = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)), lo = fminf(fminf(v.x, v.y), fminf(v.z, v.w));
                if (!(hi >= 0.0f)) hi = 0.0f;                 // (all four NaN)
                if (!(lo >= 0.0f)) lo = __builtin_inff();
                const int jb = lb >= 0 ? m >> lb : m / g.B;
                atomicMax(&s_max[c * nbt + jb], ofp_f2u(hi));
                atomicMax(&s_minv[c * nbt + jb], 0x7f800000u - ofp_f2u(lo));
            }
        };
        // the loads of four groups per thread are issued together (see k_rect_db_sym: one in flight left the pass
        // latency-bound)
        int i = threadIdx.x;
        for (; i + 768 < q * C; i += 1024) {
            const float4 v0 = *at(i), v1 = *at(i + 256), v2 = *at(i + 512), v3 = *at(i + 768);
            __builtin_amdgcn_sched_barrier(0);
            one(i, v0);
            one(i + 256, v1);
            one(i + 512, v2);
            one(i + 768, v3);
        }
        for (; i < q * C; i += 256) one(i, *at(i));
    } else {
        for (int i = threadIdx.x; i < total; i += 256) {
            const int c = i / nt, t = i - c * nt;
            float* p = buf + (clip * C + c) * g.U + u0 + t;
            const float v = ofp_rel_linear(*p, floor_db);
            if (planar) *p = v;
            tile[c * S + t] = v;
        }
    }
    __syncthreads();
    if (sum) {
        for (int i = threadIdx.x; i < C * nbt; i += 256) {
            const int c = i / nbt;
            const int64_t j = j0 + (i - c * nbt);
            if (j < nb && (s_max[i] | s_minv[i]) != 0u) {
                atomicMax(&sum_max[(clip * nb + j) * C + c], s_max[i]);
                atomicMax(&sum_minv[(clip * nb + j) * C + c], s_minv[i]);
            }
        }
    }
    if (!rel_out) return;
    float* dst = rel_out + clip * g.Nm * C;
    float* dstw = rel_warm ? rel_warm + clip * g.n_wb * C : nullptr;
    const int64_t m0 = u0 - g.n_wb;  // main-part row of this tile's first time step (tiles do not straddle n_wb
                                     // unless TU does not divide it: those take the scalar path)
    if (vec && (m0 >= 0 || m0 + nt <= 0) && (reinterpret_cast<uintptr_t>(rel_out) & 15u) == 0) {
        if (m0 < 0 && !dstw) return;
        float4* d4 = reinterpret_cast<float4*>(m0 < 0 ? dstw + u0 * C : dst + m0 * C);
        const int lc = (C & (C - 1)) == 0 ? 31 - __clz(C) : -1;
        for (int i = threadIdx.x; i < (total >> 2); i += 256) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 4 * i + e;
                const int t = lc >= 0 ? j >> lc : j / C, c = j - t * C;
                o[e] = tile[c * S + t];
            }
            d4[i] = make_float4(o[0], o[1], o[2], o[3]);
        }
        return;
    }
    for (int i = threadIdx.x; i < total; i += 256) {
        const int t = i / C, c = i - t * C;
        const int64_t m = u0 + t - g.n_wb;
        if (m >= 0) dst[m * C + c] = tile[c * S + t];
        else if (dstw) dstw[(u0 + t) * C + c] = tile[c * S + t];
    }
}

// (a workgroup takes several tiles in turn: one workgroup per 2 048 values made the launch 540 000 workgroups for 48 C2
//  clips, and in flight -- where it competes with seven other kernels for every dispatch -- it took 7.5 x its lone time,
//  against 2 x for the dB pass with its 34 000 workgroups)
__global__ __launch_bounds__(256) void k_rel_out(Geom g, float* __restrict__ buf, float* __restrict__ rel_out,
                                                 float floor_db, int TU, uint32_t* __restrict__ sum_max,
                                                 uint32_t* __restrict__ sum_minv, int64_t nb, float* __restrict__ rel_warm,
                                                 int planar, int64_t n_tiles) {
    extern __shared__ float tile[];
    for (int64_t tx = blockIdx.x; tx < n_tiles; tx += gridDim.x) {
        rel_out_tile(g, buf, rel_out, floor_db, TU, sum_max, sum_minv, nb, rel_warm, planar, tx, tile);
        __syncthreads();  // (the tile and its summaries are free again)
    }
}

// ---- block scan: first upward crossing and last sample below `off`, per
// (clip, main block, channel) -- everything of detection.py:759-770,784-790 that
// does not depend on the hysteresis state.
struct ScanArgs {
    Geom g;
    const float* rel;  // planar [clip*C + c][U]
    const float* thr_mn;
    const float* thr_mx;  // [clips][nb][C] (relative mode)
    const float* on_f;
    const float* off_f;
    const double* on_d;
    int manual;
    int64_t nb, n_clips;
    int32_t* first_cross;  // [clips][nb][C]: index or -1
    int32_t* last_below;   // [clips][nb][C]: index or -1
    uint32_t* vflag;       // [clips][nb]: some channel has an upward crossing in this block (zeroed by the host)
    const uint32_t* sum_max;   // [clips][nb][C] largest / smallest value of the block (k_rel_out), or NULL
    const uint32_t* sum_minv;
};

// One wave per (chain, block), lanes over the rows of the block (coalesced); the first crossing
// and the last row below `off` come out of two ballots per 64 rows.
__global__ __launch_bounds__(256) void k_block_scan(ScanArgs a) {
    OFP_LATENCY_BOUND_KERNEL();
    const int C = a.g.C, B = a.g.B;
    const int lane = threadIdx.x & 63;
    const int64_t total = a.n_clips * a.nb * C;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    // item order (clip, c, block): consecutive waves read consecutive blocks of one series.  The item's coordinates
    // are carried along instead of being divided out of `id` every trip (four 64-bit divisions per item were most
    // of this kernel's vector instructions)
    int64_t j = wave0 % a.nb, chain = wave0 / a.nb;
    const int64_t step_j = n_waves % a.nb, step_chain = n_waves / a.nb;
    for (int64_t id = wave0; id < total; id += n_waves, j += step_j, chain += step_chain) {
        if (j >= a.nb) {
            j -= a.nb;
            chain += 1;
        }
        const int64_t clip = (uint32_t)chain / (uint32_t)C;   // (chains < 2^31: one 32-bit division)
        const int c = (int)(chain - clip * C);
        const int64_t oi = (clip * a.nb + j) * C + c;
        float on, off;
        double on0;
        if (a.manual) {
            on = a.on_f[c];
            on0 = a.on_d[c];
            off = a.off_f[c];
        } else {
            const float mn = a.thr_mn[oi], mx = a.thr_mx[oi];
            const float t1 = mx * a.on_f[c];
            on = t1 + mn;  // detection.py:763
            on0 = (double)on;
            const float t2 = mx * a.off_f[c];
            off = t2 + mn;  // detection.py:787
        }
        const float* r = a.rel + chain * a.g.U + a.g.n_wb + j * B;
        if (a.sum_max) {
            // Most blocks hold neither: no value above `on` (no crossing: first = -1) and a last row below `off`
            // (last = B - 1) -- decided from the block's extremes and its last sample, without reading the block.
            // Exact: a crossing needs some v > on; the last row below `off` is row B - 1 whenever that row is below,
            // and none exists if the smallest value is not below.
            const float bmax = ofp_u2f(a.sum_max[oi]), bmin = ofp_u2f(0x7f800000u - a.sum_minv[oi]);
            const float vlast = r[B - 1];
            const bool scan_first = bmax > on;
            const bool last_known = vlast < off || !(bmin < off);
            if (!scan_first && last_known) {
                if (lane == 0) {
                    a.first_cross[oi] = -1;
                    a.last_below[oi] = vlast < off ? B - 1 : -1;
                }
                continue;
            }
        }
        // detection.py:769: row 0 compares prev_values (float64 copy of the previous block's last
        // row; zeros before the first main block) with the threshold
        float carry = (j == 0) ? 0.0f : r[-1];
        int first = -1, last = -1;
        // the rows of the block, 64 per step; the loads of up to 8 steps are issued together (one memory round
        // trip per 512 rows instead of one per 64)
        for (int tb = 0; tb < B; tb += 512) {
            float vv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int t = tb + 64 * q + lane;
                vv[q] = t < B ? r[t] : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int t0 = tb + 64 * q;
                if (t0 >= B) break;
                const int t = t0 + lane;
                const float v = vv[q];
                float prev = __shfl_up(v, 1);
                if (lane == 0) prev = carry;
                const bool below_before = (t == 0) ? ((double)prev < on0) : (prev < on);
                const unsigned long long mc = __ballot(t < B && v > on && below_before);
                const unsigned long long mb = __ballot(t < B && v < off);
                if (first < 0 && mc) first = t0 + __builtin_ctzll(mc);
                if (mb) last = t0 + 63 - __builtin_clzll(mb);
                carry = __shfl(v, 63);
            }
        }
        if (lane == 0) {
            a.first_cross[oi] = first;
            a.last_below[oi] = last;
            if (first >= 0) a.vflag[clip * a.nb + j] = 1u;  // every writer stores the same value
        }
    }
}

// The same pass over the INTERLEAVED envelope ([clip][Nm][CH], relative thresholds): one wave per (clip, block), lane =
// (row of a step, channel) -- a step is 64 / CH rows, one coalesced 256-byte load; the ballots then hold the CH
// channels bit-interleaved and every lane picks its channel's bits.  A block is read only if one of its channels
// needs it (same rule as above).
template <int CH>
__global__ __launch_bounds__(256) void k_block_scan_il(ScanArgs a, const float* __restrict__ rel_il) {
    OFP_LATENCY_BOUND_KERNEL();
    constexpr int RPS = 64 / CH;  // rows per step
    constexpr unsigned long long CHM = CH == 8 ? 0x0101010101010101ull : (CH == 4 ? 0x1111111111111111ull : 1ull);  // (CH = 64: one row per step)
    const int B = a.g.B;
    const int lane = threadIdx.x & 63;
    const int c = lane % CH, rl = lane / CH;
    const unsigned long long mine = CHM << c;
    const int64_t total = a.n_clips * a.nb;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const float on_c = a.on_f[c], off_c = a.off_f[c];
    for (int64_t id = wave0; id < total; id += n_waves) {
        const int64_t clip = id / a.nb, j = id - clip * a.nb;
        const int64_t oi = (clip * a.nb + j) * CH + c;
        const float mn = a.thr_mn[oi], mx = a.thr_mx[oi];
        const float t1 = mx * on_c;
        const float on = t1 + mn;  // detection.py:763
        const double on0 = (double)on;
        const float t2 = mx * off_c;
        const float off = t2 + mn;  // detection.py:787
        const float* r = rel_il + (clip * a.g.Nm + j * B) * CH;  // the block's first row
        if (a.sum_max) {
            const float bmax = ofp_u2f(a.sum_max[oi]), bmin = ofp_u2f(0x7f800000u - a.sum_minv[oi]);
            const float vlast = r[(int64_t)(B - 1) * CH + c];
            const bool need = bmax > on || !(vlast < off || !(bmin < off));
            if (!__any(need)) {
                if (lane < CH) {
                    a.first_cross[oi] = -1;
                    a.last_below[oi] = vlast < off ? B - 1 : -1;
                }
                continue;
            }
        }
        // detection.py:769: row 0 compares prev_values (the previous block's last row; zeros before the first main block)
        float carry = (j == 0) ? 0.0f : r[c - CH];
        int first = -1, last = -1;
        for (int tb = 0; tb < B; tb += 8 * RPS) {
            float vv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int t = tb + q * RPS + rl;
                vv[q] = t < B ? r[(int64_t)(tb + q * RPS) * CH + lane] : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int t0 = tb + q * RPS;
                if (t0 >= B) break;
                const int t = t0 + rl;
                const float v = vv[q];
                float prev = __shfl_up(v, CH);
                if (rl == 0) prev = carry;
                const bool below_before = (t == 0) ? ((double)prev < on0) : (prev < on);
                const unsigned long long mc = __ballot(t < B && v > on && below_before) & mine;
                const unsigned long long mb = __ballot(t < B && v < off) & mine;
                if (first < 0 && mc) first = t0 + __builtin_ctzll(mc) / CH;
                if (mb) last = t0 + (63 - __builtin_clzll(mb)) / CH;
                carry = __shfl(v, 64 - CH + c);
            }
        }
        if (lane < CH) {
            a.first_cross[oi] = first;
            a.last_below[oi] = last;
            if (first >= 0) a.vflag[clip * a.nb + j] = 1u;  // every writer stores the same value
        }
    }
}

// pc[clip][j][c]: the last block <= j of channel c that holds a row below the off threshold
// (-1: none yet).  Parallel over tiles of 256 blocks, one wave per (tile, chain): WRITE = false finds the last
// flagged block of every tile; k_last_clear_scan turns those into "the last flagged block BEFORE the tile" (one
// wave per chain, 64 tiles per step); WRITE = true fills pc from that carry.  (One wave per chain walking its
// tiles in order: 66 us for C2's 11 250 blocks, 0.56 ms for C3's 56 250.)
template <bool WRITE>
__global__ __launch_bounds__(64) void k_last_clear(const int32_t* __restrict__ lb, int32_t* __restrict__ pc,
                                                   int32_t* __restrict__ tile_last, int64_t nb, int C, int64_t n_tiles) {
    const int64_t tile = blockIdx.x % n_tiles, chain = blockIdx.x / n_tiles;  // (chains can exceed a grid's y range)
    const int c = (int)(chain % C);
    const int64_t clip = chain / C;
    const int lane = threadIdx.x;
    constexpr int U = 4;  // 4 x 64 blocks per tile: the loads are issued together
    const int64_t j0 = tile * 64 * U;
    int carry = WRITE ? tile_last[chain * n_tiles + tile] : -1;
    int f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t j = j0 + 64 * u + lane;
        f[u] = (j < nb && lb[(clip * nb + j) * C + c] >= 0) ? 1 : 0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        // the last flagged block at or before this lane's: the highest set bit of the ballot
        // among bits 0..lane (no shuffles: one ballot per 64 blocks)
        const int64_t j = j0 + 64 * u + lane;
        const unsigned long long mask = __ballot(f[u] != 0);
        if (WRITE) {
            const unsigned long long below = mask & ((2ull << lane) - 1ull);
            const int v = below ? (int)(j0 + 64 * u) + 63 - __builtin_clzll(below) : carry;
            if (j < nb) pc[(clip * nb + j) * C + c] = v;
        }
        if (mask) carry = (int)(j0 + 64 * u) + 63 - __builtin_clzll(mask);
    }
    if (!WRITE && lane == 0) tile_last[chain * n_tiles + tile] = carry;
}

// in place: tile_last[t] (last flagged block IN tile t, -1 none) -> last flagged block BEFORE tile t
__global__ __launch_bounds__(64) void k_last_clear_scan(int32_t* __restrict__ tile_last, int64_t n_tiles) {
    OFP_LATENCY_BOUND_KERNEL();
    int32_t* t = tile_last + (int64_t)blockIdx.x * n_tiles;
    const int lane = threadIdx.x;
    int carry = -1;
    for (int64_t i0 = 0; i0 < n_tiles; i0 += 64) {
        const int64_t i = i0 + lane;
        const int v = i < n_tiles ? t[i] : -1;
        const unsigned long long mask = __ballot(v >= 0);
        const unsigned long long before = mask & ((1ull << lane) - 1ull);  // flagged tiles before mine in this step
        const int src = before ? 63 - __builtin_clzll(before) : -1;
        const int got = __shfl(v, src < 0 ? 0 : src);
        if (i < n_tiles) t[i] = src >= 0 ? got : carry;
        if (mask) carry = __shfl(v, 63 - __builtin_clzll(mask));
    }
}

// The state machine only has to stop at blocks where some channel crosses its on threshold
// upwards ("visits"): everywhere else no onset can fire, the cooldown counters move in closed
// form and a latched channel is released iff a block in between holds a row below its off
// threshold -- which pc[] answers.  k_visits compacts, per clip and in order, the visited blocks
// and gathers what the machine needs there: fc, lb of the block and pc of the block before.
struct VisArgs {
    const uint32_t* vflag;
    const int32_t *fc, *lb, *pc;  // [clips][nb][C]
    int64_t nb;
    int C;
    int32_t* vis_j;               // [clips][nb] visited block indices
    int32_t *vfc, *vlb, *vpc;     // [clips][nb][C] compacted records
    int32_t* nv;                  // [clips]
};

// Three launches, all parallel over tiles of 256 blocks: count the visited blocks of every tile, scan the
// counts of a clip (one wave), then every tile copies its records behind those of the tiles before it.  (One
// workgroup per clip walking its tiles in order took 2.5 ms on C3's 56 250 blocks.)
__global__ __launch_bounds__(256) void k_visit_count(VisArgs a, int32_t* __restrict__ tile_cnt, int64_t n_tiles) {
    __shared__ int s_w[4];
    const int64_t clip = blockIdx.y, tile = blockIdx.x;
    const int64_t j = tile * 256 + threadIdx.x;
    const bool f = j < a.nb && a.vflag[clip * a.nb + j] != 0u;
    const unsigned long long m0 = __ballot(f);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = __popcll(m0);
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[clip * n_tiles + tile] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(64) void k_visit_scan(int32_t* __restrict__ tile_cnt, int64_t n_tiles, int32_t* __restrict__ nv) {
    OFP_LATENCY_BOUND_KERNEL();
    const int64_t clip = blockIdx.x;
    const int lane = threadIdx.x;
    int32_t* t = tile_cnt + clip * n_tiles;
    int base = 0;
    for (int64_t i0 = 0; i0 < n_tiles; i0 += 64) {
        const int64_t i = i0 + lane;
        const int v = i < n_tiles ? t[i] : 0;
        int inc = v;
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        if (i < n_tiles) t[i] = base + inc - v;  // exclusive offset of tile i
        base += __shfl(inc, 63);
    }
    if (lane == 0) nv[clip] = base;
}

__global__ __launch_bounds__(256) void k_visit_scatter(VisArgs a, const int32_t* __restrict__ tile_off, int64_t n_tiles) {
    __shared__ int s_w[4];
    __shared__ uint8_t s_idx[4][64];  // per wave: lane of its q-th visited block
    const int64_t clip = blockIdx.y, tile = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = a.C;
    const int64_t j0 = tile * 256;
    const int64_t j = j0 + threadIdx.x;
    const bool f = j < a.nb && a.vflag[clip * a.nb + j] != 0u;
    const unsigned long long m0 = __ballot(f);
    if (lane == 0) s_w[wave] = __popcll(m0);
    if (f) s_idx[wave][__popcll(m0 & ((1ull << lane) - 1ull))] = (uint8_t)lane;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_w[w];
    const int base = tile_off[clip * n_tiles + tile];
    // the wave copies the records of its visited blocks, lanes over (visited block, channel):
    // all loads of a step are independent
    const int n_el = __popcll(m0) * C;
    for (int e = lane; e < n_el; e += 64) {
        const int q = e / C, c = e - q * C;
        const int64_t jj = j0 + wave * 64 + s_idx[wave][q];
        const int64_t pos = clip * a.nb + base + woff + q;
        if (c == 0) a.vis_j[pos] = (int32_t)jj;
        const int64_t src = (clip * a.nb + jj) * C + c;
        a.vfc[pos * C + c] = a.fc[src];
        a.vlb[pos * C + c] = a.lb[src];
        a.vpc[pos * C + c] = jj > 0 ? a.pc[src - C] : -1;
    }
}

// ---- hysteresis / cooldown state machine over the blocks of one clip
// (detection.py:764-797).  One 64-lane wave per clip, lanes over channels.
struct SmArgs {
    Geom g;
    int64_t nb, n_clips, cap, cooldown;
    const int32_t* vis_j;            // [clips][nb] visited blocks, in order
    const int32_t *vfc, *vlb, *vpc;  // [clips][nb][C] their records (k_visits)
    const int32_t* nv;               // [clips] number of visits
    ofp_onset* records;              // [clips][cap]
    int64_t* counts;                 // [clips]
    int32_t clip_base;               // added to record.clip
};

// Visit-driven: the wave walks the compacted list of blocks that hold an upward crossing, a tile
// of visits at a time through LDS (coalesced, one tile ahead).  Between two visits nothing can
// fire: the cooldown counters advance in closed form (:780) and a latched channel is released iff
// some skipped block holds a row below its off threshold (:784-791 with on_indices.max() == 0),
// i.e. iff pc of the block before this visit is later than the previous visit.
constexpr int SM_NPL = 16;  // table entries per lane per tile (TB*C <= 64*SM_NPL)

__global__ __launch_bounds__(64) void k_state_machine(SmArgs a) {
    OFP_LATENCY_BOUND_KERNEL();
    extern __shared__ __align__(16) unsigned char smem[];
    const int C = a.g.C, B = a.g.B;
    const int64_t clip = blockIdx.x;
    const int lane = threadIdx.x;
    const int TB = max(1, min(64, (64 * SM_NPL) / C));  // visits per tile
    int64_t* deb = reinterpret_cast<int64_t*>(smem);          // [C]
    int32_t* onidx = reinterpret_cast<int32_t*>(deb + C);     // [C]
    int32_t* t_fc = onidx + C;                                // [TB*C]
    int32_t* t_lb = t_fc + TB * C;                            // [TB*C]
    int32_t* t_pc = t_lb + TB * C;                            // [TB*C]
    int32_t* t_j = t_pc + TB * C;                             // [64]
    uint8_t* state = reinterpret_cast<uint8_t*>(t_j + 64);    // [C]
    uint8_t* onflag = state + C;                              // [C]
    for (int c = lane; c < C; c += 64) {
        deb[c] = 0;
        state[c] = 0;
    }
    int64_t count = 0;
    ofp_onset* rec = a.records + clip * a.cap;
    const int64_t nvis = a.nv[clip];
    const int32_t* fc_g = a.vfc + clip * a.nb * C;
    const int32_t* lb_g = a.vlb + clip * a.nb * C;
    const int32_t* pc_g = a.vpc + clip * a.nb * C;
    const int32_t* vj_g = a.vis_j + clip * a.nb;
    int32_t rf[SM_NPL], rl[SM_NPL], rp[SM_NPL], rj = 0;
    auto load_tile = [&](int64_t k0) {
        const int64_t nvt = min<int64_t>(TB, nvis - k0);
        const int64_t n = nvt * C;
#pragma unroll
        for (int i = 0; i < SM_NPL; ++i) {
            const int64_t e = lane + 64 * i;
            rf[i] = e < n ? fc_g[k0 * C + e] : -1;
            rl[i] = e < n ? lb_g[k0 * C + e] : -1;
            rp[i] = e < n ? pc_g[k0 * C + e] : -1;
        }
        rj = lane < nvt ? vj_g[k0 + lane] : 0;
    };
    auto put_tile = [&]() {
#pragma unroll
        for (int i = 0; i < SM_NPL; ++i) {
            const int e = lane + 64 * i;
            if (e < TB * C) {
                t_fc[e] = rf[i];
                t_lb[e] = rl[i];
                t_pc[e] = rp[i];
            }
        }
        t_j[lane] = rj;
    };
    int jp = -1;  // previous visited block
    int64_t r_deb = 0;  // C <= 64: this lane's channel
    int r_state = 0;
    if (nvis > 0) load_tile(0);
    for (int64_t k0 = 0; k0 < nvis; k0 += TB) {
        __syncthreads();
        put_tile();
        __syncthreads();
        if (k0 + TB < nvis) load_tile(k0 + TB);
        const int nblk = (int)min<int64_t>(TB, nvis - k0);
        if (C <= 64) {
            // up to 64 channels: lane = channel, the channel's state lives in registers
            // (r_deb, r_state) instead of LDS -- the same steps as the general loop below
            const bool act = lane < C;
            for (int bi = 0; bi < nblk; ++bi) {
                const int j = t_j[bi];
                const int skipped = j - jp - 1;
                int f = -1, lbv = -1, pcv = -1;
                if (act) {
                    f = t_fc[bi * C + lane];
                    lbv = t_lb[bi * C + lane];
                    pcv = t_pc[bi * C + lane];
                }
                bool on = false;
                int oi = 0;
                if (act) {
                    if (skipped > 0 && r_deb > 0) r_deb -= (int64_t)B * min<int64_t>(skipped, (r_deb + B - 1) / B);
                    if (r_state && pcv > jp) r_state = 0;
                    const bool gate = !r_state && r_deb < 1;  // :764-768 (block-start values)
                    on = gate && f >= 0;
                    oi = on ? f : 0;                          // :774
                }
                int mx = 0;                                   // :790 on_indices.max() over ALL channels
                {
                    unsigned long long fm = __ballot(oi > 0);
                    while (fm) {
                        const int l = __builtin_ctzll(fm);
                        mx = max(mx, __builtin_amdgcn_readlane(oi, l));
                        fm &= fm - 1;
                    }
                }
                if (act) {
                    if (on) {                                 // :778-779
                        r_state = 1;
                        r_deb = a.cooldown;
                    }
                    if (r_deb > 0) r_deb -= B;                // :780
                    if (lbv >= mx) r_state = 0;               // :784-791
                }
                const unsigned long long m = __ballot(on);
                if (on) {
                    const int64_t pos = count + __popcll(m & ((1ull << lane) - 1ull));
                    if (pos < a.cap) {
                        rec[pos].clip = (int32_t)clip + a.clip_base;
                        rec[pos].channel = lane;
                        rec[pos].sample = (int64_t)j * B + oi;  // detection.py:80
                    }
                }
                count += __popcll(m);
                jp = j;
            }
            continue;
        }
        for (int bi = 0; bi < nblk; ++bi) {
            const int j = t_j[bi];
            const int32_t* fc = t_fc + bi * C;
            const int32_t* lb = t_lb + bi * C;
            const int32_t* pc = t_pc + bi * C;
            const int skipped = j - jp - 1;
            int mx = 0;
            for (int c = lane; c < C; c += 64) {
                // the blocks skipped since the previous visit
                int64_t d = deb[c];
                if (skipped > 0 && d > 0) d -= (int64_t)B * min<int64_t>(skipped, (d + B - 1) / B);
                deb[c] = d;
                uint8_t st = state[c];
                if (st && pc[c] > jp) st = 0;
                state[c] = st;
                // this block
                int f = fc[c];
                bool gate = !st && d < 1;                 // :764-768 (block-start values)
                bool on = gate && f >= 0;
                onflag[c] = on;
                int oi = on ? f : 0;                      // :774 argmax of an all-False column is 0
                onidx[c] = oi;
                mx = max(mx, oi);
            }
            // :790 on_indices.max() over ALL channels
            // (lanes without an onset hold 0; onsets are sparse, so read the firing lanes'
            // values one by one through scalar readlane instead of a six-step LDS shuffle tree)
            {
                unsigned long long fm = __ballot(mx > 0);
                int red = 0;
                while (fm) {
                    const int l = __builtin_ctzll(fm);
                    red = max(red, __builtin_amdgcn_readlane(mx, l));
                    fm &= fm - 1;
                }
                mx = red;
            }
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
                }
                unsigned long long m = __ballot(on);
                if (on) {
                    int64_t pos = count + __popcll(m & ((1ull << lane) - 1ull));
                    if (pos < a.cap) {
                        rec[pos].clip = (int32_t)clip + a.clip_base;
                        rec[pos].channel = c;
                        rec[pos].sample = (int64_t)j * B + onidx[c];  // detection.py:80
                    }
                }
                count += __popcll(m);
            }
            jp = j;
        }
    }
    if (lane == 0) a.counts[clip] = count;
}

// ---- the same machine, time-parallel over the visit list (long clips of up to 64 channels: C3 is ONE clip of
// 56 250 blocks, whose visits one wave walked in 5 ms).  The visits of a clip are cut into segments of SM_SEG;
// pass 0 gives every segment a start state from a warm-up over the SM_WARM visits before it (from the idle
// state -- the cooldown runs out within a few blocks and a latched channel is released at the first block with a
// row below its off threshold, so the machine forgets quickly), runs the segment and records its end state and
// number of onsets; the verification passes re-run exactly those segments whose start state differs from the
// end state of the segment before (chunk-Jacobi, as for the followers); offsets = prefix sums of the counts; a
// last pass writes the records.  The host enqueues everything without a synchronisation and reads the last
// pass's change counter with the call's final one: not converged (never seen) = the sequential kernel runs.
constexpr int SM_SEG = 256, SM_WARM = 64;

struct SmSegArgs {
    int32_t* used;    // [clips][n_seg][64][2] start state of every segment: {latched, cooldown counter}
    int32_t* endA;    // [clips][n_seg][64][2] end states, two buffers (pass parity)
    int32_t* endB;
    int32_t* cnt;     // [clips][n_seg] onsets of the segment, then (k_sm_offsets) its first record
    int64_t n_seg;
};

// visits [kb, ke) of `clip` for up to 64 channels (lane = channel), state in registers; WRITE: records at `count`
template <bool WRITE>
__device__ __forceinline__ void sm_walk64(const SmArgs& a, int64_t clip, int64_t kb, int64_t ke, int& jp, int64_t& r_deb,
                                          int& r_state, int64_t& count, int32_t* t_fc, int32_t* t_lb, int32_t* t_pc,
                                          int32_t* t_j, int TB) {
    const int C = a.g.C, B = a.g.B, lane = threadIdx.x;
    const int32_t* fc_g = a.vfc + clip * a.nb * C;
    const int32_t* lb_g = a.vlb + clip * a.nb * C;
    const int32_t* pc_g = a.vpc + clip * a.nb * C;
    const int32_t* vj_g = a.vis_j + clip * a.nb;
    ofp_onset* rec = a.records + clip * a.cap;
    const bool act = lane < C;
    for (int64_t k0 = kb; k0 < ke; k0 += TB) {
        const int nblk = (int)min<int64_t>(TB, ke - k0);
        __syncthreads();
        for (int e = lane; e < nblk * C; e += 64) {
            t_fc[e] = fc_g[k0 * C + e];
            t_lb[e] = lb_g[k0 * C + e];
            t_pc[e] = pc_g[k0 * C + e];
        }
        if (lane < nblk) t_j[lane] = vj_g[k0 + lane];
        __syncthreads();
        for (int bi = 0; bi < nblk; ++bi) {
            const int j = t_j[bi];
            const int skipped = j - jp - 1;
            int f = -1, lbv = -1, pcv = -1;
            if (act) {
                f = t_fc[bi * C + lane];
                lbv = t_lb[bi * C + lane];
                pcv = t_pc[bi * C + lane];
            }
            bool on = false;
            int oi = 0;
            if (act) {
                if (skipped > 0 && r_deb > 0) r_deb -= (int64_t)B * min<int64_t>(skipped, (r_deb + B - 1) / B);
                if (r_state && pcv > jp) r_state = 0;
                const bool gate = !r_state && r_deb < 1;  // :764-768 (block-start values)
                on = gate && f >= 0;
                oi = on ? f : 0;                          // :774
            }
            int mx = 0;                                   // :790 on_indices.max() over ALL channels
            {
                unsigned long long fm = __ballot(oi > 0);
                while (fm) {
                    const int l = __builtin_ctzll(fm);
                    mx = max(mx, __builtin_amdgcn_readlane(oi, l));
                    fm &= fm - 1;
                }
            }
            if (act) {
                if (on) {                                 // :778-779
                    r_state = 1;
                    r_deb = a.cooldown;
                }
                if (r_deb > 0) r_deb -= B;                // :780
                if (lbv >= mx) r_state = 0;               // :784-791
            }
            const unsigned long long m = __ballot(on);
            if (WRITE && on) {
                const int64_t pos = count + __popcll(m & ((1ull << lane) - 1ull));
                if (pos < a.cap) {
                    rec[pos].clip = (int32_t)clip + a.clip_base;
                    rec[pos].channel = lane;
                    rec[pos].sample = (int64_t)j * B + oi;  // detection.py:80
                }
            }
            count += __popcll(m);
            jp = j;
        }
    }
}

// pass 0: warm-up + segment; pass >= 1: verify / repair; pass < 0: write the records (offsets in s.cnt)
__global__ __launch_bounds__(64) void k_sm_seg(SmArgs a, SmSegArgs s, int pass, int* __restrict__ changed) {
    OFP_LATENCY_BOUND_KERNEL();
    extern __shared__ __align__(16) unsigned char smem[];
    const int C = a.g.C, lane = threadIdx.x;
    const int TB = max(1, min(64, 1024 / C));
    int32_t* t_fc = reinterpret_cast<int32_t*>(smem);
    int32_t* t_lb = t_fc + TB * C;
    int32_t* t_pc = t_lb + TB * C;
    int32_t* t_j = t_pc + TB * C;
    const int64_t seg = blockIdx.x, clip = blockIdx.y;
    const int64_t nvis = a.nv[clip];
    const int64_t k0 = seg * SM_SEG;
    if (k0 >= nvis) return;  // (empty segments have no state to pass on: nothing reads them)
    const int64_t k1 = min<int64_t>(k0 + SM_SEG, nvis);
    const int64_t si = ((clip * s.n_seg + seg) * 64 + lane) * 2;
    int32_t* end_prev = (pass & 1) ? s.endA : s.endB;   // pass p reads what pass p-1 wrote
    int32_t* end_next = (pass & 1) ? s.endB : s.endA;
    const int32_t* vj_g = a.vis_j + clip * a.nb;
    int r_state = 0;
    int64_t r_deb = 0, count = 0;
    int jp;
    if (pass == 0) {
        const int64_t kw = max<int64_t>(k0 - SM_WARM, 0);
        jp = kw > 0 ? vj_g[kw - 1] : -1;
        sm_walk64<false>(a, clip, kw, k0, jp, r_deb, r_state, count, t_fc, t_lb, t_pc, t_j, TB);
        s.used[si] = r_state;
        s.used[si + 1] = (int32_t)max<int64_t>(r_deb, 0);  // (every value <= 0 behaves like 0: one representative)
        count = 0;
        sm_walk64<false>(a, clip, k0, k1, jp, r_deb, r_state, count, t_fc, t_lb, t_pc, t_j, TB);
        s.endA[si] = r_state;           // pass 0 writes A (pass 1 reads A)
        s.endA[si + 1] = (int32_t)max<int64_t>(r_deb, 0);
        if (lane == 0) s.cnt[clip * s.n_seg + seg] = (int32_t)count;
        return;
    }
    if (pass > 0) {
        bool same = true;
        if (seg > 0) {
            const int64_t pi = ((clip * s.n_seg + seg - 1) * 64 + lane) * 2;
            const int32_t p0 = end_prev[pi], p1 = end_prev[pi + 1];
            same = s.used[si] == p0 && s.used[si + 1] == p1;
            same = __all(same);
            if (!same) {
                s.used[si] = p0;
                s.used[si + 1] = p1;
            }
        }
        if (same) {
            end_next[si] = end_prev[si];
            end_next[si + 1] = end_prev[si + 1];
            return;
        }
        if (lane == 0) atomicAdd(changed, 1);
        r_state = s.used[si];
        r_deb = s.used[si + 1];
        jp = vj_g[k0 - 1];
        sm_walk64<false>(a, clip, k0, k1, jp, r_deb, r_state, count, t_fc, t_lb, t_pc, t_j, TB);
        end_next[si] = r_state;
        end_next[si + 1] = (int32_t)max<int64_t>(r_deb, 0);
        if (lane == 0) s.cnt[clip * s.n_seg + seg] = (int32_t)count;
        return;
    }
    // write pass
    r_state = s.used[si];
    r_deb = s.used[si + 1];
    jp = k0 > 0 ? vj_g[k0 - 1] : -1;
    count = s.cnt[clip * s.n_seg + seg];  // first record of this segment (k_sm_offsets)
    sm_walk64<true>(a, clip, k0, k1, jp, r_deb, r_state, count, t_fc, t_lb, t_pc, t_j, TB);
}

// per clip: the segments' onset counts -> their first record; the clip's total
__global__ __launch_bounds__(64) void k_sm_offsets(SmArgs a, SmSegArgs s) {
    OFP_LATENCY_BOUND_KERNEL();
    const int64_t clip = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t n_used = cdiv((int64_t)a.nv[clip], SM_SEG);
    int32_t* t = s.cnt + clip * s.n_seg;
    long long base = 0;
    for (int64_t i0 = 0; i0 < n_used; i0 += 64) {
        const int64_t i = i0 + lane;
        const int v = i < n_used ? t[i] : 0;
        int inc = v;
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        if (i < n_used) t[i] = (int32_t)(base + inc - v);
        base += __shfl(inc, 63);
    }
    if (lane == 0) a.counts[clip] = base;
}

// ---- backtracking (detection.py:800-825 == envelope_follower.c:59-85 with the Python loop
// bound), one thread per onset.  The ring buffer of the reference (last N rows after writing
// the current block) is a window of the main relative envelope; rows before the stream start
// or outside the N-row window read as zero.
struct BtArgs {
    Geom g;
    const float* rel;  // planar
    const float* rel_il;  // or the interleaved output [clip][Nm][C] (tracker on the interleaved envelope: no planar copy)
    ofp_onset* records;
    const int64_t* counts;
    int64_t cap, n_clips, N;  // N = backtrack_buffer_size
    float alpha, tol;
};

__device__ __forceinline__ float bt_at(const BtArgs& a, int64_t clip, int64_t block_end, int64_t i, int c) {
    const int64_t m = block_end - i;  // buffer[-i]
    if (m < 0 || i > a.N) return 0.0f;
    if (a.rel_il) return a.rel_il[(clip * a.g.Nm + m) * a.g.C + c];
    return a.rel[(clip * a.g.C + c) * a.g.U + a.g.n_wb + m];
}

__global__ __launch_bounds__(64) void k_backtrack(BtArgs a) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t clip = id / a.cap, k = id % a.cap;
    if (clip >= a.n_clips) return;
    const int64_t n = min(a.counts[clip], a.cap);
    if (k >= n) return;
    ofp_onset& r = a.records[clip * a.cap + k];
    const int B = a.g.B;
    const int c = r.channel;
    const int64_t j = r.sample / B;
    int64_t delta = r.sample % B;
    const int64_t block_end = (j + 1) * B;
    const float omba = (float)(1.0 - (double)a.alpha);
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
constexpr int HP_MAX_ROUNDS = 16;     // IIR verification rounds enqueued ahead, at most
constexpr int AHEAD_MAX_PASSES = 24;  // follower / tracker: verifying passes enqueued ahead, at most (light groups count as one)
constexpr int OFP_N_COUNTERS = 512;  // int slots at the head of the zeroed region

struct Layout {
    Geom g;
    int64_t nb;
    int64_t hp_L, hp_W, hp_chunks, hp_delta;
    int hp_R, hp_S, hp_span;
    bool sm_seg;     // hysteresis machine time-parallel over the visit list (k_sm_seg; long clips of <= 64 channels)
    bool merge;      // followers / tracker: the two recurrences of a chunk in one lane (k_*_both; saturated launches)
    bool hp_early;   // whole runs stop early at a sub-chunk boundary (k_hp_run)
    bool hp_staged;  // candidates in stages with duplicate runs removed between them (k_hp_seg*)
    bool in_il;      // the IIR stage reads the caller's interleaved audio (no planar copy of the input is made)
    int64_t ar_L, ar_W, ar_Wc, ar_Wf, ar_chunks, ar_S;
    bool ar_sym;  // closed-form guess for the slow follower (k_ar_guess_sym)
    int64_t mm_L, mm_W, mm_chunks, mm_S;
    int tu;  // time steps per transpose tile
    // byte offsets
    int64_t o_xt, o_xdb, o_dif, o_relw, o_hp_U, o_hp_E, o_hp_sel, o_hp_done, o_hp_M, o_hp_nxt, o_hp_guess, o_hp_ran, o_hp_gs, o_hp_pos, o_hp_mrg, o_hp_runs, o_hp_goff, o_hp_stage_n, o_ar_state, o_ar_P, o_mm_state, o_mm_dirty, o_hp_rounds, o_pass_flags, o_sum,
        o_thr_mn, o_thr_mx, o_first, o_last, o_vflag, o_pc, o_visj, o_vrec, o_nv, o_vtile, o_ltile, o_smseg, o_flags, o_zero, zero_bytes, total;
};

int64_t pick(int64_t user, int64_t dflt) { return user > 0 ? user : dflt; }
int64_t pick_warm(int64_t user, int64_t dflt) { return user > 0 ? user : (user < 0 ? 0 : dflt); }

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
    g.Nv = align_up(g.n_w + N, 4);
    l.nb = g.Nm / g.B;
    // longest follower time constant in samples (coefficient = 1/samples)
    const float cmin = std::min(std::min(p.fast_attack, p.fast_release), std::min(p.slow_attack, p.slow_release));
    const double tau = cmin > 0 ? 1.0 / cmin : 1.0;
    // Defaults are for the latency regime (few chains: C2 has 8): many short chunks, 16 candidates
    // each, so that every SIMD holds about one wave.  A big batch (C3: 64 x 600 s, C4: 2048 chains)
    // has parallelism to spare and pays for every redundant step instead, so candidates are traded
    // for chunk length until the launch is about two waves per SIMD.
    const int64_t chains = n_clips * g.C;
    // concurrent_calls: the caller keeps that many calls of this size in flight; each lays its passes out for
    // its share of the GPU (the work-efficient layout of the larger batch they form together)
    const int64_t share = std::max<int64_t>(1, d->t.concurrent_calls);
    const int64_t cus = std::max<int64_t>(1, (int64_t)d->n_cus / share);
    const int64_t lane_budget = 2 * 4 * 64 * cus;
    int64_t hpL = 8192, hpR = 16, arL = 4096, mmL = 4096;
    if (d->t.hp_chunk <= 0 && d->t.hp_candidates <= 0) {
        if (chains * cdiv(g.V, hpL) * hpR > lane_budget) hpR = 8;  // fewer leave too many chain breaks
        // (up to 131 072 since round 3: 64 C2 clips per call, four calls in flight: 163 -> 172 M frames/s over the cap of 65 536)
        while (hpL < 131072 && chains * cdiv(g.V, hpL) * hpR > lane_budget) hpL *= 2;
    }
    // (followers: their chunk pass is the cheap part, the overlapping warm-up windows the expensive one, so the
    // chunks grow earlier -- at 3/4 of a wave per SIMD; measured on 512 clips x 4 ch: 6.2 -> 4.8 ms)
    while (arL < 32768 && chains * cdiv(g.U, arL) > (int64_t)3 * 64 * cus) arL *= 2;
    while (mmL < 32768 && chains * cdiv(g.U, mmL) > lane_budget) mmL *= 2;
    l.hp_L = pick(d->t.hp_chunk, hpL);
    l.hp_W = pick_warm(d->t.hp_warm, 40960);
    l.hp_R = (int)std::max<int64_t>(1, std::min<int64_t>(HP_MAXR, pick(d->t.hp_candidates, hpR)));
    // candidate starts 8 samples apart: distinct rounding histories (measured over nine inputs: as
    // few verification rounds as with ~1000), yet the 16 lanes of a chunk read 4 cache lines per
    // load instead of 16 and no lane runs much longer than W + L.  < 0: common start, see kernel.
    l.hp_delta = d->t.hp_candidate_offset < 0 ? 0 : pick(d->t.hp_candidate_offset, 8);
    l.hp_span = (d->t.hp_span == 2 || d->t.hp_span == 4) && l.hp_R % (int)d->t.hp_span == 0 ? (int)d->t.hp_span : 1;
    // Unset: a candidates launch of more than about one wave per SIMD is throughput-bound, and sharing a
    // warm-up between two chunks then does 2/3 of the steps in half the waves (measured, detector only:
    // 8 x C2 10.2 -> 8.9 ms, C4 36.6 -> 31.7 ms); a lone clip (720 waves) stays at 1, the latency setting.
    if (d->t.hp_span <= 0 && l.hp_R % 2 == 0 &&
        chains * cdiv(g.V, pick(d->t.hp_chunk, hpL)) * l.hp_R > (int64_t)64 * 4 * cus)
        l.hp_span = 2;
    // Staged candidates with duplicates removed (k_hp_seg0 / k_hp_dedupe / k_hp_seg / k_hp_seg_chunk): a third of
    // the steps in six launches instead of one -- for launches that are throughput-bound (the same condition as
    // the automatic span 2, which it replaces: every chunk then has its own R candidates again).
    // hp_dedupe: 0 auto, 1 always, < 0 never.
    {
        const bool busy = chains * cdiv(g.V, pick(d->t.hp_chunk, hpL)) * l.hp_R > (int64_t)64 * 4 * cus;
        // (auto = the caller said that calls overlap: in flight the stages are +4-5 % frames/s; for ONE call at a time they
        //  are no faster, and slower beside the STFT launch that shares the GPU with them -- C3 at full size 75.6
        //  against 72.2 ms)
        l.hp_staged = d->t.hp_dedupe > 0 || (d->t.hp_dedupe == 0 && d->t.hp_span <= 0 && busy && d->t.concurrent_calls >= 2);
        if (l.hp_W < 8192 || l.hp_R < 2) l.hp_staged = false;
        if (l.hp_staged) l.hp_span = 1;
    }
    // The caller's interleaved audio instead of a planar copy (staged candidates, 4 or 8 channels, high-pass on: without it
    // the dB pass reads the planar copy): only on request (tuning `interleaved` 2 / 3).  Measured, 48 C2 clips: the planar
    // copy costs 8.9 GB per call, but a lane that walks an interleaved series issues one 4-byte load per step instead of
    // one 16-byte load per four, and a wave's load instruction costs about as many cycles (~64) as the IIR step it feeds
    // (~66): k_hp_seg_chunk 3.7 -> 7.9 ms, k_hp_seg 1.9 -> 5.3 ms alone, 183 -> 168 M frames/s in flight.
    l.in_il = d->t.interleaved >= 2 && l.hp_staged && (g.C == 4 || g.C == 8) && p.hp_enabled;
    // sub-chunks: run in parallel once a chunk's start is verified (every candidate records its state at the inner
    // boundaries).  Long chunks (batches) get more of them, about 4096 samples each, and their whole runs from a
    // true start state stop at the first boundary where they have joined a candidate (k_hp_run, `early`): a break
    // then costs a few thousand sequential steps instead of the 32-64 k of the chunk.
    l.hp_S = (l.hp_L % (4 * 64) == 0) ? 4 : 1;
    if (l.hp_L >= 32768 && l.hp_L % (16 * 64) == 0) l.hp_S = (int)std::min<int64_t>(16, l.hp_L / 4096);
    l.hp_early = l.hp_S > 1 && l.hp_L / l.hp_S >= 4096 && d->t.hp_early >= 0;
    if (d->t.hp_early > 0 && l.hp_S > 1) l.hp_early = true;
    l.ar_L = pick(d->t.ar_chunk, arL);
    l.ar_W = pick_warm(d->t.ar_warm, align_up((int64_t)std::min(10.0 * tau, 4.0e6), 1024));
    l.ar_Wc = pick_warm(d->t.ar_coarse_warm, align_up((int64_t)std::min(14.0 * tau, 8.0e6), 1024));
    l.ar_Wf = l.ar_W;
    // symmetric slow follower: closed-form guess (k_ar_sym_local/combine) instead of the
    // sequential approximate pass.  The guess is good to ~10 ulps (it ignores the fp32 roundings
    // of the real trajectory) and a difference of a few ulps dies out like exp(-t/tau) only, so
    // the exact warm-up stays 11 tau: shorter ones leave chunk starts an ulp off (2 % at 5.5 tau)
    // and each repair pass costs more than the steps saved.  The fast follower starts from the
    // floor and needs 24 of its own time constants.
    const bool sym = p.slow_attack == p.slow_release && p.slow_attack > 0 && p.slow_attack < 1;
    l.ar_sym = d->t.ar_guess == 2 ? sym : (d->t.ar_guess == 1 ? false : sym);
    if (l.ar_sym) {
        const float cf = std::min(p.fast_attack, p.fast_release);
        const double tau_s = 1.0 / p.slow_attack, tau_f = cf > 0 ? 1.0 / cf : 1.0;
        l.ar_W = pick_warm(d->t.ar_warm, align_up((int64_t)std::min(std::max(11.0 * tau_s, 24.0 * tau_f), 4.0e6), 1024));
        l.ar_W = align_up(l.ar_W, l.ar_L);  // the guess is available at chunk boundaries
        l.ar_Wf = align_up((int64_t)std::min(24.0 * tau_f, 4.0e6), 1024);  // the fast follower's own window
        l.ar_Wc = pick_warm(d->t.ar_coarse_warm, align_up((int64_t)std::min(16.0 * tau_s, 8.0e6), 1024));
    }
    l.mm_L = pick(d->t.mm_chunk, mmL);
    l.mm_W = pick_warm(d->t.mm_warm, 49152);
    l.hp_chunks = std::max<int64_t>(1, cdiv(g.V, l.hp_L));
    l.ar_chunks = std::max<int64_t>(1, cdiv(g.U, l.ar_L));
    l.mm_chunks = std::max<int64_t>(1, cdiv(g.U, l.mm_L));
    // Span of the speculative warm-ups (k_ar_warm2 / k_mm_warm2): a run that walks on through S-1 more
    // chunks after its warm-up costs (W + (S-1) L) / S samples per chunk instead of W, but is a longer
    // dependent walk.  Model: the launch takes max(longest walk x time per step, bytes read / the rate
    // scattered 16-byte streams sustain); the largest S within 10 % of the best estimate is taken (a lone
    // clip stays latency-bound: S = 2; a batch reads up to 5x less).
    auto pick_span = [&](int64_t user, int64_t W, int64_t L, int64_t n_chunks, double ns_per_step) -> int64_t {
        if (user > 0) return std::min<int64_t>(user, std::max<int64_t>(n_chunks, 1));
        double best = 1e300, t[5];
        const int64_t cand[5] = {1, 2, 4, 8, 16};
        for (int i = 0; i < 5; ++i) {
            const double walk = (double)(W + (cand[i] - 1) * L);
            const double groups = (double)cdiv(n_chunks, cand[i]);
            const double t_path = walk * ns_per_step * 1e-9;
            const double t_mem = (double)share * (double)chains * groups * walk * 4.0 / 3.0e12;
            t[i] = std::max(t_path, t_mem);
            best = std::min(best, t[i]);
        }
        int64_t S = 1;
        for (int i = 0; i < 5; ++i)
            if (t[i] <= 1.10 * best && cand[i] <= std::max<int64_t>(n_chunks, 1)) S = cand[i];
        return S;
    };
    // Saturated launches (more than two waves per SIMD of this call's share): the fast / slow follower and the
    // tracker's min / max walk as ONE lane per chunk -- each line of the stream is then read once per pass instead of
    // twice at different times.  lane_merge: 0 auto, 1 always, < 0 never.
    // (measured, 16 x C2 / 512 x C4: four calls in flight +10 % / +5 % frames/s; ONE call at a time 11 % / 7 % slower --
    //  even a 2048-chain call is bound by its lanes' chains, not by bytes, when it has the GPU to itself: so auto
    //  means "the caller said that calls overlap", concurrent_calls >= 2)
    l.merge = d->t.lane_merge > 0 || (d->t.lane_merge == 0 && d->t.concurrent_calls >= 2);
    l.ar_S = l.ar_sym ? pick_span(d->t.ar_span, l.ar_W, l.ar_L, l.ar_chunks, 16.0) : 1;
    l.mm_S = pick_span(d->t.mm_span, l.mm_W, l.mm_L, l.mm_chunks, 11.0);
    l.tu = (int)std::max<int64_t>(1, std::min<int64_t>(256, 8192 / g.C));
    int64_t o = 0;
    auto take = [&](int64_t bytes) {
        int64_t r = o;
        o += align_up(bytes, 256);
        return r;
    };
    l.o_xt = take(n_clips * g.C * g.Nv * 4 + 64);
    l.o_xdb = take(n_clips * g.C * g.U * 4 + 64);
    l.o_dif = take(n_clips * g.C * g.U * 4 + 64);
    l.o_relw = take(n_clips * g.C * g.n_wb * 4 + 64);  // interleaved rel of the warm-up rows (tracker on the interleaved envelope)
    {
        const int64_t cc = n_clips * l.hp_chunks * g.C;
        l.o_hp_U = take(cc * (l.hp_R + 1) * 16);
        l.o_hp_E = take(cc * (l.hp_R + 1) * 16);
        l.o_hp_sel = take(cc);
        l.o_hp_done = take(cc * l.hp_S);
        l.o_hp_M = take(cc * l.hp_R * (l.hp_S - 1) * 16 + 16);
        l.o_hp_nxt = take(cc * (l.hp_R + 1));
        l.o_hp_guess = take(cc);
        l.o_hp_ran = take(cc);
        l.o_hp_gs = take(cc);
        l.o_hp_mrg = take(cc);
        // staged candidates: two work lists of {group, slot set, state} and the groups' ranges in the current one
        l.o_hp_runs = take(l.hp_staged ? 2 * cc * l.hp_R * 24 + 64 : 0);
        l.o_hp_goff = take(l.hp_staged ? 2 * cc * 4 : 0);
    }
    l.o_ar_state = take(3 * n_clips * l.ar_chunks * g.C * 2 * 4);
    l.o_ar_P = take(n_clips * l.ar_chunks * g.C * 8);
    l.o_mm_state = take(3 * n_clips * l.mm_chunks * g.C * 2 * 4);
    l.o_thr_mn = take(n_clips * l.nb * g.C * 4);
    l.o_thr_mx = take(n_clips * l.nb * g.C * 4);
    l.o_first = take(n_clips * l.nb * g.C * 4);
    l.o_last = take(n_clips * l.nb * g.C * 4);
    l.o_pc = take(n_clips * l.nb * g.C * 4);
    l.o_visj = take(n_clips * l.nb * 4);
    l.o_vrec = take(3 * n_clips * l.nb * g.C * 4);
    l.o_nv = take(n_clips * 4);
    l.o_vtile = take(n_clips * cdiv(std::max<int64_t>(l.nb, 1), 256) * 4);
    l.o_ltile = take(n_clips * g.C * cdiv(std::max<int64_t>(l.nb, 1), 256) * 4);
    // segmented state machine: long clips only (a batch of short clips has its parallelism across clips, and the
    // segments cost a few launches), at most 64 channels, cooldown counters that fit 32 bits
    l.sm_seg = d->t.sm_segments >= 0 && g.C <= 64 && p.cooldown < (1ll << 30) && g.B < (1 << 30) &&
               (l.nb >= 16384 || d->t.sm_segments > 0) && l.nb > SM_SEG;
    {
        const int64_t n_seg = cdiv(std::max<int64_t>(l.nb, 1), SM_SEG);
        l.o_smseg = take(l.sm_seg ? n_clips * n_seg * (3 * 64 * 2 * 4 + 4) : 0);
    }
    // everything that has to start a call as zero lies in one region: one fill per call
    l.o_flags = take(OFP_N_COUNTERS * 4);  // pass / round counters, one fresh slot per use
    l.o_zero = l.o_flags;
    l.o_hp_pos = take(n_clips * g.C * 4);
    l.o_hp_stage_n = take(16 * 4);  // runs left after each dedupe of the staged candidates
    l.o_mm_dirty = take(n_clips * l.mm_chunks * g.C);
    l.o_hp_rounds = take(2 * HP_MAX_ROUNDS * 4);  // IIR stage: {stuck, pending} of every round enqueued ahead
    l.o_pass_flags = take(2 * AHEAD_MAX_PASSES * 4);  // follower / tracker stage: change counter of every pass enqueued ahead
    l.o_vflag = take(n_clips * l.nb * 4);
    l.o_sum = take(2 * n_clips * l.nb * g.C * 4);  // block extremes for the crossing pass (k_rel_out -> k_block_scan)
    l.zero_bytes = o - l.o_zero;
    l.total = o;
    return l;
}

// Zeroed int counters for the verification passes: the region is cleared once per call and every
// use takes a fresh slot (the last slot is recycled with a fill when a call needs more).
struct Counters {
    int* base;
    int next;
    hipStream_t stream;
    int take(int n, int** out) {  // n <= 16 consecutive zeroed slots
        if (next + n <= OFP_N_COUNTERS - 16) {
            *out = base + next;
            next += n;
            return OFP_OK;
        }
        *out = base + OFP_N_COUNTERS - 16;
        OFP_HIP(hipMemsetAsync(*out, 0, 16 * sizeof(int), stream));
        return OFP_OK;
    }
};

// chunk-Jacobi driver: `chunk` kernel pass 0 (from used[]), then verification/repair passes
// until a pass changes nothing.  used[] has been filled by the stage's warm-up kernels.
template <class K, class A>
int run_jacobi(const char* name, K chunk, const A& args, int64_t n_threads, int64_t n_chunks, uint32_t* used,
               Counters& ctr, int* h_flags, int max_passes, int group, hipStream_t stream, int64_t* passes,
               int64_t* repaired,
               void (*light_pass)(const A&, int64_t, const uint32_t*, uint32_t*, uint32_t*, int*, const int*, hipStream_t) = nullptr,
               int64_t words = 0, int block = 64) {
    if (words == 0) words = n_threads * 2;  // state words per array
    uint32_t* endA = used + words;
    uint32_t* endB = endA + words;
    const unsigned grid = (unsigned)cdiv(n_threads, 64);
    int* d_changed = ctr.base;  // pass 0 counts nothing
    hipLaunchKernelGGL(chunk, dim3(grid), dim3(block), 0, stream, args, 0, n_threads, (const uint32_t*)endB, endA, used,
                       d_changed, (const int*)nullptr);
    OFP_LAUNCH_CHECK(name);
    *passes = 1;
    if (n_chunks == 1) return OFP_OK;  // a single chunk starts from the true state: exact
    uint32_t* prev = endA;
    uint32_t* next = endB;
    // Verification passes are enqueued `group` at a time with ONE host synchronisation per group: a pass
    // that finds nothing to repair costs microseconds on the GPU (every lane compares two words and
    // copies its end state), a host round trip costs tens.  Converged = a pass of the group changed nothing.
    group = std::max(1, std::min(group, 8));
    for (int pass = 1;;) {
        if (int rc = ctr.take(group, &d_changed)) return rc;
        for (int q = 0; q < group; ++q) {
            hipLaunchKernelGGL(chunk, dim3(grid), dim3(block), 0, stream, args, pass + q, n_threads, (const uint32_t*)prev,
                               next, used, d_changed + q, (const int*)nullptr);
            std::swap(prev, next);
        }
        OFP_LAUNCH_CHECK(name);
        OFP_HIP(hipMemcpyAsync(h_flags, d_changed, group * sizeof(int), hipMemcpyDeviceToHost, stream));
        OFP_HIP(hipStreamSynchronize(stream));
        int changed = 0;
        for (int q = 0; q < group; ++q) {
            changed = h_flags[q];
            *passes += 1;
            *repaired += changed;
            if (changed == 0) break;
        }
        pass += group;
        if (changed == 0) break;
        if (max_passes > 0 && pass > max_passes)
            return ofp::fail(OFP_ERR_NOCONVERGE, "%s: %d chunks still changing after %d passes", name, changed, pass - 1);
        if (light_pass) {  // a cascade is under way: light passes, eight per host round trip
            for (int grp = 0; grp < 4096; ++grp) {
                if (int rc = ctr.take(1, &d_changed)) return rc;
                for (int q = 0; q < 8; ++q) {
                    light_pass(args, n_threads, prev, next, used, d_changed, nullptr, stream);
                    std::swap(prev, next);
                }
                OFP_LAUNCH_CHECK(name);
                OFP_HIP(hipMemcpyAsync(h_flags, d_changed, sizeof(int), hipMemcpyDeviceToHost, stream));
                OFP_HIP(hipStreamSynchronize(stream));
                *repaired += h_flags[0];
                if (h_flags[0] == 0) break;
            }
        }
    }
    return OFP_OK;
}

// The same stage with NO host round trip (the default since round 3): pass 0 and `nv` verifying passes are enqueued
// ahead; every pass after the first verifying one is gated on the change counter of the pass it follows -- zero: that
// pass repaired nothing (and left both end arrays identical), so this one returns at once.  With `light_pass`, a group
// of eight light passes follows every second verifying pass, gated on it likewise.  flags[j-1] = chunks pass j
// repaired; the stage has converged iff flags[nv-1] == 0, which the host reads with the call's final synchronisation
// (not converged: the call is repeated in the host-verified form above).  flags: 2 * nv zeroed ints.
template <class K, class A>
int run_jacobi_ahead(const char* name, K chunk, const A& args, int64_t n_threads, int64_t n_chunks, uint32_t* used,
                     int* flags, int nv, hipStream_t stream,
                     void (*light_pass)(const A&, int64_t, const uint32_t*, uint32_t*, uint32_t*, int*, const int*, hipStream_t) = nullptr,
                     int64_t words = 0, int block = 64) {
    if (words == 0) words = n_threads * 2;
    uint32_t* endA = used + words;
    uint32_t* endB = endA + words;
    const unsigned grid = (unsigned)cdiv(n_threads, 64);
    hipLaunchKernelGGL(chunk, dim3(grid), dim3(block), 0, stream, args, 0, n_threads, (const uint32_t*)endB, endA, used,
                       flags, (const int*)nullptr);
    uint32_t* prev = endA;
    uint32_t* next = endB;
    if (n_chunks > 1) {
        for (int j = 1; j <= nv; ++j) {
            const int* gate = j > 1 ? flags + j - 2 : nullptr;
            hipLaunchKernelGGL(chunk, dim3(grid), dim3(block), 0, stream, args, j, n_threads, (const uint32_t*)prev, next, used,
                               flags + j - 1, gate);
            std::swap(prev, next);
            if (light_pass && j % 2 == 0 && j < nv) {
                for (int q = 0; q < 8; ++q) {
                    light_pass(args, n_threads, prev, next, used, flags + nv + j - 1, flags + j - 1, stream);
                    std::swap(prev, next);
                }
            }
        }
    }
    OFP_LAUNCH_CHECK(name);
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
    for (int k = 0; k < 10 && e == hipSuccess; ++k) e = hipEventCreate(&d->ev[k]);
    if (e == hipSuccess) e = hipHostMalloc((void**)&d->h_flags, 1024, hipHostMallocDefault);
    if (e == hipSuccess) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            d->n_cus = cus;
    }
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
    if (d->h_flags) (void)hipHostFree(d->h_flags);
    for (int k = 0; k < 10; ++k)
        if (d->ev[k]) (void)hipEventDestroy(d->ev[k]);
    delete d;
    return OFP_OK;
}

int ofp_detector_set_thresholds(ofp_detector* d, const double* on_threshold, const double* off_threshold) {
    OFP_REQUIRE(d && on_threshold && off_threshold, "ofp_detector_set_thresholds: NULL argument");
    const int C = d->p.n_channels;
    d->on.assign(on_threshold, on_threshold + C);
    d->off.assign(off_threshold, off_threshold + C);
    std::vector<float> onf(C), offf(C);
    for (int c = 0; c < C; ++c) {
        onf[c] = (float)on_threshold[c];
        offf[c] = (float)off_threshold[c];
    }
    // blocking copies: the arrays are read by launches the caller enqueues afterwards
    OFP_HIP(hipMemcpy(d->d_on_f, onf.data(), C * sizeof(float), hipMemcpyHostToDevice));
    OFP_HIP(hipMemcpy(d->d_off_f, offf.data(), C * sizeof(float), hipMemcpyHostToDevice));
    OFP_HIP(hipMemcpy(d->d_on_d, d->on.data(), C * sizeof(double), hipMemcpyHostToDevice));
    return OFP_OK;
}

int ofp_detector_set_tuning(ofp_detector* d, const ofp_detect_tuning* t) {
    OFP_REQUIRE(d && t, "ofp_detector_set_tuning: NULL argument");
    d->t = *t;
    return OFP_OK;
}

const float* ofp_detect_planar_input(const ofp_detector* d, int64_t n_clips, int64_t n_samples, int64_t warm,
                                     const void* d_ws) {
    if (!d || !d_ws || n_clips < 1 || n_samples < 0) return nullptr;
    const Layout l = make_layout(d, n_clips, n_samples, warm);
    if (l.in_il) return nullptr;  // no planar copy in this layout: read the caller's own array
    return reinterpret_cast<const float*>(static_cast<const char*>(d_ws) + l.o_xt) + l.g.n_w;
}

int64_t ofp_detect_planar_stride(const ofp_detector* d, int64_t n_clips, int64_t n_samples, int64_t warm) {
    if (!d || n_clips < 1 || n_samples < 0) return -1;
    const Layout l = make_layout(d, n_clips, n_samples, warm);
    return l.in_il ? 0 : l.g.Nv;  // 0: no planar copy (see ofp_detect_planar_input)
}

int64_t ofp_detect_workspace_bytes(const ofp_detector* d, int64_t n_clips, int64_t n_samples,
                                   int64_t warm) {
    if (!d || n_clips < 0 || n_samples < 0) return -1;
    return make_layout(d, n_clips, n_samples, warm).total;
}

// phase 0: everything; phase 1: only the asynchronous head (transpose + candidates launch);
// phase 2: everything after the head (the caller ran phase 1 with the same arguments);
// phase 5 + phase 6: phase 1 in two calls, the planar input copy / the IIR candidate launch;
// phase 7: completion of a call that was only enqueued (run_mode 1), after the caller has synchronised.
// run_mode 0: enqueue, synchronise, complete; 1: enqueue only (phases 0 and 2; nothing in it blocks or reads
// device results on the host, so the stream may be capturing a hipGraph).
// from_stage (with phase 2, by the completion's fall-back only): 1 = repeat the call from the follower stage on, 2 = from
// the tracker stage on -- everything before it (the filtered dB stream / the relative envelope) is in the work space.
static int detect_impl(ofp_detector* d, const float* d_x, int64_t n_clips, int64_t N, int64_t warm,
                       float* d_rel, ofp_onset* d_records, int64_t cap, int64_t* d_counts,
                       void* d_ws, int64_t ws_bytes, int64_t* h_info, void* stream_, int phase, int run_mode = 0,
                       int from_stage = 0) {
    OFP_REQUIRE(d && d_counts && d_ws, "ofp_detect_offline: NULL argument");
    OFP_REQUIRE(n_clips >= 1 && N >= 0 && cap >= 0, "ofp_detect_offline: bad sizes");
    OFP_REQUIRE(d_x || N == 0, "ofp_detect_offline: d_x is NULL");
    OFP_REQUIRE(d_records || cap == 0, "ofp_detect_offline: d_records is NULL");
    OFP_REQUIRE(n_clips <= 65535, "ofp_detect_offline: at most 65535 clips per call");
    hipStream_t stream = (hipStream_t)stream_;
    const Layout l = make_layout(d, n_clips, N, warm);
    if (ws_bytes < l.total)
        return ofp::fail(OFP_ERR_WORKSPACE, "work space %lld < required %lld bytes", (long long)ws_bytes,
                         (long long)l.total);
    const Geom& g = l.g;
    const auto& p = d->p;
    const bool host_verify = d->t.host_verify == 1;   // (2 / 3, experiments: the IIR stage only / the followers and the tracker only)
    const bool hv_hp = host_verify || d->t.host_verify == 2, hv_fm = host_verify || d->t.host_verify == 3;
    const bool enqueue_only = run_mode == 1;
    OFP_REQUIRE(!(enqueue_only && d->t.host_verify > 0), "ofp_detect_offline_enqueue: not with tuning host_verify (its passes are "
                "verified on the host)");
    unsigned char* ws = static_cast<unsigned char*>(d_ws);
    float* xt = reinterpret_cast<float*>(ws + l.o_xt);
    float* xdb = reinterpret_cast<float*>(ws + l.o_xdb);
    float* dif = reinterpret_cast<float*>(ws + l.o_dif);
    Counters ctr{reinterpret_cast<int*>(ws + l.o_flags), 0, stream};
    int* pass_flags = reinterpret_cast<int*>(ws + l.o_pass_flags);  // follower passes at 0, tracker passes at AHEAD_MAX_PASSES
    ofp_detect_pending& pend = d->pend;
    int64_t info[OFP_DETECT_INFO_LEN] = {0};
    hipEvent_t* ev = d->ev;
    const bool do_head = phase == 0 || phase == 1 || phase == 5;   // zero fill + transpose
    const bool do_cand = phase == 0 || phase == 1 || phase == 6;   // the IIR candidate launch
    // stage timing by HIP events -- not while the stream is capturing (an event recorded into a graph has no time)
    bool timed = true;
    if (phase != 7) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) timed = false;
        if (!timed) OFP_REQUIRE(enqueue_only || phase == 1 || phase == 5 || phase == 6,
                                "ofp_detect_offline: the stream is capturing; use ofp_detect_offline_enqueue");
        if (do_head) pend.timed = timed;
        else timed = timed && pend.timed;
    }
    if (from_stage > 0) timed = pend.timed = false;
    if (do_head && timed) OFP_HIP(hipEventRecord(ev[0], stream));
    if (l.nb == 0) {  // fewer samples than one block: nothing is processed (detection.py:74-75)
        if (phase == 1 || phase == 5 || phase == 6) return OFP_OK;
        if (phase != 7) {
            hipLaunchKernelGGL(k_zero_i64, dim3((unsigned)cdiv(n_clips, 256)), dim3(256), 0, stream, d_counts, n_clips);
            OFP_LAUNCH_CHECK("k_zero_i64");
        }
        if (enqueue_only) {
            pend.valid = true;
            pend.empty = true;
            return OFP_OK;
        }
        if (phase != 7) OFP_HIP(hipStreamSynchronize(stream));
        pend.valid = false;
        if (h_info) std::memcpy(h_info, info, sizeof(info));
        return OFP_OK;
    }
    const int64_t chains = n_clips * g.C;
    const int64_t n_elem = chains * g.U;
    const unsigned ew_grid = (unsigned)std::min<int64_t>(cdiv(n_elem, 256), 256 * 16);
    const size_t tile_lds = (size_t)g.C * (l.tu + 4) * sizeof(float);
    const float* rel = dif;  // (the relative envelope overwrites the follower difference in place)
    // tracker, crossing pass and backtracking on the interleaved envelope (see k_mm_warm_il): the planar copy is not written
    const bool mm_il = d->t.interleaved >= 0 && d->t.interleaved != 2 && l.merge && d_rel != nullptr && !p.manual &&
                       (g.C == 4 || g.C == 8 || g.C == 64);
    float* rel_warm = reinterpret_cast<float*>(ws + l.o_relw);

    // --- the last stage's arguments, needed by the completion as well (sequential machine as the fall-back)
    int32_t* va_nv = reinterpret_cast<int32_t*>(ws + l.o_nv);
    SmArgs sm;
    sm.g = g;
    sm.nb = l.nb;
    sm.n_clips = n_clips;
    sm.cap = cap;
    sm.cooldown = p.cooldown;
    sm.vis_j = reinterpret_cast<int32_t*>(ws + l.o_visj);
    sm.vfc = reinterpret_cast<int32_t*>(ws + l.o_vrec);
    sm.vlb = sm.vfc + n_clips * l.nb * g.C;
    sm.vpc = sm.vlb + n_clips * l.nb * g.C;
    sm.nv = va_nv;
    sm.records = d_records;
    sm.counts = d_counts;
    sm.clip_base = 0;
    auto sequential_machine = [&]() -> int {
        const int tb = std::max(1, std::min(64, (64 * SM_NPL) / g.C));
        size_t lds = (size_t)g.C * (8 + 4 + 1 + 1) + (size_t)3 * tb * g.C * 4 + 64 * 4 + 16;
        static ofp::LdsAttrCache attr;
        if (int rc = ofp::ensure_dynamic_lds(reinterpret_cast<const void*>(k_state_machine), lds, attr)) return rc;
        hipLaunchKernelGGL(k_state_machine, dim3((unsigned)n_clips), dim3(64), lds, stream, sm);
        OFP_LAUNCH_CHECK("k_state_machine");
        return OFP_OK;
    };
    auto backtrack = [&]() -> int {
        if (!(p.backtrack && cap > 0)) return OFP_OK;
        BtArgs bt;
        bt.g = g;
        bt.rel = rel;
        bt.rel_il = mm_il ? d_rel : nullptr;
        bt.records = d_records;
        bt.counts = d_counts;
        bt.cap = cap;
        bt.n_clips = n_clips;
        bt.N = p.backtrack_buffer_size;
        bt.alpha = p.backtrack_alpha;
        bt.tol = p.backtrack_tol;
        hipLaunchKernelGGL(k_backtrack, dim3((unsigned)cdiv(n_clips * cap, 64)), dim3(64), 0, stream, bt);
        OFP_LAUNCH_CHECK("k_backtrack");
        return OFP_OK;
    };
    // after the final synchronisation: the segmented machine's verdict, the pass statistics, the stage times
    auto complete = [&]() -> int {
        // (pend stays valid: a graph that captured the enqueued call may be replayed and completed any number of times)
        std::memcpy(info, pend.info, sizeof(info));
        if (pend.sm_flag && (d->h_flags[40] != 0 || d->t.sm_segments == 2)) {  // (2: tests exercise this path)
            // the last verification pass still changed a segment's start state (a machine that does not forget within
            // four segments): the sequential machine decides
            if (int rc = sequential_machine()) return rc;
            if (int rc = backtrack()) return rc;
            if (pend.timed) OFP_HIP(hipEventRecord(ev[6], stream));
            OFP_HIP(hipStreamSynchronize(stream));
            info[14] = 1;
        }
        {
            const int* pf = d->h_flags + 96;   // change counters of the follower (0..) / tracker (AHEAD_MAX_PASSES..) passes
            const int* hr = d->h_flags + 64;   // {stuck, pending} per enqueued IIR round
            bool hp_open = false, ar_err = false, mm_err = false;
            if (pend.hp_rounds > 0) {  // (0: no high-pass, or its rounds were verified on the host)
                int used_rounds = 1;
                while (used_rounds < pend.hp_rounds && hr[2 * (used_rounds - 1)] + hr[2 * (used_rounds - 1) + 1] != 0) ++used_rounds;
                for (int q = 0; q < used_rounds; ++q) info[3] += hr[2 * q] + hr[2 * q + 1];
                hp_open = hr[2 * (pend.hp_rounds - 1)] + hr[2 * (pend.hp_rounds - 1) + 1] != 0;
                // the next call enqueues what this one needed plus a margin (decaying slowly)
                d->hp_rounds_hint = std::max(used_rounds + 2, d->hp_rounds_hint - 1);
                info[0] = used_rounds;
            }
            if (pend.ahead) {
                // passes that ran: pass 0, the first verifying one, and every further one whose predecessor repaired
                auto ran = [&](const int* f, int nv, int* hint, int64_t* passes) -> bool {
                    if (nv == 0) {
                        *passes = 1;
                        return false;
                    }
                    int used_v = 1;
                    while (used_v < nv && f[used_v - 1] != 0) ++used_v;
                    for (int j = 0; j < used_v; ++j) info[3] += f[j];
                    for (int j = nv; j < 2 * nv; ++j) info[3] += f[j];  // (the light groups)
                    *passes = 1 + used_v;
                    *hint = std::max(used_v + 1, *hint - 1);
                    return f[nv - 1] != 0;
                };
                ar_err = ran(pf, pend.ar_nv, &d->ar_pass_hint, &info[1]);
                mm_err = ran(pf + AHEAD_MAX_PASSES, pend.mm_nv, &d->mm_pass_hint, &info[2]);
                if (p.manual) info[2] = 0;
            }
            if (hp_open || ar_err || mm_err || d->t.host_verify <= -2) {  // (-2 / -3 / -4: tests exercise these paths)
                // What was enqueued ahead did not suffice: the call again from the first stage that has not converged,
                // its passes verified on the host -- everything (the IIR rounds), from the follower stage on (the dB
                // stream is final) or from the tracker stage on (the relative envelope is final).
                const int from = (hp_open || d->t.host_verify == -2) ? 0 : ((ar_err || d->t.host_verify == -3) ? 1 : 2);
                const ofp_detect_tuning keep = d->t;
                const ofp_detect_pending keep_pend = pend;  // (a graph that captured this call may be replayed again)
                d->t.host_verify = 1;
                if (hp_open) d->hp_rounds_hint = std::min(HP_MAX_ROUNDS, 2 * pend.hp_rounds);
                if (ar_err) d->ar_pass_hint = std::min(AHEAD_MAX_PASSES / 2, 2 * pend.ar_nv);
                if (mm_err) d->mm_pass_hint = std::min(AHEAD_MAX_PASSES / 2, 2 * pend.mm_nv);
                const int rc = detect_impl(d, d_x, n_clips, N, warm, d_rel, d_records, cap, d_counts, d_ws, ws_bytes, h_info,
                                           stream_, from == 0 ? 0 : 2, 0, from);
                d->t = keep;
                d->pend = keep_pend;
                if (rc == OFP_OK && h_info) {
                    if (from >= 1) h_info[0] = info[0];   // (the stages that were not repeated keep their figures)
                    if (from >= 2) h_info[1] = info[1];
                    h_info[15] = 1 + (hp_open ? 1 : 0) + (ar_err ? 2 : 0) + (mm_err ? 4 : 0);  // (see include/onsetfp.h)
                }
                return rc;
            }
        }
        if (pend.staged) {  // distinct runs that walked each later segment / the chunk (slightly over: clamped windows)
            const int* n = d->h_flags + 16;
            for (int m = 0; m + 1 < pend.n_cuts; ++m) info[12] += (int64_t)n[m] * (pend.cuts[m + 1] - pend.cuts[m]);
            info[12] += (int64_t)n[pend.n_cuts - 1] * l.hp_L;
            info[13] = n[pend.n_cuts - 1];  // runs that walked a chunk (of chains * chunks * R candidates)
        }
        if (pend.timed) {  // stage durations in nanoseconds (HIP events on the launch stream)
            for (int k = 0; k < 6; ++k) {
                float ms = 0.0f;
                OFP_HIP(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
                info[4 + k] = (int64_t)(ms * 1.0e6);
            }
            float ms = 0.0f;
            OFP_HIP(hipEventElapsedTime(&ms, ev[0], ev[6]));
            info[10] = (int64_t)(ms * 1.0e6);
            if (pend.hp_timed) {  // the IIR candidate launch(es)
                OFP_HIP(hipEventElapsedTime(&ms, ev[8], ev[7]));
                info[11] = (int64_t)(ms * 1.0e6);
            }
        }
        if (h_info) std::memcpy(h_info, info, sizeof(info));
        return OFP_OK;
    };
    if (phase == 7) {
        OFP_REQUIRE(pend.valid, "ofp_detect_offline_complete: no enqueued call is pending on this detector");
        if (pend.empty) {
            if (h_info) std::memcpy(h_info, info, sizeof(info));
            return OFP_OK;
        }
        return complete();
    }
    if (do_head) {
        pend.hp_timed = false;
        pend.staged = false;
        pend.empty = false;
        std::memset(pend.info, 0, sizeof(pend.info));
    }

    // --- transpose in
    if (do_head) {
        {  // counters, flags: see make_layout (regions are 256-byte multiples)
            const int64_t n16 = l.zero_bytes / 16;
            hipLaunchKernelGGL(k_zero, dim3((unsigned)cdiv(n16, 256)), dim3(256), 0, stream, reinterpret_cast<uint4*>(ws + l.o_zero), n16);
            OFP_LAUNCH_CHECK("k_zero");
        }
        if (!l.in_il) {
            const int64_t n_tiles = cdiv(N, l.tu);
            const unsigned gx = (unsigned)std::min<int64_t>(n_tiles, std::max<int64_t>(1, (int64_t)32 * d->n_cus / n_clips));
            hipLaunchKernelGGL(k_transpose_in, dim3(gx, (unsigned)n_clips), dim3(256), tile_lds,
                               stream, d_x, xt, N, g.C, l.tu, g.n_w, g.Nv, n_tiles);
            OFP_LAUNCH_CHECK("k_transpose_in");
        }
        if (timed) OFP_HIP(hipEventRecord(ev[8], stream));
    }
    if (phase == 5) return OFP_OK;

    if (from_stage > 0) {  // the words the repeated stages count in / flag must start as zero again
        auto zero = [&](int64_t off, int64_t bytes) {
            const int64_t n16 = align_up(bytes, 256) / 16;
            hipLaunchKernelGGL(k_zero, dim3((unsigned)cdiv(n16, 256)), dim3(256), 0, stream, reinterpret_cast<uint4*>(ws + off), n16);
        };
        zero(l.o_flags, OFP_N_COUNTERS * 4);
        zero(l.o_mm_dirty, n_clips * l.mm_chunks * g.C);
        zero(l.o_vflag, n_clips * l.nb * 4);
        if (from_stage == 1) zero(l.o_sum, 2 * n_clips * l.nb * g.C * 4);
        OFP_LAUNCH_CHECK("k_zero");
    }

    // --- hp + dB
    if (p.hp_enabled && from_stage == 0) {
        HpCand hc;
        hc.st.g = g;
        hc.st.xt = xt;
        hc.st.x_il = d_x;
        hc.st.out = xdb;
        const int ch = l.in_il ? g.C : 0;
        const auto kseg0 = ch == 8 ? k_hp_seg0<8> : (ch == 4 ? k_hp_seg0<4> : k_hp_seg0<0>);
        const auto kseg = ch == 8 ? k_hp_seg<8> : (ch == 4 ? k_hp_seg<4> : k_hp_seg<0>);
        const auto kseg_chunk = ch == 8 ? k_hp_seg_chunk<8> : (ch == 4 ? k_hp_seg_chunk<4> : k_hp_seg_chunk<0>);
        // complete-line stores for the output walk (k_hp_run_lines): throughput layout, planar input, every position that
        // matters a multiple of 32 steps
        const bool hp_lines = d->t.line_stores >= 0 && ch == 0 &&
                              (l.merge || d->t.line_stores > 0 || chains * l.hp_chunks * l.hp_S >= (int64_t)2 * 64 * 4 * d->n_cus) &&  // (calls in flight, or a big one) (g.n_w & 31) == 0 && (g.n_wb & 31) == 0 &&
                              (g.V & 31) == 0 && (g.U & 31) == 0 && (g.Nv & 3) == 0 && (l.hp_L & 31) == 0 &&
                              ((l.hp_L / l.hp_S) & 31) == 0 && l.hp_L % l.hp_S == 0;
        const auto krun = hp_lines ? k_hp_run_lines : (ch == 8 ? k_hp_run<8> : (ch == 4 ? k_hp_run<4> : k_hp_run<0>));
        std::memcpy(hc.st.b, d->b, sizeof(hc.st.b));
        std::memcpy(hc.st.a, d->a, sizeof(hc.st.a));
        hc.st.L = l.hp_L;
        hc.st.W = l.hp_W;
        hc.st.n_chunks = l.hp_chunks;
        hc.R = l.hp_R;
        hc.delta = l.hp_delta;
        hc.span = l.hp_span;
        hc.U = reinterpret_cast<uint32_t*>(ws + l.o_hp_U);
        hc.E = reinterpret_cast<uint32_t*>(ws + l.o_hp_E);
        hc.sel = reinterpret_cast<int8_t*>(ws + l.o_hp_sel);
        hc.done = reinterpret_cast<uint8_t*>(ws + l.o_hp_done);
        hc.S = l.hp_S;
        hc.M = reinterpret_cast<uint32_t*>(ws + l.o_hp_M);
        hc.nxt = reinterpret_cast<uint8_t*>(ws + l.o_hp_nxt);
        hc.guessed = reinterpret_cast<uint8_t*>(ws + l.o_hp_guess);
        hc.ran = reinterpret_cast<int8_t*>(ws + l.o_hp_ran);
        hc.gs = reinterpret_cast<int8_t*>(ws + l.o_hp_gs);
        hc.mrg = reinterpret_cast<int8_t*>(ws + l.o_hp_mrg);
        hc.early = l.hp_early ? 1 : 0;
        hc.counters = ctr.base;
        hc.pos = reinterpret_cast<int32_t*>(ws + l.o_hp_pos);
        hc.probe = nullptr;
        hc.prev = nullptr;
        const char* probe_path = (do_cand && timed) ? getenv("OFP_HP_PROBE") : nullptr;
        const int64_t probe_waves = cdiv(chains * l.hp_chunks * (l.hp_R / l.hp_span), 64);
        if (probe_path) OFP_HIP(hipMalloc(&hc.probe, probe_waves * 32));
        const int64_t nA = chains * l.hp_chunks * (hc.R / hc.span);
        const int64_t nM = chains * l.hp_chunks * (hc.R + 1);
        const int64_t nC = chains * l.hp_chunks * l.hp_S;
        const int64_t nC0 = chains * l.hp_chunks;
        // staged candidates: window offsets at which duplicates are removed (the last one is the chunk start)
        int64_t cuts[16];
        int n_cuts = 0;
        if (l.hp_staged) {
            // (the runs of a group merge fastest early on: 8 -> 4.8 distinct within 8 192 steps, -> 2.2 by 24 576)
            for (int64_t c = 4096; c < l.hp_W && n_cuts < 15; c += (c < 8192 ? 4096 : (c < 40960 ? 8192 : 16384)))
                cuts[n_cuts++] = c;
            cuts[n_cuts++] = l.hp_W;
        }
        int* stage_n = reinterpret_cast<int*>(ws + l.o_hp_stage_n);
        if (do_cand && l.hp_staged) {
            const int64_t n0 = nC0 * hc.R;
            OFP_REQUIRE(n0 < (1ll << 31), "ofp_detect_offline: %lld speculative runs in one call", (long long)n0);
            HpRuns rl[2];
            // layout of the two lists: z (16 B) of both first, then grp, then mask
            unsigned char* rb = ws + l.o_hp_runs;
            rl[0].z = reinterpret_cast<float4*>(rb);
            rl[1].z = reinterpret_cast<float4*>(rb + n0 * 16);
            rl[0].grp = reinterpret_cast<int32_t*>(rb + n0 * 32);
            rl[1].grp = reinterpret_cast<int32_t*>(rb + n0 * 36);
            rl[0].mask = reinterpret_cast<uint32_t*>(rb + n0 * 40);
            rl[1].mask = reinterpret_cast<uint32_t*>(rb + n0 * 44);
            int32_t* goff = reinterpret_cast<int32_t*>(ws + l.o_hp_goff);
            int32_t* gcnt = goff + nC0;
            const unsigned full_grid = (unsigned)cdiv(n0, HP_CAND_THREADS);
            hipLaunchKernelGGL(kseg0, dim3(full_grid), dim3(HP_CAND_THREADS), 0, stream, hc, rl[0], goff, gcnt, cuts[0], n0);
            OFP_LAUNCH_CHECK("k_hp_seg0");
            int cur = 0;
            for (int m = 0; m < n_cuts; ++m) {
                hipLaunchKernelGGL(k_hp_dedupe, dim3((unsigned)cdiv(nC0 * 16, 256)), dim3(256), 0, stream, rl[cur], rl[cur ^ 1],
                                   goff, gcnt, nC0, stage_n + m);
                cur ^= 1;
                // (the grid is sized for the worst case, every run distinct; the lanes beyond the list leave at once)
                if (m + 1 < n_cuts)
                    hipLaunchKernelGGL(kseg, dim3(full_grid), dim3(HP_CAND_THREADS), 0, stream, hc, rl[cur],
                                       (const int*)(stage_n + m), cuts[m], cuts[m + 1]);
                else
                    hipLaunchKernelGGL(kseg_chunk, dim3(full_grid), dim3(HP_CAND_THREADS), 0, stream, hc, rl[cur],
                                       (const int*)(stage_n + m));
            }
            OFP_LAUNCH_CHECK("k_hp_dedupe / k_hp_seg / k_hp_seg_chunk");
            if (timed) OFP_HIP(hipEventRecord(ev[7], stream));
        } else if (do_cand) {
            const unsigned cand_grid = (unsigned)cdiv(nA, HP_CAND_THREADS);
            if ((int64_t)cand_grid <= d->n_cus && d->t.concurrent_calls <= 1)
                hipLaunchKernelGGL(k_hp_candidates<true>, dim3(cand_grid), dim3(HP_CAND_THREADS), 0, stream, hc, nA);
            else
                hipLaunchKernelGGL(k_hp_candidates<false>, dim3(cand_grid), dim3(HP_CAND_THREADS), 0, stream, hc, nA);
            OFP_LAUNCH_CHECK("k_hp_candidates");
            if (timed) OFP_HIP(hipEventRecord(ev[7], stream));
            if (hc.probe) {  // diagnostics: where and when every wave of the launch ran (tools/wave_placement.py)
                std::vector<unsigned long long> h(probe_waves * 4);
                OFP_HIP(hipStreamSynchronize(stream));
                OFP_HIP(hipMemcpy(h.data(), hc.probe, probe_waves * 32, hipMemcpyDeviceToHost));
                (void)hipFree(hc.probe);
                hc.probe = nullptr;
                if (FILE* f = fopen(probe_path, "w")) {
                    for (int64_t w = 0; w < probe_waves; ++w)
                        fprintf(f, "%lld %llu %llu %llu %llu\n", (long long)w, h[4 * w], h[4 * w + 1], h[4 * w + 2], h[4 * w + 3]);
                    fclose(f);
                }
            }
        }
        if (do_cand) {
            pend.hp_timed = timed;
            pend.staged = l.hp_staged;
            pend.n_cuts = n_cuts;
            std::memcpy(pend.cuts, cuts, sizeof(cuts));
            int64_t steps = 0;
            if (l.hp_staged) {  // stage 0 here; the later stages once their run counts are on the host (completion)
                for (int64_t j = 0; j < l.hp_chunks; ++j)
                    for (int r = 0; r < hc.R; ++r)
                        steps += std::max<int64_t>(j * l.hp_L - l.hp_W + cuts[0], 0) -
                                 std::max<int64_t>(j * l.hp_L - l.hp_W - r * hc.delta, 0);
            } else {  // IIR steps this launch executes over all its lanes (the speculation's redundant work)
                const int Rm = hc.R / hc.span;
                for (int64_t j = 0; j < l.hp_chunks; ++j) {
                    const int64_t run_end = std::min<int64_t>((j + hc.span) * l.hp_L, g.V);
                    for (int r = 0; r < Rm; ++r)
                        steps += run_end - std::max<int64_t>(j * l.hp_L - l.hp_W - r * hc.delta, 0);
                }
            }
            pend.info[12] = steps * chains;
        }
        if (phase == 1 || phase == 6) return OFP_OK;
        if (l.hp_staged)
            OFP_HIP(hipMemcpyAsync(d->h_flags + 16, stage_n, 16 * sizeof(int), hipMemcpyDeviceToHost, stream));
        if (!hv_hp) {
            // A fixed number of rounds enqueued ahead, no host round trip: a round whose predecessor left nothing
            // open returns at once (a few microseconds).  The count follows what the detector's recent calls needed
            // (+2); the last round's counters are read with the final synchronisation, and a call that has not
            // converged by then (never seen with the margin) is repeated in the host-verified form.
            hipLaunchKernelGGL(k_hp_plurality, dim3((unsigned)cdiv(nC0 * 16, 256)), dim3(256), 0, stream, hc, nC0);
            OFP_LAUNCH_CHECK("k_hp_plurality");
            const int NR = std::max(3, std::min(HP_MAX_ROUNDS, d->hp_rounds_hint));
            int* c = reinterpret_cast<int*>(ws + l.o_hp_rounds);
            for (int q = 0; q < NR; ++q) {
                hc.counters = c + 2 * q;
                hc.prev = q > 0 ? c + 2 * (q - 1) : nullptr;
                hipLaunchKernelGGL(k_hp_match, dim3((unsigned)cdiv(nM, 256)), dim3(256), 0, stream, hc, nM);
                hipLaunchKernelGGL(k_hp_resolve, dim3((unsigned)chains), dim3(64), 0, stream, hc);
                hipLaunchKernelGGL(krun, dim3((unsigned)cdiv(nC, 64)), dim3(64), 0, stream, hc, nC);
            }
            OFP_LAUNCH_CHECK("k_hp_match / k_hp_resolve / k_hp_run");
            pend.hp_rounds = NR;
            OFP_HIP(hipMemcpyAsync(d->h_flags + 64, c, 2 * NR * sizeof(int), hipMemcpyDeviceToHost, stream));
        } else {
            hipLaunchKernelGGL(k_hp_plurality, dim3((unsigned)cdiv(nC0 * 16, 256)), dim3(256), 0, stream, hc, nC0);
            OFP_LAUNCH_CHECK("k_hp_plurality");
            // Verification rounds are enqueued a group at a time (three, then two) with ONE host synchronisation
            // per group; a round whose predecessor left nothing unresolved returns at once (HpCand::prev).
            const int* last = nullptr;
            for (int it = 0;;) {
                const int G = d->t.verify_group > 0 ? (int)std::min<int64_t>(d->t.verify_group, 8) : (it == 0 ? 3 : 2);
                int* c = nullptr;
                if (int rc = ctr.take(2 * G, &c)) return rc;
                // (a call that needs hundreds of rounds recycles the last counter slots, which are zeroed again: the
                //  previous round's counters may be among them and would then read "nothing left" -- no skip check then)
                if (c == ctr.base + OFP_N_COUNTERS - 16) last = nullptr;
                for (int q = 0; q < G; ++q) {
                    hc.counters = c + 2 * q;
                    hc.prev = last;
                    hipLaunchKernelGGL(k_hp_match, dim3((unsigned)cdiv(nM, 256)), dim3(256), 0, stream, hc, nM);
                    hipLaunchKernelGGL(k_hp_resolve, dim3((unsigned)chains), dim3(64), 0, stream, hc);
                    hipLaunchKernelGGL(krun, dim3((unsigned)cdiv(nC, 64)), dim3(64), 0, stream, hc, nC);
                    last = hc.counters;
                }
                OFP_LAUNCH_CHECK("k_hp_match / k_hp_resolve / k_hp_run");
                int* flags = d->h_flags;  // per round: chains stuck at a break, chains with unverified guesses
                OFP_HIP(hipMemcpyAsync(flags, c, 2 * G * sizeof(int), hipMemcpyDeviceToHost, stream));
                OFP_HIP(hipStreamSynchronize(stream));
                int stuck = 0;
                for (int q = 0; q < G; ++q) {
                    stuck = flags[2 * q] + flags[2 * q + 1];
                    pend.info[0] += 1;
                    pend.info[3] += stuck;
                    if (stuck == 0) break;
                }
                it += G;
                if (stuck == 0) break;
                if (d->t.max_passes > 0 && it > d->t.max_passes)
                    return ofp::fail(OFP_ERR_NOCONVERGE, "hp stage: %d chains still unresolved after %d rounds", stuck, it);
            }
        }
    }
    if (phase == 1 || phase == 6) return OFP_OK;  // (no high-pass: the head is the transpose alone)
    pend.ahead = !hv_fm;
    if (hv_fm || p.manual) pend.mm_nv = 0;
    if (hv_fm) pend.ar_nv = 0;
    if (!p.hp_enabled || hv_hp) pend.hp_rounds = 0;
    if (timed) OFP_HIP(hipEventRecord(ev[1], stream));
    ArArgs a;
    a.g = g;
    a.xdb = xdb;
    a.dif = dif;
    a.fa = p.fast_attack;
    a.fr = p.fast_release;
    a.sa = p.slow_attack;
    a.sr = p.slow_release;
    a.floor_db = p.floor_db;
    a.L = l.ar_L;
    a.W = l.ar_W;
    a.Wc = l.ar_Wc;
    a.Wf = l.ar_Wf;
    a.n_chunks = l.ar_chunks;
    a.S = l.ar_S;
    {   // walk-through chunks as their own pass 0 (k_ar_warm_both): the merged layout with 16-byte-congruent buffers
        a.through = (d->t.walk_through >= 0 && l.merge && l.ar_sym && (g.U & 3) == 0 && (l.ar_L & 3) == 0) ? 1 : 0;
        // complete-line stores for the output walks (walk_lines): the throughput layout, everything a multiple of 32 steps
        a.lines = (d->t.line_stores >= 0 && (a.through || d->t.line_stores > 0 || chains * l.ar_chunks >= (int64_t)32 * 4 * d->n_cus) && (g.U & 31) == 0 && (l.ar_L & 31) == 0) ? 1 : 0;
    }
    // dB and the per-chunk sums of the closed-form guess in one pass whenever both are wanted and the geometry
    // allows 16-byte groups (otherwise k_rect_db, then k_ar_sym_local reading the dB stream once more)
    const bool db_sym = l.ar_sym && p.hp_enabled && (g.U & 3) == 0 && (l.ar_L & 3) == 0 && d->t.fuse_db_sums >= 0;
    if (from_stage > 0) {
        // (the dB stream and the sums of the follower guess are in place)
    } else if (db_sym) {
        hipLaunchKernelGGL(k_rect_db_sym, dim3((unsigned)(chains * l.ar_chunks)), dim3(64), 0, stream, a, xdb,
                           chains * l.ar_chunks, reinterpret_cast<double*>(ws + l.o_ar_P));
        OFP_LAUNCH_CHECK("k_rect_db_sym");
    } else {
        hipLaunchKernelGGL(k_rect_db, dim3(ew_grid), dim3(256), 0, stream, g, xt, xdb, chains, p.hp_enabled ? 0 : 1,
                           p.floor_db);
        OFP_LAUNCH_CHECK("k_rect_db");
    }
    if (timed) OFP_HIP(hipEventRecord(ev[2], stream));

    // --- followers
    if (from_stage <= 1) {
        const int64_t nt = chains * l.ar_chunks;
        uint32_t* used = reinterpret_cast<uint32_t*>(ws + l.o_ar_state);
        const unsigned grid = (unsigned)cdiv(nt, 64);
        if (l.ar_sym) {
            double* P = reinterpret_cast<double*>(ws + l.o_ar_P);
            if (!db_sym) {
                hipLaunchKernelGGL(k_ar_sym_local, dim3((unsigned)nt), dim3(64), 0, stream, a, nt, P);
                OFP_LAUNCH_CHECK("k_ar_sym_local");
            }
            hipLaunchKernelGGL(k_ar_sym_combine, dim3((unsigned)chains), dim3(64), 0, stream, a, (const double*)P,
                               used);
            OFP_LAUNCH_CHECK("k_ar_sym_combine");
        } else {
            hipLaunchKernelGGL(k_ar_coarse, dim3(grid), dim3(64), 0, stream, a, nt, used);
            OFP_LAUNCH_CHECK("k_ar_coarse");
        }
        if (l.ar_sym) {
            const int64_t ntg = chains * cdiv(l.ar_chunks, l.ar_S);  // one run per group of S chunks
            if (l.merge)
                hipLaunchKernelGGL(a.lines ? k_ar_warm_both<true> : k_ar_warm_both<false>, dim3((unsigned)cdiv(ntg, 64)), dim3(64), 0,
                                   stream, a, ntg, used, used + 2 * nt);
            else
                hipLaunchKernelGGL(k_ar_warm2, dim3(2 * (unsigned)cdiv(ntg, 64)), dim3(64), 0, stream, a, ntg, used);
            OFP_LAUNCH_CHECK("k_ar_warm2");
        } else {
            hipLaunchKernelGGL(k_ar_warm, dim3(grid), dim3(64), 0, stream, a, nt, used);
            OFP_LAUNCH_CHECK("k_ar_warm");
        }
        if (!hv_fm) {
            pend.ar_nv = l.ar_chunks > 1 ? std::max(2, std::min(AHEAD_MAX_PASSES / 2, d->ar_pass_hint)) : 0;
            if (int rc = run_jacobi_ahead("follower stage", a.lines ? k_ar_chunk<true> : k_ar_chunk<false>, a, nt, l.ar_chunks, used, pass_flags, pend.ar_nv, stream))
                return rc;
        } else {
            int rc = run_jacobi("follower stage", a.lines ? k_ar_chunk<true> : k_ar_chunk<false>, a, nt, l.ar_chunks, used, ctr, d->h_flags, d->t.max_passes,
                                d->t.verify_group > 0 ? (int)d->t.verify_group : 2, stream, &pend.info[1], &pend.info[3]);
            if (rc != OFP_OK) return rc;
        }
    }
    if (timed) OFP_HIP(hipEventRecord(ev[3], stream));
    // block extremes for the crossing pass: whenever the 16-byte path of k_rel_out is taken for every tile (the same
    // conditions as in the kernel) and a group of four never straddles two blocks
    const bool use_sum = d->t.scan_skip >= 0 && (g.U & 3) == 0 && (l.tu & 3) == 0 && ((g.n_wb * g.C) & 3) == 0 &&
                         (((int64_t)l.tu * g.C) & 3) == 0 && ((g.Nm * g.C) & 3) == 0 && (g.B & 3) == 0 && g.B >= 32;
    uint32_t* sum_max = use_sum ? reinterpret_cast<uint32_t*>(ws + l.o_sum) : nullptr;
    uint32_t* sum_minv = use_sum ? sum_max + n_clips * l.nb * g.C : nullptr;
    if (from_stage <= 1) {
        const size_t lds = tile_lds + (use_sum ? (size_t)2 * g.C * (l.tu / g.B + 2) * 4 : 0);
        const int64_t n_tiles = cdiv(g.U, l.tu);
        const unsigned gx = (unsigned)std::min<int64_t>(n_tiles, std::max<int64_t>(1, (int64_t)32 * d->n_cus / n_clips));
        hipLaunchKernelGGL(k_rel_out, dim3(gx, (unsigned)n_clips), dim3(256), lds, stream, g, dif, d_rel,
                           p.floor_db, l.tu, sum_max, sum_minv, l.nb, mm_il ? rel_warm : nullptr, mm_il ? 0 : 1, n_tiles);
        OFP_LAUNCH_CHECK("k_rel_out");
    }
    if (timed) OFP_HIP(hipEventRecord(ev[4], stream));

    // --- tracker (relative thresholds only; in manual mode its state is never read)
    float* thr_mn = reinterpret_cast<float*>(ws + l.o_thr_mn);
    float* thr_mx = reinterpret_cast<float*>(ws + l.o_thr_mx);
    if (!p.manual) {
        MmArgs a;
        a.g = g;
        a.rel = rel;
        a.rel_il = d_rel;
        a.rel_warm = rel_warm;
        a.thr_mn = thr_mn;
        a.thr_mx = thr_mx;
        a.alpha_min = p.alpha_min;
        a.alpha_max = p.alpha_max;
        a.ialpha_min = d->ialpha_min;
        a.ialpha_max = d->ialpha_max;
        a.minmin = p.minmin;
        a.min0 = p.min0;
        a.max0 = p.max0;
        a.nb = l.nb;
        a.L = l.mm_L;
        a.W = l.mm_W;
        a.n_chunks = l.mm_chunks;
        a.n_chains = chains;
        a.S = l.mm_S;
        a.through = (mm_il && d->t.walk_through >= 0) ? 1 : 0;
        a.dirty = reinterpret_cast<uint8_t*>(ws + l.o_mm_dirty);
        const int64_t nt = chains * l.mm_chunks;
        uint32_t* used = reinterpret_cast<uint32_t*>(ws + l.o_mm_state);
        {
            const int64_t ntg = chains * cdiv(l.mm_chunks, l.mm_S);  // one run per group of S chunks
            if (mm_il && g.C == 8)
                hipLaunchKernelGGL(k_mm_warm_il<8>, dim3((unsigned)cdiv(ntg, 64)), dim3(64), 0, stream, a, ntg, used, used + 2 * nt);
            else if (mm_il && g.C == 64)
                hipLaunchKernelGGL(k_mm_warm_il<64>, dim3((unsigned)cdiv(ntg, 64)), dim3(64), 0, stream, a, ntg, used, used + 2 * nt);
            else if (mm_il)
                hipLaunchKernelGGL(k_mm_warm_il<4>, dim3((unsigned)cdiv(ntg, 64)), dim3(64), 0, stream, a, ntg, used, used + 2 * nt);
            else if (l.merge)
                hipLaunchKernelGGL(k_mm_warm_both, dim3((unsigned)cdiv(ntg, 64)), dim3(64), 0, stream, a, ntg, used);
            else
                hipLaunchKernelGGL(k_mm_warm2, dim3(2 * (unsigned)cdiv(ntg, 64)), dim3(64), 0, stream, a, ntg, used);
            OFP_LAUNCH_CHECK("k_mm_warm2");
        }
        using MmLight = void (*)(const MmArgs&, int64_t, const uint32_t*, uint32_t*, uint32_t*, int*, const int*, hipStream_t);
        const MmLight light_pl = +[](const MmArgs& m, int64_t, const uint32_t* ep, uint32_t* en, uint32_t* u, int* ch, const int* gate,
                                     hipStream_t st) {
            const int64_t n = m.n_chains * m.n_chunks;
            hipLaunchKernelGGL(k_mm_maxpass, dim3((unsigned)cdiv(n, 64)), dim3(64), 0, st, m, n, ep, en, u, ch, gate);
        };
        const MmLight light_il = +[](const MmArgs& m, int64_t, const uint32_t* ep, uint32_t* en, uint32_t* u, int* ch, const int* gate,
                                     hipStream_t st) {
            const int64_t n = m.n_chains * m.n_chunks;
            if (m.g.C == 8)
                hipLaunchKernelGGL(k_mm_maxpass_il<8>, dim3((unsigned)cdiv(n, 64)), dim3(64), 0, st, m, n, ep, en, u, ch, gate);
            else if (m.g.C == 64)
                hipLaunchKernelGGL(k_mm_maxpass_il<64>, dim3((unsigned)cdiv(n, 64)), dim3(64), 0, st, m, n, ep, en, u, ch, gate);
            else
                hipLaunchKernelGGL(k_mm_maxpass_il<4>, dim3((unsigned)cdiv(n, 64)), dim3(64), 0, st, m, n, ep, en, u, ch, gate);
        };
        const MmLight light = mm_il ? light_il : light_pl;
        const auto mm_chunk_k = mm_il ? (g.C == 8 ? k_mm_chunk_il<8> : (g.C == 64 ? k_mm_chunk_il<64> : k_mm_chunk_il<4>))
                                      : (l.merge ? k_mm_chunk_both : k_mm_chunk);
        if (!hv_fm) {
            pend.mm_nv = l.mm_chunks > 1 ? std::max(2, std::min(AHEAD_MAX_PASSES / 2, d->mm_pass_hint)) : 0;
            if (int rc = run_jacobi_ahead("tracker stage", mm_chunk_k, a,
                                          l.merge ? nt : 2 * 64 * cdiv(nt, 64), l.mm_chunks, used, pass_flags + AHEAD_MAX_PASSES,
                                          pend.mm_nv, stream, light, 2 * nt))
                return rc;
        } else {
            int rc = run_jacobi("tracker stage", mm_chunk_k, a, l.merge ? nt : 2 * 64 * cdiv(nt, 64),
                                l.mm_chunks, used, ctr, d->h_flags,
                                d->t.max_passes, d->t.verify_group > 0 ? (int)d->t.verify_group : 2, stream, &pend.info[2], &pend.info[3],
                                light, 2 * nt);
            if (rc != OFP_OK) return rc;
        }
    }
    if (timed) OFP_HIP(hipEventRecord(ev[5], stream));

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
    sa.vflag = reinterpret_cast<uint32_t*>(ws + l.o_vflag);
    sa.sum_max = sum_max;
    sa.sum_minv = sum_minv;
    {
        const int64_t total = n_clips * l.nb * g.C;
        const unsigned bs_grid = (unsigned)std::min<int64_t>(cdiv(total, 4), 256 * 32);  // 4 waves per workgroup
        if (mm_il) {
            const unsigned il_grid = (unsigned)std::min<int64_t>(cdiv(n_clips * l.nb, 4), 256 * 32);
            if (g.C == 8)
                hipLaunchKernelGGL(k_block_scan_il<8>, dim3(il_grid), dim3(256), 0, stream, sa, d_rel);
            else if (g.C == 64)
                hipLaunchKernelGGL(k_block_scan_il<64>, dim3(il_grid), dim3(256), 0, stream, sa, d_rel);
            else
                hipLaunchKernelGGL(k_block_scan_il<4>, dim3(il_grid), dim3(256), 0, stream, sa, d_rel);
        } else {
            hipLaunchKernelGGL(k_block_scan, dim3(bs_grid), dim3(256), 0, stream, sa);
        }
        OFP_LAUNCH_CHECK("k_block_scan");
    }
    VisArgs va;
    va.vflag = sa.vflag;
    va.fc = sa.first_cross;
    va.lb = sa.last_below;
    int32_t* pc = reinterpret_cast<int32_t*>(ws + l.o_pc);
    va.pc = pc;
    va.nb = l.nb;
    va.C = g.C;
    va.vis_j = const_cast<int32_t*>(sm.vis_j);
    va.vfc = const_cast<int32_t*>(sm.vfc);
    va.vlb = const_cast<int32_t*>(sm.vlb);
    va.vpc = const_cast<int32_t*>(sm.vpc);
    va.nv = va_nv;
    {
        const int64_t n_lt = cdiv(l.nb, 256);
        int32_t* lt = reinterpret_cast<int32_t*>(ws + l.o_ltile);
        OFP_REQUIRE(n_lt * chains < (1ll << 31), "ofp_detect_offline: %lld block tiles in one call", (long long)(n_lt * chains));
        const dim3 lgrid((unsigned)(n_lt * chains));
        hipLaunchKernelGGL(k_last_clear<false>, lgrid, dim3(64), 0, stream, (const int32_t*)sa.last_below, pc, lt, l.nb, g.C, n_lt);
        hipLaunchKernelGGL(k_last_clear_scan, dim3((unsigned)chains), dim3(64), 0, stream, lt, n_lt);
        hipLaunchKernelGGL(k_last_clear<true>, lgrid, dim3(64), 0, stream, (const int32_t*)sa.last_below, pc, lt, l.nb, g.C, n_lt);
        OFP_LAUNCH_CHECK("k_last_clear");
    }
    {
        const int64_t n_tiles = cdiv(l.nb, 256);
        int32_t* vtile = reinterpret_cast<int32_t*>(ws + l.o_vtile);
        hipLaunchKernelGGL(k_visit_count, dim3((unsigned)n_tiles, (unsigned)n_clips), dim3(256), 0, stream, va, vtile, n_tiles);
        hipLaunchKernelGGL(k_visit_scan, dim3((unsigned)n_clips), dim3(64), 0, stream, vtile, n_tiles, va.nv);
        hipLaunchKernelGGL(k_visit_scatter, dim3((unsigned)n_tiles, (unsigned)n_clips), dim3(256), 0, stream, va,
                           (const int32_t*)vtile, n_tiles);
        OFP_LAUNCH_CHECK("k_visit_count / k_visit_scan / k_visit_scatter");
    }
    pend.sm_flag = false;
    if (l.sm_seg && l.nb > 0) {
        SmSegArgs ss;
        ss.n_seg = cdiv(l.nb, SM_SEG);
        const int64_t words = n_clips * ss.n_seg * 64 * 2;
        ss.used = reinterpret_cast<int32_t*>(ws + l.o_smseg);
        ss.endA = ss.used + words;
        ss.endB = ss.endA + words;
        ss.cnt = ss.endB + words;
        const int tb = std::max(1, std::min(64, 1024 / g.C));
        const size_t lds = (size_t)3 * tb * g.C * 4 + 64 * 4;
        const dim3 grid((unsigned)ss.n_seg, (unsigned)n_clips);
        constexpr int V = 4;  // verification passes enqueued ahead (a pass that finds nothing costs microseconds)
        int* c = nullptr;
        if (int rc = ctr.take(V, &c)) return rc;
        hipLaunchKernelGGL(k_sm_seg, grid, dim3(64), lds, stream, sm, ss, 0, c);
        for (int q = 1; q <= V; ++q) hipLaunchKernelGGL(k_sm_seg, grid, dim3(64), lds, stream, sm, ss, q, c + q - 1);
        hipLaunchKernelGGL(k_sm_offsets, dim3((unsigned)n_clips), dim3(64), 0, stream, sm, ss);
        hipLaunchKernelGGL(k_sm_seg, grid, dim3(64), lds, stream, sm, ss, -1, c);
        OFP_LAUNCH_CHECK("k_sm_seg / k_sm_offsets");
        pend.sm_flag = true;
        OFP_HIP(hipMemcpyAsync(d->h_flags + 40, c + V - 1, sizeof(int), hipMemcpyDeviceToHost, stream));
    } else {
        if (int rc = sequential_machine()) return rc;
    }
    if (int rc = backtrack()) return rc;
    if (!hv_fm)
        OFP_HIP(hipMemcpyAsync(d->h_flags + 96, pass_flags, 2 * AHEAD_MAX_PASSES * sizeof(int), hipMemcpyDeviceToHost, stream));
    if (timed) OFP_HIP(hipEventRecord(ev[6], stream));
    pend.valid = true;
    if (enqueue_only) return OFP_OK;
    OFP_HIP(hipStreamSynchronize(stream));
    return complete();
}

int ofp_detect_offline(ofp_detector* d, const float* d_x, int64_t n_clips, int64_t N, int64_t warm, float* d_rel,
                       ofp_onset* d_records, int64_t cap, int64_t* d_counts, void* d_ws, int64_t ws_bytes,
                       int64_t* h_info, void* stream) {
    return detect_impl(d, d_x, n_clips, N, warm, d_rel, d_records, cap, d_counts, d_ws, ws_bytes, h_info, stream, 0);
}

int ofp_detect_offline_enqueue(ofp_detector* d, const float* d_x, int64_t n_clips, int64_t N, int64_t warm, float* d_rel,
                               ofp_onset* d_records, int64_t cap, int64_t* d_counts, void* d_ws, int64_t ws_bytes,
                               void* stream) {
    return detect_impl(d, d_x, n_clips, N, warm, d_rel, d_records, cap, d_counts, d_ws, ws_bytes, nullptr, stream, 0, 1);
}

int ofp_detect_offline_complete(ofp_detector* d, const float* d_x, int64_t n_clips, int64_t N, int64_t warm, float* d_rel,
                                ofp_onset* d_records, int64_t cap, int64_t* d_counts, void* d_ws, int64_t ws_bytes,
                                int64_t* h_info, void* stream) {
    return detect_impl(d, d_x, n_clips, N, warm, d_rel, d_records, cap, d_counts, d_ws, ws_bytes, h_info, stream, 7);
}

int ofp_detect_offline_begin(ofp_detector* d, const float* d_x, int64_t n_clips, int64_t N, int64_t warm,
                             void* d_ws, int64_t ws_bytes, void* stream) {
    int64_t dummy = 0;  // d_counts is not touched by the head; a non-NULL value passes the argument check
    return detect_impl(d, d_x, n_clips, N, warm, nullptr, nullptr, 0, &dummy, d_ws, ws_bytes, nullptr, stream, 1);
}

int ofp_detect_offline_finish(ofp_detector* d, const float* d_x, int64_t n_clips, int64_t N, int64_t warm,
                              float* d_rel, ofp_onset* d_records, int64_t cap, int64_t* d_counts, void* d_ws,
                              int64_t ws_bytes, int64_t* h_info, void* stream) {
    return detect_impl(d, d_x, n_clips, N, warm, d_rel, d_records, cap, d_counts, d_ws, ws_bytes, h_info, stream, 2);
}

int ofp_detect_offline_finish_enqueue(ofp_detector* d, const float* d_x, int64_t n_clips, int64_t N, int64_t warm,
                                      float* d_rel, ofp_onset* d_records, int64_t cap, int64_t* d_counts, void* d_ws,
                                      int64_t ws_bytes, void* stream) {
    return detect_impl(d, d_x, n_clips, N, warm, d_rel, d_records, cap, d_counts, d_ws, ws_bytes, nullptr, stream, 2, 1);
}

int ofp_detect_offline_begin_input(ofp_detector* d, const float* d_x, int64_t n_clips, int64_t N, int64_t warm,
                                   void* d_ws, int64_t ws_bytes, void* stream) {
    int64_t dummy = 0;
    return detect_impl(d, d_x, n_clips, N, warm, nullptr, nullptr, 0, &dummy, d_ws, ws_bytes, nullptr, stream, 5);
}

int ofp_detect_offline_begin_iir(ofp_detector* d, const float* d_x, int64_t n_clips, int64_t N, int64_t warm,
                                 void* d_ws, int64_t ws_bytes, void* stream) {
    int64_t dummy = 0;
    return detect_impl(d, d_x, n_clips, N, warm, nullptr, nullptr, 0, &dummy, d_ws, ws_bytes, nullptr, stream, 6);
}

}  // extern "C"
