// Device-side pieces of the streaming detector shared by csrc/ofp_stream.hip (ofp_stream_process) and
// csrc/ofp_hop.hip (the per-hop session runs the detector as one workgroup of its fused kernel).
#pragma once
#include "../../include/ofp_math.h"
#include "ofp_detector.h"

namespace ofpstream {

// SoA state block in device memory
struct StreamState {
    float* zi;      // [4][C]
    float* yf;      // [C]
    float* ys;      // [C]
    float* mn;      // [C]
    float* mx;      // [C]
    double* prev;   // [C]
    int64_t* deb;   // [C]
    int32_t* state; // [C]
    float* hist;    // [bt_N][C] backtracking history (oldest row first)
    float* relbuf;  // [B][C] relative envelope of the current block
};

__host__ __device__ inline int64_t state_bytes(int C, int B, int64_t btN) {
    int64_t f = (int64_t)C * (4 + 4) * 4;     // zi, yf, ys, mn, mx
    int64_t d = (int64_t)C * (8 + 8);         // prev, deb
    int64_t i = (int64_t)C * 4;               // state
    int64_t h = btN * C * 4 + (int64_t)B * C * 4;
    return ((f + d + i + h + 255) / 256) * 256;
}

__host__ __device__ inline StreamState carve(void* base, int C, int B, int64_t btN) {
    StreamState s;
    unsigned char* p = static_cast<unsigned char*>(base);
    s.prev = reinterpret_cast<double*>(p);  p += (int64_t)C * 8;
    s.deb = reinterpret_cast<int64_t*>(p);  p += (int64_t)C * 8;
    s.zi = reinterpret_cast<float*>(p);     p += (int64_t)C * 16;
    s.yf = reinterpret_cast<float*>(p);     p += (int64_t)C * 4;
    s.ys = reinterpret_cast<float*>(p);     p += (int64_t)C * 4;
    s.mn = reinterpret_cast<float*>(p);     p += (int64_t)C * 4;
    s.mx = reinterpret_cast<float*>(p);     p += (int64_t)C * 4;
    s.state = reinterpret_cast<int32_t*>(p); p += (int64_t)C * 4;
    s.hist = reinterpret_cast<float*>(p);   p += btN * C * 4;
    s.relbuf = reinterpret_cast<float*>(p);
    return s;
}

struct StreamArgs {
    int C, B;
    float floor_db;
    int hp_on;
    float b[5], a[5];
    float fa, fr, sa, sr;
    float alpha_min, alpha_max, ialpha_min, ialpha_max, minmin, min0, max0;
    int manual;
    int64_t cooldown;
    int backtrack;
    int64_t btN;
    float bt_alpha, bt_tol;
    const float* on_f;
    const float* off_f;
    const double* on_d;
    void* state;
    const float* x;
    int64_t n_blocks, n_rows, sample_base;
    int warmup;
    float* rel;
    ofp_onset* records;
    int64_t cap;
    int64_t* count;
    int fresh_count;   // != 0: the count starts at 0 instead of *count (per-hop session: no read of the result block)
};

// ---- phase-split form of the same block step -------------------------------------------------
// k_stream walks a block with ONE lane per channel: ~150 dependent instructions per sample (two fp64
// table evaluations for dB and back, the filter, two followers, two trackers), ~0.4 us per sample on a
// lone wave -- 100 us for a 256-sample hop.  Only three short recurrences are actually sequential;
// everything else is elementwise.  k_stream_par therefore runs a block as phases separated by
// workgroup barriers, with the block staged in LDS:
//   1 IIR               one lane per channel (10 instructions per sample)       [sequential]
//   2 rectified dB      all lanes over the B x C samples                        [elementwise]
//   3 followers         fast and slow on separate lanes (8 each)                [sequential]
//   4 back to linear    all lanes                                               [elementwise]
//   5 tracker           min and max on separate lanes (4 each)                  [sequential]
//   6 crossings         all lanes, first / last per channel through LDS atomics [elementwise]
//   7 hysteresis / cooldown / records as in k_stream                            [per channel]
// Every value goes through the same ofp_* operations in the same order, so the results are bit-identical
// to k_stream (and to the oracle).  Needs 3 x B x C floats of LDS and C <= 512; other shapes and the
// one-off warm-up take k_stream.
constexpr int PAR_MAX_C = 512;

// ofp_df2t4_step with the same operations in the same order, arranged as 2-wide vectors so that the
// compiler can use packed fp32 (v_pk_mul_f32 / v_pk_add_f32 are per-lane IEEE fp32): pairs (z0,z2) and
// (z1,z3); the last tap adds -0.0f, exact for every addend.  10 instructions per sample instead of 17
// (the same form the offline kernels use, csrc/ofp_detect.hip HpStep).
typedef float ofp_v2f __attribute__((ext_vector_type(2)));
struct HpPacked {
    float z[4];
    float b0;
    ofp_v2f B13, B24, A13, A24;
    __device__ void init(const float* b, const float* a, const float* z0) {
        b0 = b[0];
        B13 = ofp_v2f{b[1], b[3]}; B24 = ofp_v2f{b[2], b[4]};
        A13 = ofp_v2f{a[1], a[3]}; A24 = ofp_v2f{a[2], a[4]};
        for (int k = 0; k < 4; ++k) z[k] = z0[k];
    }
    __device__ float operator()(float xv) {
        const ofp_v2f O = {z[1], z[3]};
        const ofp_v2f Wz = {z[2], -0.0f};
        const float y = z[0] + b0 * xv;
        const ofp_v2f xx = {xv, xv}, yy = {y, y};
        const ofp_v2f E2 = (O + B13 * xx) - A13 * yy;   // (z0', z2')
        const ofp_v2f O2 = (Wz + B24 * xx) - A24 * yy;  // (z1', z3')
        z[0] = E2.x; z[2] = E2.y; z[1] = O2.x; z[3] = O2.y;
        return y;
    }
};

template <class F>
__device__ __forceinline__ void seq_walk(const float* __restrict__ in, int B, int C, F&& f) {
    // a lane's column of the staged block, eight loads ahead of the recurrence that consumes them
    int t0 = 0;
    for (; t0 + 8 <= B; t0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = in[(t0 + u) * C];
#pragma unroll
        for (int u = 0; u < 8; ++u) f(t0 + u, v[u]);
    }
    for (; t0 < B; ++t0) f(t0, in[t0 * C]);
}

__device__ __forceinline__ void stream_par_blocks(const StreamArgs& a, float* sbuf /* three [B][C] planes of LDS */) {
    __shared__ int s_first[PAR_MAX_C], s_last[PAR_MAX_C];
    __shared__ float s_on[PAR_MAX_C], s_off[PAR_MAX_C];
    __shared__ double s_on0[PAR_MAX_C], s_prev[PAR_MAX_C];
    __shared__ int s_red[16];
    __shared__ int s_max;
    __shared__ long long s_base;
    const int C = a.C, B = a.B, BC = a.B * a.C;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = (nt + 63) >> 6;
    float* p0 = sbuf;
    float* p1 = sbuf + BC;
    float* p2 = sbuf + 2 * BC;
    StreamState s = carve(a.state, C, B, a.btN);
    // per-channel lane (tid < C): filter state, hysteresis; role lanes (tid < 2C): one follower / one tracker word
    const bool act = tid < C;
    const int c = tid;
    const bool role = tid < 2 * C;
    const int rc = tid >> 1, which = tid & 1;
    float z[4] = {0, 0, 0, 0};
    double prev = 0.0;
    int64_t deb = 0;
    int state = 0;
    float fol = 0.0f, trk = 0.0f;
    if (act) {
        for (int k = 0; k < 4; ++k) z[k] = s.zi[k * C + c];
        prev = s.prev[c];
        deb = s.deb[c];
        state = s.state[c];
    }
    if (role) {
        fol = which ? s.ys[rc] : s.yf[rc];
        trk = which ? s.mx[rc] : s.mn[rc];
    }
    if (tid == 0) s_base = a.fresh_count ? 0 : *a.count;
    __syncthreads();
    long long base = s_base;
    for (int64_t blk = 0; blk < a.n_blocks; ++blk) {
        float* relcol = (a.rel ? a.rel + blk * BC : s.relbuf);
        // 0: the block into LDS with coalesced loads (the input may be host memory: one PCIe round trip
        // for the whole hop instead of one per batch of a lane's sequential walk)
        {
            const float* xg = a.x + blk * BC;
            for (int i = tid; i < BC; i += nt) p1[i] = xg[i];
            __syncthreads();
        }
        const float* xb = p1;
        // 1: high-pass (:743-744)
        if (a.hp_on) {
            if (act) {
                HpPacked hp;
                hp.init(a.b, a.a, z);
                seq_walk(xb + c, B, C, [&](int t, float v) { p0[t * C + c] = hp(v); });
                for (int k = 0; k < 4; ++k) z[k] = hp.z[k];
            }
            __syncthreads();
        }
        // 2: rectified dB with floor (:747-748)
        for (int i = tid; i < BC; i += nt) p0[i] = ofp_rect_db(a.hp_on ? p0[i] : xb[i], a.floor_db);
        if (act) {
            s_first[c] = 0x7fffffff;
            s_last[c] = -1;
            s_prev[c] = prev;
        }
        __syncthreads();
        // 3: the two followers (:751), one lane each
        if (role) {
            float* out = which ? p2 : p1;
            const float att = which ? a.sa : a.fa, rel = which ? a.sr : a.fr;
            seq_walk(p0 + rc, B, C, [&](int t, float v) {
                fol = ofp_ar_step(v, fol, att, rel);
                out[t * C + rc] = fol;
            });
        }
        __syncthreads();
        // 4: back to linear and clip (:753-754)
        for (int i = tid; i < BC; i += nt) {
            const float r = ofp_rel_linear(p1[i] - p2[i], a.floor_db);
            p0[i] = r;
            relcol[i] = r;
        }
        __syncthreads();
        // 5: min / max tracker over the whole block (:762), one lane each
        if (!a.manual) {
            if (role) {
                if (which) seq_walk(p0 + rc, B, C, [&](int, float v) { trk = ofp_max_step(v, trk, a.ialpha_max, a.alpha_max); });
                else seq_walk(p0 + rc, B, C, [&](int, float v) { trk = ofp_min_step(v, trk, a.ialpha_min, a.alpha_min, a.minmin); });
                (which ? p2 : p1)[rc] = trk;  // planes 1 / 2 are free again: post-block min / max per channel
            }
            __syncthreads();
        }
        // thresholds from the post-block tracker (:763, :787)
        if (act) {
            float on, off;
            double on0;
            if (a.manual) {
                on = a.on_f[c]; on0 = a.on_d[c]; off = a.off_f[c];
            } else {
                const float mn = p1[c], mx = p2[c];
                const float t1 = mx * a.on_f[c]; on = t1 + mn; on0 = (double)on;
                const float t2 = mx * a.off_f[c]; off = t2 + mn;
            }
            s_on[c] = on; s_off[c] = off; s_on0[c] = on0;
            prev = (double)p0[(B - 1) * C + c];                                // :792
            if (a.backtrack) {  // :755-756 ring-buffer write == shift by B rows, append
                for (int64_t r = 0; r + B < a.btN; ++r) s.hist[r * C + c] = s.hist[(r + B) * C + c];
                for (int t = 0; t < B; ++t) s.hist[(a.btN - B + t) * C + c] = p0[t * C + c];
            }
        }
        __syncthreads();
        // 6: first upward crossing and last row below `off`, per channel (:764-770, :784-790)
        for (int i = tid; i < BC; i += nt) {
            const int t = i / C, cc = i - t * C;
            const float v = p0[i];
            const float on = s_on[cc];
            const bool below_before = t == 0 ? (s_prev[cc] < s_on0[cc]) : (p0[i - C] < on);   // :769-770
            if (v > on && below_before) atomicMin(&s_first[cc], t);
            if (v < s_off[cc]) atomicMax(&s_last[cc], t);
        }
        __syncthreads();
        // 7: gate, cross-channel max, records, state update (as k_stream)
        int first = -1, last = -1;
        if (act) {
            first = s_first[c] == 0x7fffffff ? -1 : s_first[c];
            last = s_last[c];
        }
        const bool gate = act && !state && deb < 1;                              // :764-768
        const bool onf = gate && first >= 0;
        const int oi = onf ? first : 0;                                          // :774
        int m = oi;
        for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
        const unsigned long long bal = __ballot(onf);
        if (lane == 0) s_red[wave] = m;
        __syncthreads();
        if (tid == 0) {
            int mm = 0;
            for (int w = 0; w < nwaves; ++w) mm = max(mm, s_red[w]);
            s_max = mm;
        }
        __syncthreads();
        const int omax = s_max;
        if (lane == 0) s_red[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < nwaves; ++w) { const int n = s_red[w]; if (w < wave) woff += n; tot += n; }
        if (act) {
            if (onf) { state = 1; deb = a.cooldown; }                            // :778-779
            if (deb > 0) deb -= B;                                               // :780
            if (last >= omax) state = 0;                                         // :784-791
            if (onf) {
                int64_t delta = oi;
                if (a.backtrack) {                                               // :800-825
                    const float omba = (float)(1.0 - (double)a.bt_alpha);
                    int64_t i = B - delta;
                    float cur = s.hist[(a.btN - i) * C + c];
                    i += 1;
                    float pv = i <= a.btN ? s.hist[(a.btN - i) * C + c] : 0.0f;
                    float ps = a.bt_alpha * pv + omba * cur;
                    while (cur > ps && fabsf(ps - pv) > a.bt_tol && (i + 1 < a.btN)) {
                        delta -= 1; i += 1; cur = ps;
                        pv = s.hist[(a.btN - i) * C + c];
                        ps = a.bt_alpha * pv + omba * cur;
                    }
                }
                const long long pos = base + woff + __popcll(bal & ((1ull << lane) - 1ull));
                if (pos < a.cap) {
                    a.records[pos].clip = 0;
                    a.records[pos].channel = c;
                    a.records[pos].sample = a.sample_base + blk * B + delta;
                }
            }
        }
        base += tot;
        __syncthreads();
    }
    if (act) {
        for (int k = 0; k < 4; ++k) s.zi[k * C + c] = z[k];
        s.prev[c] = prev; s.deb[c] = deb; s.state[c] = state;
    }
    if (role) {
        (which ? s.ys : s.yf)[rc] = fol;
        (which ? s.mx : s.mn)[rc] = trk;
    }
    if (tid == 0) *a.count = base;
}


// the launch arguments of a detector handle (defined in ofp_stream.hip)
StreamArgs make_stream_args(const ofp_detector* d);

}  // namespace ofpstream
