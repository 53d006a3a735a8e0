// Streaming form of the amplitude onset detector: AmplitudeOnsetDetector.__call__
// (detection.py:727-798) and init_minmax_tracker (:827-840) with the detector
// state resident in HBM.  One launch per call, no host synchronisation, so a
// per-hop detect step can be captured into a hipGraph (BASELINE config 5).
//
// One workgroup per detector instance; one thread per channel walks its column of
// the block (the recurrences are sequential in time); the block-level logic
// needs two cross-channel exchanges (max first-crossing index, record order).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "../../include/ofp_math.h"
#include "ofp_detector.h"
#include "ofp_stream_dev.h"

using namespace ofpstream;

namespace {

__global__ void k_stream_init(StreamArgs a) {
    StreamState s = carve(a.state, a.C, a.B, a.btN);
    for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
        for (int k = 0; k < 4; ++k) s.zi[k * a.C + c] = 0.0f;       // detection.py:497
        s.yf[c] = a.floor_db;                                       // :697-702
        s.ys[c] = a.floor_db;
        s.mn[c] = a.min0;                                           // :704
        s.mx[c] = a.max0;
        s.prev[c] = 0.0;                                            // :711
        s.deb[c] = 0;                                               // :712
        s.state[c] = 0;                                             // :710
        for (int64_t r = 0; r < a.btN; ++r) s.hist[r * a.C + c] = 0.0f;
    }
}

__global__ __launch_bounds__(1024) void k_stream(StreamArgs a) {
    __shared__ int s_red[16];
    __shared__ int s_max;
    __shared__ long long s_base;
    const int C = a.C, B = a.B;
    const int c = threadIdx.x;
    const bool act = c < C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = (blockDim.x + 63) >> 6;
    StreamState s = carve(a.state, C, B, a.btN);
    float z[4] = {0, 0, 0, 0}, yf = 0, ys = 0, mn = 0, mx = 0;
    if (act) {
        for (int k = 0; k < 4; ++k) z[k] = s.zi[k * C + c];
        yf = s.yf[c]; ys = s.ys[c]; mn = s.mn[c]; mx = s.mx[c];
    }
    if (a.warmup) {
        // init_minmax_tracker (detection.py:827-840): high-pass over every row, the
        // followers and the tracker over full blocks only; thresholds/state untouched
        if (act) {
            const int64_t full = (a.n_rows / B) * B;
            for (int64_t t = 0; t < a.n_rows; ++t) {
                float v = a.x[t * C + c];
                if (a.hp_on) v = ofp_df2t4_step(v, a.b, a.a, z);
                if (t < full) {
                    v = ofp_rect_db(v, a.floor_db);
                    yf = ofp_ar_step(v, yf, a.fa, a.fr);
                    ys = ofp_ar_step(v, ys, a.sa, a.sr);
                    float r = ofp_rel_linear(yf - ys, a.floor_db);
                    mn = ofp_min_step(r, mn, a.ialpha_min, a.alpha_min, a.minmin);
                    mx = ofp_max_step(r, mx, a.ialpha_max, a.alpha_max);
                }
            }
            for (int k = 0; k < 4; ++k) s.zi[k * C + c] = z[k];
            s.yf[c] = yf; s.ys[c] = ys; s.mn[c] = mn; s.mx[c] = mx;
        }
        return;
    }
    double prev = 0.0;
    int64_t deb = 0;
    int state = 0;
    if (act) { prev = s.prev[c]; deb = s.deb[c]; state = s.state[c]; }
    if (threadIdx.x == 0) s_base = *a.count;
    __syncthreads();
    long long base = s_base;
    for (int64_t blk = 0; blk < a.n_blocks; ++blk) {
        float* relcol = (a.rel ? a.rel + blk * B * C : s.relbuf);
        // phase 1: the per-sample chain for this channel (:743-754, :762)
        if (act) {
            const float* xb = a.x + blk * B * C + c;
            for (int t = 0; t < B; ++t) {
                float v = xb[(int64_t)t * C];
                if (a.hp_on) v = ofp_df2t4_step(v, a.b, a.a, z);
                v = ofp_rect_db(v, a.floor_db);
                yf = ofp_ar_step(v, yf, a.fa, a.fr);
                ys = ofp_ar_step(v, ys, a.sa, a.sr);
                float r = ofp_rel_linear(yf - ys, a.floor_db);
                relcol[(int64_t)t * C + c] = r;
                if (!a.manual) {
                    mn = ofp_min_step(r, mn, a.ialpha_min, a.alpha_min, a.minmin);
                    mx = ofp_max_step(r, mx, a.ialpha_max, a.alpha_max);
                }
            }
            if (a.backtrack) {  // :755-756 ring-buffer write == shift by B rows, append
                for (int64_t r = 0; r + B < a.btN; ++r) s.hist[r * C + c] = s.hist[(r + B) * C + c];
                for (int t = 0; t < B; ++t) s.hist[(a.btN - B + t) * C + c] = relcol[(int64_t)t * C + c];
            }
        }
        // phase 2: thresholds from the post-block tracker, first crossing, last-below
        float on = 0, off = 0;
        double on0 = 0;
        int first = -1, last = -1;
        if (act) {
            if (a.manual) { on = a.on_f[c]; on0 = a.on_d[c]; off = a.off_f[c]; }
            else {
                float t1 = mx * a.on_f[c]; on = t1 + mn; on0 = (double)on;   // :763
                float t2 = mx * a.off_f[c]; off = t2 + mn;                   // :787
            }
            bool below_before = prev < on0;                                  // :769
            float v = 0.0f;
            for (int t = 0; t < B; ++t) {
                v = relcol[(int64_t)t * C + c];
                if (first < 0 && v > on && below_before) first = t;
                if (v < off) last = t;
                below_before = v < on;                                       // :770
            }
            prev = (double)v;                                                // :792
        }
        bool gate = act && !state && deb < 1;                                // :764-768
        bool onf = gate && first >= 0;
        int oi = onf ? first : 0;                                            // :774
        // max over all channels (:790)
        int m = oi;
        for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
        unsigned long long bal = __ballot(onf);
        if (lane == 0) { s_red[wave] = m; }
        __syncthreads();
        if (threadIdx.x == 0) {
            int mm = 0;
            for (int w = 0; w < nwaves; ++w) mm = max(mm, s_red[w]);
            s_max = mm;
        }
        __syncthreads();
        const int omax = s_max;
        // record order: channel ascending -> prefix over waves
        if (lane == 0) s_red[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < nwaves; ++w) { int n = s_red[w]; if (w < wave) woff += n; tot += n; }
        if (act) {
            if (onf) { state = 1; deb = a.cooldown; }                        // :778-779
            if (deb > 0) deb -= B;                                           // :780
            if (last >= omax) state = 0;                                     // :784-791
            if (onf) {
                int64_t delta = oi;
                if (a.backtrack) {                                           // :800-825
                    float omba = (float)(1.0 - (double)a.bt_alpha);
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
                long long pos = base + woff + __popcll(bal & ((1ull << lane) - 1ull));
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
        s.yf[c] = yf; s.ys[c] = ys; s.mn[c] = mn; s.mx[c] = mx;
        s.prev[c] = prev; s.deb[c] = deb; s.state[c] = state;
    }
    if (threadIdx.x == 0) *a.count = base;
}

__global__ __launch_bounds__(1024) void k_stream_par(StreamArgs a) {
    extern __shared__ __align__(16) float sbuf[];
    stream_par_blocks(a, sbuf);
}

// AmplitudeOnsetDetector.init (detection.py:842-888), the passes over the samples; one lane per
// channel, rows in order (a one-off calibration over some seconds of audio, not a hot path).
struct CalArgs {
    StreamArgs a;
    int64_t n, r0, r1, n_rev;
    float* xdb;   // [n][C] scratch: the unclipped rectified dB of the filtered rows
    float* rel;   // [n][C] out: fast - slow in dB (:862-867)
};

__global__ __launch_bounds__(64) void k_calibrate(CalArgs q) {
    const StreamArgs& a = q.a;
    const int C = a.C;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    StreamState s = carve(a.state, C, a.B, a.btN);
    float z[4];
    for (int k = 0; k < 4; ++k) z[k] = s.zi[k * C + c];
    float yf = s.yf[c], ys = s.ys[c];
    const float ninf = -__builtin_inff();
    for (int64_t t = 0; t < q.n; ++t) {                       // :849-852 (no floor clip here)
        float v = a.x[t * C + c];
        if (a.hp_on) v = ofp_df2t4_step(v, a.b, a.a, z);
        q.xdb[t * C + c] = ofp_rect_db(v, ninf);
    }
    for (int64_t t = q.r0; t < q.r1; ++t) {                   // :855-860
        const float v = q.xdb[t * C + c];
        yf = ofp_ar_step(v, yf, a.fa, a.fr);
        ys = ofp_ar_step(v, ys, a.sa, a.sr);
    }
    for (int64_t t = 0; t < q.n; ++t) {                       // :862-867
        const float v = q.xdb[t * C + c];
        yf = ofp_ar_step(v, yf, a.fa, a.fr);
        ys = ofp_ar_step(v, ys, a.sa, a.sr);
        q.rel[t * C + c] = yf - ys;
    }
    for (int64_t t = q.n_rev - 1; t >= 0; --t) {              // :883-888
        const float v = q.xdb[t * C + c];
        yf = ofp_ar_step(v, yf, a.fa, a.fr);
        ys = ofp_ar_step(v, ys, a.sa, a.sr);
    }
    for (int k = 0; k < 4; ++k) s.zi[k * C + c] = z[k];
    s.yf[c] = yf;
    s.ys[c] = ys;
}

StreamArgs make_args(const ofp_detector* d) { return ofpstream::make_stream_args(d); }

}  // namespace

StreamArgs ofpstream::make_stream_args(const ofp_detector* d) {
    StreamArgs a;
    std::memset(&a, 0, sizeof(a));
    const auto& p = d->p;
    a.C = p.n_channels; a.B = p.block_size; a.floor_db = p.floor_db; a.hp_on = p.hp_enabled;
    std::memcpy(a.b, d->b, sizeof(a.b));
    std::memcpy(a.a, d->a, sizeof(a.a));
    a.fa = p.fast_attack; a.fr = p.fast_release; a.sa = p.slow_attack; a.sr = p.slow_release;
    a.alpha_min = p.alpha_min; a.alpha_max = p.alpha_max;
    a.ialpha_min = d->ialpha_min; a.ialpha_max = d->ialpha_max;
    a.minmin = p.minmin; a.min0 = p.min0; a.max0 = p.max0;
    a.manual = p.manual; a.cooldown = p.cooldown;
    a.backtrack = p.backtrack; a.btN = p.backtrack ? p.backtrack_buffer_size : 0;
    a.bt_alpha = p.backtrack_alpha; a.bt_tol = p.backtrack_tol;
    a.on_f = d->d_on_f; a.off_f = d->d_off_f; a.on_d = d->d_on_d;
    return a;
}

extern "C" {

int64_t ofp_stream_state_bytes(const ofp_detector* d) {
    if (!d) return -1;
    return state_bytes(d->p.n_channels, d->p.block_size, d->p.backtrack ? d->p.backtrack_buffer_size : 0);
}

int ofp_stream_state_init(ofp_detector* d, void* d_state, void* stream) {
    OFP_REQUIRE(d && d_state, "ofp_stream_state_init: NULL argument");
    StreamArgs a = make_args(d);
    a.state = d_state;
    hipLaunchKernelGGL(k_stream_init, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    OFP_LAUNCH_CHECK("k_stream_init");
    return OFP_OK;
}

int ofp_stream_process(ofp_detector* d, void* d_state, const float* d_x, int64_t n_blocks,
                       int64_t n_rows, int32_t warmup, int64_t sample_base, float* d_rel,
                       ofp_onset* d_records, int64_t cap, int64_t* d_count, void* stream) {
    OFP_REQUIRE(d && d_state, "ofp_stream_process: NULL argument");
    OFP_REQUIRE(d->p.n_channels <= 1024, "streaming form supports at most 1024 channels (got %d)",
                d->p.n_channels);
    StreamArgs a = make_args(d);
    a.state = d_state;
    a.x = d_x;
    a.n_blocks = n_blocks;
    a.n_rows = n_rows;
    a.warmup = warmup;
    a.sample_base = sample_base;
    a.rel = d_rel;
    a.records = d_records;
    a.cap = cap;
    a.count = d_count;
    if (warmup) {
        if (n_rows <= 0) return OFP_OK;
        OFP_REQUIRE(d_x, "ofp_stream_process: d_x is NULL");
    } else {
        if (n_blocks <= 0) return OFP_OK;
        OFP_REQUIRE(d_x && d_count && (d_records || cap == 0), "ofp_stream_process: NULL argument");
    }
    const int C = d->p.n_channels;
    const size_t par_lds = (size_t)3 * d->p.block_size * C * sizeof(float);
    const char* force = getenv("OFP_STREAM_KERNEL");  // "seq" / "par": diagnostics and tests
    const bool par_ok = !warmup && C <= PAR_MAX_C && par_lds <= 120 * 1024;
    if (par_ok && !(force && force[0] == 's')) {
        // enough lanes for the elementwise phases of a hop, at least two per channel for the role lanes
        const int threads = (int)std::min<int64_t>(1024, std::max<int64_t>(ofp::align_up(2 * C, 64),
                                                    std::min<int64_t>(256, ofp::align_up((int64_t)d->p.block_size * C, 64))));
        static ofp::LdsAttrCache attr;
        if (int rc = ofp::ensure_dynamic_lds(reinterpret_cast<const void*>(k_stream_par), par_lds, attr, 65536 - 16384)) return rc;
        hipLaunchKernelGGL(k_stream_par, dim3(1), dim3(threads), par_lds, (hipStream_t)stream, a);
        OFP_LAUNCH_CHECK("k_stream_par");
        return OFP_OK;
    }
    int threads = (int)ofp::align_up(C, 64);
    hipLaunchKernelGGL(k_stream, dim3(1), dim3(threads), 0, (hipStream_t)stream, a);
    OFP_LAUNCH_CHECK("k_stream");
    return OFP_OK;
}

int ofp_stream_calibrate(ofp_detector* d, void* d_state, const float* d_x, int64_t n_rows, int64_t r0, int64_t r1,
                         int64_t n_rev, float* d_scratch, float* d_rel, void* stream) {
    OFP_REQUIRE(d && d_state && d_x && d_scratch && d_rel, "ofp_stream_calibrate: NULL argument");
    OFP_REQUIRE(n_rows >= 1 && 0 <= r0 && r0 <= r1 && r1 <= n_rows && 0 <= n_rev && n_rev <= n_rows,
                "ofp_stream_calibrate: row ranges outside [0, %lld]", (long long)n_rows);
    CalArgs q;
    q.a = make_args(d);
    q.a.state = d_state;
    q.a.x = d_x;
    q.n = n_rows;
    q.r0 = r0;
    q.r1 = r1;
    q.n_rev = n_rev;
    q.xdb = d_scratch;
    q.rel = d_rel;
    hipLaunchKernelGGL(k_calibrate, dim3((unsigned)ofp::cdiv(d->p.n_channels, 64)), dim3(64), 0, (hipStream_t)stream, q);
    OFP_LAUNCH_CHECK("k_calibrate");
    return OFP_OK;
}

}  // extern "C"
