/*
 * ofp_oracle.c -- CPU restatement of the reference's onset-detection path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or
 * executed by the product (onset_fingerprinting_amd/).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / the reported CPU baseline.
 *
 * Parity status: PINNED.  Every function below is checked against golden
 * vectors captured from the reference itself in the build container
 * (tests/golden/make_golden.py -> tests/golden/ *.npz, tests/test_oracle_golden.py):
 * followers, tracker, backtracking and IIR bit-for-bit; onset indices
 * index-for-index; the dB / linear conversions within 2 ulp because the
 * reference's numpy float32 log10/power are not correctly rounded and depend
 * on the host CPU (include/ofp_math.h states the canon used instead).
 *
 * Each function cites the reference lines it follows
 * (paths under /root/reference/onset_fingerprinting/).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no -Ofast: results
 * must not depend on value-changing optimisations).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ofp_math.h"

/* ---- envelope_follower.c:6-25 ------------------------------------------- */
void oracle_ar_envelope(const float* x, float* y, float attack, float release,
                        int size, int num_samples) {
    for (int j = 0; j < num_samples; ++j) {
        for (int i = 0; i < size; ++i) {
            int index = j * size + i;
            /* row 0 continues from the LAST row written by the previous call */
            int prev_index = (j > 0) ? (j - 1) * size + i : (num_samples - 1) * size + i;
            y[index] = ofp_ar_step(x[index], y[prev_index], attack, release);
        }
    }
}

/* ---- envelope_follower.c:27-57 ------------------------------------------ */
void oracle_minmax_envelope(const float* x, float* min_val, float* max_val,
                            float alpha_min, float alpha_max, float minmin,
                            int n_samples, int n_channels) {
    float ialpha_min = ofp_ialpha(alpha_min);
    float ialpha_max = ofp_ialpha(alpha_max);
    for (int j = 0; j < n_channels; ++j) {
        float mn = min_val[j], mx = max_val[j];
        for (int i = 0; i < n_samples; ++i) {
            float xi = x[i * n_channels + j];
            mn = ofp_min_step(xi, mn, ialpha_min, alpha_min, minmin);
            mx = ofp_max_step(xi, mx, ialpha_max, alpha_max);
        }
        min_val[j] = mn;
        max_val[j] = mx;
    }
}

/* ---- envelope_follower.c:59-85 (== detection.py:800-825) ----------------- */
void oracle_backtrack_onsets(const float* buffer, const long* channels, long* deltas,
                             float alpha, float tol, long buffer_length,
                             long n_onsets, long n_channels, long block_size) {
    float omba = (float)(1.0 - (double)alpha);
    long N = buffer_length;
    for (long j = 0; j < n_onsets; j++) {
        long channel = channels[j];
        long i = block_size - deltas[j];
        long idx = (N - i) * n_channels + channel;
        float current_smoothed = buffer[idx];
        idx -= n_channels;
        float prev = buffer[idx];
        float prev_smoothed = alpha * prev + omba * current_smoothed;
        while ((current_smoothed > prev_smoothed) &&
               (fabsf(prev_smoothed - prev) > tol) && (i + 1 < N)) {
            deltas[j] -= 1;
            i += 1;
            idx -= n_channels;
            current_smoothed = prev_smoothed;
            prev = buffer[idx];
            prev_smoothed = alpha * prev + omba * current_smoothed;
        }
    }
}

/* ---- detection.py:800-825: the Python backtracking that AmplitudeOnsetDetector
 * actually runs.  Same arithmetic as the C function above, but `i` is advanced
 * BEFORE the loop (detection.py:813), so the walk stops one row earlier than
 * envelope_follower.c:73-76 allows.  `buffer` is the last N rows of the relative
 * envelope, oldest first (detection.py:802-803).  The C variant is pinned by
 * golden set g9; this variant cannot be run from the reference here (its ring
 * buffer class comes from the absent `loopmate` package) and is pinned only
 * through the shared arithmetic: PARITY UNPINNED for the loop bound. */
void oracle_backtrack_onsets_py(const float* buffer, const long* channels, long* deltas,
                                float alpha, float tol, long buffer_length,
                                long n_onsets, long n_channels, long block_size) {
    float omba = (float)(1.0 - (double)alpha); /* np.float32(1 - self.b_alpha) */
    long N = buffer_length;
    for (long j = 0; j < n_onsets; j++) {
        long channel = channels[j];
        long i = block_size - deltas[j];
        float current_smoothed = buffer[(N - i) * n_channels + channel];
        i += 1;
        float prev = (N - i) >= 0 ? buffer[(N - i) * n_channels + channel] : 0.0f;
        float prev_smoothed = alpha * prev + omba * current_smoothed;
        while ((current_smoothed > prev_smoothed) &&
               (fabsf(prev_smoothed - prev) > tol) && (i + 1 < N)) {
            deltas[j] -= 1;
            i += 1;
            current_smoothed = prev_smoothed;
            prev = buffer[(N - i) * n_channels + channel];
            prev_smoothed = alpha * prev + omba * current_smoothed;
        }
    }
}

/* ---- detection.py:487-501: scipy.signal.lfilter(b, a, x, axis=0, zi) in
 * float32, order 4; zi is [4][C]; x,y are [n][C].  b,a are normalised by a[0]
 * in fp32 first, as scipy's C kernel does. */
void oracle_lfilter4(const float* x, float* y, const float* b_in, const float* a_in,
                     float* zi, long n, int C) {
    float b[5], a[5];
    for (int k = 0; k < 5; ++k) {
        b[k] = b_in[k] / a_in[0];
        a[k] = a_in[k] / a_in[0];
    }
    for (int c = 0; c < C; ++c) {
        float z[4] = {zi[0 * C + c], zi[1 * C + c], zi[2 * C + c], zi[3 * C + c]};
        for (long t = 0; t < n; ++t) y[t * C + c] = ofp_df2t4_step(x[t * C + c], b, a, z);
        for (int k = 0; k < 4; ++k) zi[k * C + c] = z[k];
    }
}

/* ---- detection.py:747-748 / 753-754, elementwise -------------------------- */
void oracle_rect_db(const float* x, float* y, long n, float floor_db) {
    for (long i = 0; i < n; ++i) y[i] = ofp_rect_db(x[i], floor_db);
}
void oracle_rel_linear(const float* d, float* y, long n, float floor_db) {
    for (long i = 0; i < n; ++i) y[i] = ofp_rel_linear(d[i], floor_db);
}
void oracle_log10f(const float* x, float* y, long n) {
    for (long i = 0; i < n; ++i) y[i] = ofp_log10f(x[i]);
}
void oracle_exp10f(const float* x, float* y, long n) {
    for (long i = 0; i < n; ++i) y[i] = ofp_exp10f(x[i]);
}

/* ---- AmplitudeOnsetDetector: detection.py:595-888 ------------------------- */
typedef struct {
    int C, B;
    float floor_db;
    int hp_on;
    float b[5], a[5];          /* raw fp32 butter coefficients (detection.py:496) */
    float fast_att, fast_rel;  /* np.float32(1/attack) ... (detection.py:514-515) */
    float slow_att, slow_rel;
    float alpha_min, alpha_max, minmin; /* detection.py:703-708 */
    int manual;                /* detection.py:687 */
    long cooldown;
    int backtrack;
    long bt_N;                 /* backtrack_buffer_size */
    float bt_alpha, bt_tol;    /* detection.py:722-725 */
} oracle_params;

typedef struct {
    oracle_params p;
    float* zi;      /* [4][C] */
    float* yf;      /* [C] fast follower: last row of its y array */
    float* ys;      /* [C] */
    float* mn;      /* [C] */
    float* mx;      /* [C] */
    float* on_f;    /* [C] on threshold (manual) or factor (relative), as fp32 */
    float* off_f;   /* [C] */
    double* on_d;   /* [C] manual mode: the Python double, used for row 0 (detection.py:769) */
    uint8_t* state; /* [C] detection.py:710 */
    double* prev;   /* [C] detection.py:711 (float64 array) */
    long* deb;      /* [C] detection.py:712 */
    float* hist;    /* [bt_N][C] backtracking history, oldest first */
    float* tmp;     /* [B][C] scratch */
    float* tmp2;    /* [B][C] scratch */
    /* The two transcendental elementwise maps.  Default: the fp64-evaluated canon
     * of include/ofp_math.h.  Tests may substitute the host's numpy float32
     * log10/power (what the reference itself runs) to pin every OTHER step of the
     * restatement bit-for-bit against the golden vectors. */
    void (*db_fn)(const float*, float*, long, float);
    void (*lin_fn)(const float*, float*, long, float);
} oracle_detector;

void oracle_detector_destroy(oracle_detector* d) {
    if (!d) return;
    free(d->zi); free(d->yf); free(d->ys); free(d->mn); free(d->mx);
    free(d->on_f); free(d->off_f); free(d->on_d);
    free(d->state); free(d->prev); free(d->deb); free(d->hist);
    free(d->tmp); free(d->tmp2);
    free(d);
}

/* detection.py:631-725.  on_thr/off_thr: C doubles (a scalar broadcast by the caller). */
oracle_detector* oracle_detector_create(const oracle_params* p, const double* on_thr,
                                        const double* off_thr) {
    oracle_detector* d = (oracle_detector*)calloc(1, sizeof(*d));
    d->p = *p;
    int C = p->C, B = p->B;
    d->zi = (float*)calloc(4 * C, sizeof(float)); /* detection.py:497 */
    d->yf = (float*)malloc(C * sizeof(float));
    d->ys = (float*)malloc(C * sizeof(float));
    d->mn = (float*)malloc(C * sizeof(float));
    d->mx = (float*)malloc(C * sizeof(float));
    d->on_f = (float*)malloc(C * sizeof(float));
    d->off_f = (float*)malloc(C * sizeof(float));
    d->on_d = (double*)malloc(C * sizeof(double));
    d->state = (uint8_t*)calloc(C, 1);
    d->prev = (double*)calloc(C, sizeof(double));
    d->deb = (long*)calloc(C, sizeof(long));
    d->hist = p->backtrack ? (float*)calloc((size_t)p->bt_N * C, sizeof(float)) : NULL;
    d->tmp = (float*)malloc((size_t)B * C * sizeof(float));
    d->tmp2 = (float*)malloc((size_t)B * C * sizeof(float));
    for (int c = 0; c < C; ++c) {
        d->yf[c] = p->floor_db; /* detection.py:697-702 */
        d->ys[c] = p->floor_db;
        d->mn[c] = 0.0f;        /* detection.py:704: x0 = [[0..],[10..]] */
        d->mx[c] = 10.0f;
        d->on_f[c] = (float)on_thr[c];
        d->off_f[c] = (float)off_thr[c];
        d->on_d[c] = on_thr[c];
    }
    d->db_fn = oracle_rect_db;
    d->lin_fn = oracle_rel_linear;
    return d;
}

/* fast_slide(x) - slow_slide(x) -> linear, for one block (detection.py:751-754,
 * 835-838); xdb is [B][C] rectified dB; rel out [B][C]. */
static void oracle_rel_block(oracle_detector* d, const float* xdb, float* rel) {
    const oracle_params* p = &d->p;
    int C = p->C, B = p->B;
    for (int t = 0; t < B; ++t) {
        for (int c = 0; c < C; ++c) {
            float x = xdb[t * C + c];
            d->yf[c] = ofp_ar_step(x, d->yf[c], p->fast_att, p->fast_rel);
            d->ys[c] = ofp_ar_step(x, d->ys[c], p->slow_att, p->slow_rel);
            rel[t * C + c] = d->yf[c] - d->ys[c];
        }
    }
    d->lin_fn(rel, rel, (long)B * C, p->floor_db);
}

/* init_minmax_tracker: detection.py:827-840.  x is [n][C]. */
void oracle_detector_warmup(oracle_detector* d, const float* x, long n) {
    const oracle_params* p = &d->p;
    int C = p->C, B = p->B;
    if (n <= 0) return;
    float* xf = (float*)malloc((size_t)n * C * sizeof(float));
    if (p->hp_on) oracle_lfilter4(x, xf, p->b, p->a, d->zi, n, C);
    else memcpy(xf, x, (size_t)n * C * sizeof(float));
    d->db_fn(xf, xf, n * C, p->floor_db);
    for (long i = 0; i + B <= n; i += B) {
        oracle_rel_block(d, xf + i * C, d->tmp);
        oracle_minmax_envelope(d->tmp, d->mn, d->mx, p->alpha_min, p->alpha_max,
                               p->minmin, B, C);
    }
    free(xf);
}

/* init: detection.py:842-888, the passes over the samples.  x is [n][C]; the caller guarantees
 * that n and every range below are whole blocks (the reference's follower calls always process
 * block_size rows, detection.py:534-537, so anything else reads past its buffers there).
 *   high-pass over all rows (:849-850), unclipped rectified dB (:852: no clip here);
 *   followers over rows [r0, r1) (:855-860: "assumes that first half second is silent");
 *   followers over all rows, rel_db[n][C] = fast - slow in dB (:862-867);
 *   followers over rows n_rev-1 .. 0 (:883-888: continuity with the starting point).
 * The thresholds follow from statistics of rel_db (:869-881) and are set by the caller
 * (oracle_detector_set_thresholds); tracker and hysteresis state are untouched. */
void oracle_detector_calibrate(oracle_detector* d, const float* x, long n, long r0, long r1,
                               long n_rev, float* rel_db) {
    const oracle_params* p = &d->p;
    int C = p->C;
    if (n <= 0) return;
    float* xf = (float*)malloc((size_t)n * C * sizeof(float));
    if (p->hp_on) oracle_lfilter4(x, xf, p->b, p->a, d->zi, n, C);
    else memcpy(xf, x, (size_t)n * C * sizeof(float));
    d->db_fn(xf, xf, n * C, -INFINITY);
    for (long t = r0; t < r1; ++t)
        for (int c = 0; c < C; ++c) {
            d->yf[c] = ofp_ar_step(xf[t * C + c], d->yf[c], p->fast_att, p->fast_rel);
            d->ys[c] = ofp_ar_step(xf[t * C + c], d->ys[c], p->slow_att, p->slow_rel);
        }
    for (long t = 0; t < n; ++t)
        for (int c = 0; c < C; ++c) {
            d->yf[c] = ofp_ar_step(xf[t * C + c], d->yf[c], p->fast_att, p->fast_rel);
            d->ys[c] = ofp_ar_step(xf[t * C + c], d->ys[c], p->slow_att, p->slow_rel);
            rel_db[t * C + c] = d->yf[c] - d->ys[c];
        }
    for (long t = n_rev - 1; t >= 0; --t)
        for (int c = 0; c < C; ++c) {
            d->yf[c] = ofp_ar_step(xf[t * C + c], d->yf[c], p->fast_att, p->fast_rel);
            d->ys[c] = ofp_ar_step(xf[t * C + c], d->ys[c], p->slow_att, p->slow_rel);
        }
    free(xf);
}

/* self.on_threshold / self.off_threshold replaced by per-channel arrays (detection.py:871-872) */
void oracle_detector_set_thresholds(oracle_detector* d, const double* on_thr, const double* off_thr) {
    for (int c = 0; c < d->p.C; ++c) {
        d->on_f[c] = (float)on_thr[c];
        d->off_f[c] = (float)off_thr[c];
        d->on_d[c] = on_thr[c];
    }
}

/* __call__: detection.py:727-798.  x [B][C] -> rel [B][C]; channels/deltas
 * (capacity C) ; returns number of onsets in this block. */
long oracle_detector_block(oracle_detector* d, const float* x, float* rel,
                           long* channels, long* deltas) {
    const oracle_params* p = &d->p;
    int C = p->C, B = p->B;
    float* xf = d->tmp2;
    if (p->hp_on) oracle_lfilter4(x, xf, p->b, p->a, d->zi, B, C); /* :743-744 */
    else memcpy(xf, x, (size_t)B * C * sizeof(float));
    d->db_fn(xf, xf, (long)B * C, p->floor_db);                     /* :747-748 */
    oracle_rel_block(d, xf, rel);                                   /* :751-754 */
    if (p->backtrack) {                                             /* :755-756 */
        long N = p->bt_N;
        memmove(d->hist, d->hist + (size_t)B * C, (size_t)(N - B) * C * sizeof(float));
        memcpy(d->hist + (size_t)(N - B) * C, rel, (size_t)B * C * sizeof(float));
    }
    float on[C], off[C];
    double on0[C];
    if (p->manual) {                                                /* :759-760 */
        for (int c = 0; c < C; ++c) { on[c] = d->on_f[c]; on0[c] = d->on_d[c]; }
    } else {                                                        /* :762-763 */
        oracle_minmax_envelope(rel, d->mn, d->mx, p->alpha_min, p->alpha_max,
                               p->minmin, B, C);
        for (int c = 0; c < C; ++c) {
            float t = d->mx[c] * d->on_f[c];
            on[c] = t + d->mn[c];
            on0[c] = (double)on[c];
        }
    }
    long on_idx[C];
    int onflag[C];
    long on_idx_max = 0;
    for (int c = 0; c < C; ++c) {
        int gate = (!d->state[c]) && (d->deb[c] < 1);               /* :764-768 */
        long first = -1;
        if (gate) {
            for (int t = 0; t < B; ++t) {
                int below_before = (t == 0) ? (d->prev[c] < on0[c])            /* :769 */
                                            : (rel[(t - 1) * C + c] < on[c]);  /* :770 */
                if ((rel[t * C + c] > on[c]) && below_before) { first = t; break; }
            }
        }
        on_idx[c] = first < 0 ? 0 : first;                          /* :774 argmax */
        onflag[c] = first >= 0;                                     /* :775 */
        if (on_idx[c] > on_idx_max) on_idx_max = on_idx[c];
    }
    for (int c = 0; c < C; ++c)                                     /* :778-779 */
        if (onflag[c]) { d->state[c] = 1; d->deb[c] = p->cooldown; }
    for (int c = 0; c < C; ++c)                                     /* :780 */
        if (d->deb[c] > 0) d->deb[c] -= B;
    for (int c = 0; c < C; ++c) {                                   /* :784-791 */
        if (p->manual) off[c] = d->off_f[c];
        else { float t = d->mx[c] * d->off_f[c]; off[c] = t + d->mn[c]; }
        int any = 0;
        for (long t = on_idx_max; t < B; ++t)                       /* :790 */
            if (rel[t * C + c] < off[c]) { any = 1; break; }
        if (any) d->state[c] = 0;
        d->prev[c] = (double)rel[(B - 1) * C + c];                  /* :792 */
    }
    long k = 0;
    for (int c = 0; c < C; ++c)                                     /* :795 */
        if (onflag[c]) { channels[k] = c; deltas[k] = on_idx[c]; ++k; }
    if (p->backtrack && k > 0)                                      /* :796-797 */
        oracle_backtrack_onsets_py(d->hist, channels, deltas, p->bt_alpha, p->bt_tol,
                                   p->bt_N, k, C, B);
    return k;
}

/* detect_onsets_amplitude: detection.py:19-86 (after construction).
 * x [N][C]; warm = int(0.5*sr) clipped to N by the caller (x[:int(0.5*sr)]).
 * rel must hold floor(N/B)*B*C floats; out arrays hold up to `cap` onsets.
 * Returns the number of onsets (may exceed cap: only cap are stored). */
long oracle_detect(oracle_detector* d, const float* x, long N, long warm, float* rel,
                   long* out_channels, long* out_onsets, long cap) {
    int C = d->p.C, B = d->p.B;
    oracle_detector_warmup(d, x, warm);                             /* :70 */
    long ch[C], de[C];
    long total = 0;
    for (long i = 0; i + B <= N; i += B) {                          /* :73-75 */
        long k = oracle_detector_block(d, x + i * C, rel + i * C, ch, de);
        for (long j = 0; j < k; ++j) {                              /* :78-82 */
            if (total < cap) { out_channels[total] = ch[j]; out_onsets[total] = i + de[j]; }
            ++total;
        }
    }
    return total;
}

/* substitute the elementwise maps (tests only; NULL keeps the canon) */
void oracle_detector_set_math(oracle_detector* d,
                              void (*db_fn)(const float*, float*, long, float),
                              void (*lin_fn)(const float*, float*, long, float)) {
    d->db_fn = db_fn ? db_fn : oracle_rect_db;
    d->lin_fn = lin_fn ? lin_fn : oracle_rel_linear;
}

/* state accessors for tests */
void oracle_detector_get_state(const oracle_detector* d, float* zi, float* yf, float* ys,
                               float* mn, float* mx, uint8_t* state, double* prev, long* deb) {
    int C = d->p.C;
    if (zi) memcpy(zi, d->zi, 4 * C * sizeof(float));
    if (yf) memcpy(yf, d->yf, C * sizeof(float));
    if (ys) memcpy(ys, d->ys, C * sizeof(float));
    if (mn) memcpy(mn, d->mn, C * sizeof(float));
    if (mx) memcpy(mx, d->mx, C * sizeof(float));
    if (state) memcpy(state, d->state, C);
    if (prev) memcpy(prev, d->prev, C * sizeof(double));
    if (deb) memcpy(deb, d->deb, C * sizeof(long));
}

/* ---- cross-correlation ------------------------------------------------
 * cross_correlation_lag, reference detection.py:244-250: entries [lo, hi) of
 *   cc = np.correlate(x, y, "full");  cc[:n] /= normalizer;  cc[n:] /= normalizer[n-2::-1]
 * with normalizer[i] = max(i + 1, cutoff) for i < cutoff, i + 1 beyond (detection.py:247-248).
 * Canon shared with the HIP kernel: every dot product is accumulated in fp64 over ascending i
 * (fp32 x fp32 products are exact in fp64), rounded once to fp32 -- the dtype np.correlate
 * returns for float32 input -- and divided in fp32 (numpy divides the float32 value by the int64
 * count in fp64 and rounds to fp32, which is the correctly rounded fp32 quotient). */
void oracle_xcorr_slice(const float* x, const float* y, long n, long cutoff, long lo, long hi, float* cc) {
    for (long j = lo; j < hi; ++j) {
        long k = j - (n - 1); /* cc[j] = sum_i x[i + k] * y[i] */
        long i0 = k < 0 ? -k : 0, i1 = k > 0 ? n - k : n;
        double acc = 0.0;
        for (long i = i0; i < i1; ++i) acc += (double)x[i + k] * (double)y[i];
        long m = j < n ? j : 2 * n - 2 - j; /* index into the normalizer */
        long cnt = m < cutoff ? cutoff : m + 1;
        cc[j - lo] = (float)acc / (float)cnt;
    }
}
