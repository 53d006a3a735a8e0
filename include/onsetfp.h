/*
 * onsetfp.h -- C ABI of libonsetfp.so, the MI355X (gfx950) implementation of the
 * onset-fingerprinting hot path:
 *
 *   envelope follower / amplitude onset detector  ->  framing + Hann + rFFT
 *   power spectrum  ->  mel fingerprint  ->  small classifier forward.
 *
 * Plain pointers and sizes only; no torch or C++ types.  All `d_` pointers are
 * DEVICE pointers (HBM); `stream` is a hipStream_t passed as void* (NULL = the
 * null stream).  Every entry point except the three legacy symbols returns an
 * int status (OFP_OK == 0) and records a message retrievable with
 * ofp_last_error().  The library never allocates inside a launch function:
 * work space is provided by the caller (size from the *_workspace_bytes
 * query), so every launch function can be captured into a hipGraph.
 *
 * Reference interfaces replaced (paths under the upstream repo's
 * onset_fingerprinting/ directory):
 *   envelope_follower.c:6,27,59     the ctypes boundary of detection.py:517-578
 *   detection.py:595-888            AmplitudeOnsetDetector (__call__, warm-up)
 *   detection.py:19-86              detect_onsets_amplitude driver loop
 *   data.py:55-120                  FrameExtractor gather
 *   data.py:581-654                 stft_frame / stft
 *   data.py:657-680                 cspec_to_mfcc (mel + dB + DCT)
 *   calibration.py:463-560          FCNN forward
 *   model.py:52-120                 CNN forward
 */
#ifndef ONSETFP_H
#define ONSETFP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFP_OK 0
#define OFP_ERR_INVALID 1   /* bad argument (shape, NULL, unsupported option) */
#define OFP_ERR_HIP 2       /* a HIP runtime call or kernel launch failed */
#define OFP_ERR_NODEVICE 3  /* no usable GPU */
#define OFP_ERR_WORKSPACE 4 /* caller-provided work space too small */
#define OFP_ERR_NOCONVERGE 5 /* speculative time-parallel pass did not converge */

#define OFP_ABI_VERSION 3

/* ---- status ---------------------------------------------------------------- */
int ofp_abi_version(void);
const char* ofp_last_error(void);
/* number of visible GPUs (>= 0) or a negative OFP_ERR_* */
int ofp_device_count(void);
/* writes the gcnArchName of `device` into buf; fails unless it is gfx950 */
int ofp_device_check(int device, char* buf, int buflen);

/* ---- legacy symbols: drop-in for envelope_follower.so ---------------------------
 * Identical names, signatures and HOST-pointer semantics as envelope_follower.c:6,
 * :27 and :59, so that the reference's detection.py:517-578 can CDLL this library
 * unchanged.  Each call stages its arrays to the GPU, runs one kernel and copies
 * the in/out arrays back; they return void and report failures only through
 * ofp_last_error() (the reference ABI has no status).  */
void ar_envelope(float* x, float* y, float attack, float release, int size, int num_samples);
void minmax_envelope(float* x, float* min_val, float* max_val, float alpha_min, float alpha_max,
                     float minmin, int n_samples, int n_channels);
void backtrack_onsets(float* buffer, long* channels, long* deltas, float alpha, float tol,
                      long buffer_length, long n_onsets, long n_channels, long block_size);

/* ButterworthFilter.__call__ (detection.py:499-501): scipy.signal.lfilter(b, a, x, axis=0, zi) in
 * float32 (direct form II transposed, coefficients normalised by a[0] in fp32), HOST pointers like
 * the legacy symbols: x,y [n][n_channels]; b,a [order+1]; zi [order][n_channels] in/out; order <= 8 */
int ofp_lfilter(const float* x, float* y, const float* b, const float* a, int order, float* zi, long n,
                int n_channels);

/* ---- amplitude onset detector (detection.py:595-888) ----------------------------- */
typedef struct ofp_detector_params {
    int32_t n_channels;   /* C: signals per detector instance (detection.py:633) */
    int32_t block_size;   /* B (detection.py:634) */
    float floor_db;       /* floor (detection.py:635) */
    int32_t hp_enabled;   /* hipass_freq != 0 (detection.py:692-696) */
    float hp_b[5];        /* np.float32(butter(4, f, "high", fs=sr)) (detection.py:492-496) */
    float hp_a[5];
    float fast_attack;    /* np.float32(1/attack): the COEFFICIENT (detection.py:514-515) */
    float fast_release;
    float slow_attack;
    float slow_release;
    float alpha_min;      /* 1e-4 (detection.py:705) */
    float alpha_max;      /* 1e-5 */
    float minmin;         /* 2 */
    float min0;           /* tracker start: 0 (detection.py:704) */
    float max0;           /* 10 */
    int32_t manual;       /* scalar on_threshold > 1 (detection.py:687) */
    int64_t cooldown;     /* samples (detection.py:640) */
    int32_t backtrack;    /* detection.py:641 */
    int64_t backtrack_buffer_size;
    float backtrack_alpha; /* np.float32(2/(smooth+1)) (detection.py:722) */
    float backtrack_tol;   /* np.float32((1-alpha)**buffer_size) (detection.py:723-725) */
} ofp_detector_params;

/* one detected onset: 16 bytes, also the record all-gathered across ranks */
typedef struct ofp_onset {
    int32_t clip;     /* clip (detector instance) index within the call */
    int32_t channel;  /* channel index in [0, C) */
    int64_t sample;   /* block_start + delta, relative to the clip's first sample */
} ofp_onset;

/* tuning of the speculative time-parallel passes; zero-initialise for defaults */
typedef struct ofp_detect_tuning {
    int64_t hp_chunk, hp_warm;   /* samples per chunk / speculative warm-up, IIR stage */
    int64_t ar_chunk, ar_warm;   /* follower stage */
    int64_t mm_chunk, mm_warm;   /* min/max tracker stage */
    int32_t max_passes;          /* repair passes before giving up (0: no limit) */
    int64_t ar_coarse_warm;      /* follower stage: approximate-arithmetic warm-up that
                                    produces the guess for the exact warm-up (<0: none) */
    int64_t hp_candidates;       /* IIR stage: speculative candidates per chunk (default 16, max 16) */
    int64_t hp_candidate_offset; /* IIR stage: samples between candidate starts (default 8); < 0: all
                                    candidates of a chunk start at the same sample from slightly
                                    different states */
    int64_t ar_guess;            /* follower stage, how the exact warm-up gets its starting guess:
                                    0 auto, 1 sequential approximate pass, 2 closed-form dot product
                                    (needs slow attack == slow release; auto picks it when they are) */
    int64_t hp_span;             /* IIR stage: 1 (default), 2 or 4 = chunks one speculative run walks through
                                    after its warm-up; > 1 shares the warm-up between chunks: less work,
                                    fewer waves, longer launch (throughput instead of latency) */
    int64_t ar_span, mm_span;    /* follower / tracker stage: chunks one speculative warm-up run walks through
                                    (0: chosen from the batch size, see make_layout) */
    int64_t verify_group;        /* verification passes / rounds enqueued per host synchronisation (0: default 2-3;
                                    1: one host round trip per pass, the round-1 behaviour) */
    int64_t hp_dedupe;           /* IIR stage: speculative candidates in stages with duplicate runs removed between
                                    them (a third of the steps, six launches instead of one): 0 auto (batches whose
                                    candidate launch is throughput-bound, when concurrent_calls >= 2), 1 always,
                                    < 0 never */
    int64_t hp_early;            /* IIR stage: a chunk re-run whole from its true start state stops at the first
                                    sub-chunk boundary where it has joined a candidate's recorded trajectory:
                                    0 auto (chunks of 32768 samples and more), 1 always, < 0 never */
    int64_t lane_merge;          /* follower / tracker stage: the two recurrences of a chunk (fast / slow follower,
                                    min / max) in one lane instead of two: half the reads of those passes, a longer
                                    dependent chain per lane: more frames/s when calls overlap, a slower lone call.
                                    0 auto (on when concurrent_calls >= 2), 1 always, < 0 never */
    int64_t fuse_db_sums;        /* the dB pass also forms the per-chunk sums of the slow follower's closed-form guess
                                    (one pass over the filtered stream instead of two): 0 on, < 0 off */
    int64_t sm_segments;         /* hysteresis / cooldown machine time-parallel over the list of visited blocks (clips of
                                    up to 64 channels): 0 auto (clips of 16384 blocks and more), 1 always, < 0 never; 2 = as 1 and then
                                    the sequential machine as if the segments had not converged (tests) */
    int64_t concurrent_calls;    /* how many detector calls of about this size the caller keeps in flight on the GPU
                                    at once (0 / 1: this call has the GPU to itself).  The layout of the
                                    speculative passes is chosen for the GPU's share: with k calls in flight each
                                    gets 1/k of the lane budget, i.e. the work-efficient layout of a k times larger
                                    batch instead of the latency layout of a lone call.  Results do not change. */
    int64_t scan_skip;           /* crossing pass: blocks whose extremes (left by the back-to-linear pass) show that they
                                    hold no value above `on` and whose last row is below `off` are decided without
                                    reading their samples: 0 on, < 0 off */
    int64_t host_verify;         /* who drives the verification passes of the three time-parallel stages: 0 (default)
                                    chain-local kernels -- one workgroup owns whole chains and iterates its passes
                                    between workgroup barriers until nothing changes: no host round trip, one launch per
                                    stage, the call can be captured in a hipGraph; 1 the round-1/2 form: one launch per
                                    pass, the host reads a change counter per group of passes (verify_group,
                                    max_passes apply to this form only).  Results do not change. */
    int64_t interleaved;         /* 4 or 8 channels: stages that work on the caller's interleaved arrays instead of planar
                                    copies.  `rel` side (throughput layout, lane_merge; `rel` requested): tracker, crossing
                                    pass and backtracking read the `rel` output, no planar copy of it is written (+7 %
                                    frames/s in flight).  Input side (staged candidates, high-pass on): the IIR stage reads
                                    the audio as it is, no planar copy of the input is made -- slower (one 4-byte load per
                                    step and lane), kept for measurement.  0 auto (the `rel` side whenever its conditions
                                    hold), < 0 never, 1 the `rel` side, 2 the input side, 3 both.  Results do not change. */
    int64_t walk_through;        /* throughput layout: the chunks a speculative warm-up run walks through after its warm-up
                                    (all but the last of every group of `span` chunks) count as their pass 0 -- the run
                                    leaves their outputs and end states, the chunk pass runs the others only: one pass
                                    over the stream less for those chunks (followers: the merged layout with the
                                    closed-form guess; tracker: on the interleaved envelope).  0 on, < 0 off.  Results
                                    do not change. */
    int64_t line_stores;         /* throughput layout (lane_merge), everything a multiple of 32 steps: the output walks of
                                    the IIR stage and the followers hand every batch of 32 outputs over through LDS and
                                    the wave stores complete 128-byte lines (8 lanes x 16 B) instead of 64 lane-private
                                    16-byte pieces per instruction: +11 % frames/s in flight, +5 % for a lone BIG call
                                    (C3), slower for a small one.  0 auto (the throughput layout, or launches of more
                                    than two / half a wave per SIMD), 1 whenever the sizes allow, < 0 off.  Results do
                                    not change. */
} ofp_detect_tuning;

typedef struct ofp_detector ofp_detector; /* opaque */

/* on_threshold / off_threshold: C doubles each (the reference's scalar broadcast,
 * or per-channel values as set by AmplitudeOnsetDetector.init, detection.py:866-867) */
int ofp_detector_create(const ofp_detector_params* p, const double* on_threshold,
                        const double* off_threshold, ofp_detector** out);
int ofp_detector_destroy(ofp_detector* det);
int ofp_detector_set_tuning(ofp_detector* det, const ofp_detect_tuning* t);

/* Work space (bytes) ofp_detect_offline needs for n_clips clips of n_samples. */
int64_t ofp_detect_workspace_bytes(const ofp_detector* det, int64_t n_clips, int64_t n_samples,
                                   int64_t warm_samples);

/* detect_onsets_amplitude (detection.py:19-86) for a batch of independent clips.
 *   d_x        [n_clips][n_samples][C] float32, interleaved as the reference expects
 *   warm       number of leading samples double-processed by init_minmax_tracker
 *              (detection.py:70: int(0.5*sr)); clipped to n_samples; 0 = no warm-up
 *   d_rel      [n_clips][floor(n_samples/B)*B][C] float32 relative envelope, or NULL
 *   d_records  [n_clips][cap_per_clip] onset records, ordered as the reference orders
 *              them (block, then channel)
 *   d_counts   [n_clips] int64 number of onsets per clip (may exceed cap_per_clip:
 *              only cap_per_clip are stored)
 *   d_ws       work space of at least ofp_detect_workspace_bytes()
 * Synchronises `stream` ONCE, at its end (the speculative passes are verified on the device by
 * chain-local kernels; tuning host_verify = 1 restores the host-verified pass groups of ABI 2); on return
 * all outputs are complete.  h_info (optional, host, int64
 * [OFP_DETECT_INFO_LEN]) receives {0: hp passes, 1: follower passes, 2: tracker
 * passes, 3: repaired chunks, 4..9: nanoseconds (HIP events on `stream`) spent in
 * the hp, dB, follower, linear, tracker and crossing/state-machine stages,
 * 10: total nanoseconds, 11: nanoseconds of the IIR candidate launch(es) (k_hp_candidates, the
 * longest single launch; or the stages k_hp_seg0 .. k_hp_seg_chunk), 12: IIR steps they execute
 * over all their lanes (17 fp32 operations each), 13: staged candidates only: distinct runs that
 * walked a chunk, of chains * chunks * candidates, 14: 1 if the segmented state machine did not converge within its
 * pre-enqueued passes and the sequential one decided, 15: non-zero if (a part of) the call was repeated with
 * host-verified passes because what had been enqueued ahead did not converge (1, + 1: the IIR rounds -- the whole call
 * again; + 2: the follower passes -- again from the follower stage; + 4: the tracker passes -- again from the tracker
 * stage; the detector then enqueues more passes in its next calls)}. */
#define OFP_DETECT_INFO_LEN 16
int ofp_detect_offline(ofp_detector* det, const float* d_x, int64_t n_clips, int64_t n_samples,
                       int64_t warm, float* d_rel, ofp_onset* d_records, int64_t cap_per_clip,
                       int64_t* d_counts, void* d_ws, int64_t ws_bytes, int64_t* h_info,
                       void* stream);

/* ofp_detect_offline without its synchronisation, in two calls.  _enqueue only ENQUEUES the whole call on `stream`
 * (nothing in it blocks or reads a device result on the host: the stream may be capturing a hipGraph, and a captured
 * graph may be replayed on new contents of the same buffers); after the caller has synchronised the stream (or the
 * graph launch), _complete with the same arguments reads the few words the call left in pinned host memory, fills
 * h_info (stage times only when the call was not captured) and, in the one case that needs a decision on the host --
 * the segmented state machine of a long clip did not converge within its pre-enqueued passes (info 14; never observed)
 * -- runs the sequential machine on `stream` and synchronises.  One enqueued call per detector at a time (it may be
 * completed once per replay of a graph that captured it).  Not with tuning host_verify.  ofp_detect_offline == _enqueue + hipStreamSynchronize + _complete. */
int ofp_detect_offline_enqueue(ofp_detector* det, const float* d_x, int64_t n_clips, int64_t n_samples,
                               int64_t warm, float* d_rel, ofp_onset* d_records, int64_t cap_per_clip,
                               int64_t* d_counts, void* d_ws, int64_t ws_bytes, void* stream);
int ofp_detect_offline_complete(ofp_detector* det, const float* d_x, int64_t n_clips, int64_t n_samples,
                                int64_t warm, float* d_rel, ofp_onset* d_records, int64_t cap_per_clip,
                                int64_t* d_counts, void* d_ws, int64_t ws_bytes, int64_t* h_info, void* stream);

/* The same call in two halves, so that a caller can overlap other work with the long, sparsely
 * occupied tail: _begin only ENQUEUES the head (input transpose + the IIR candidate launch, the
 * part that is heavy on the memory system) and returns; _finish does everything else and
 * synchronises.  Both take the arguments of ofp_detect_offline. */
int ofp_detect_offline_begin(ofp_detector* det, const float* d_x, int64_t n_clips, int64_t n_samples,
                             int64_t warm, void* d_ws, int64_t ws_bytes, void* stream);
int ofp_detect_offline_finish(ofp_detector* det, const float* d_x, int64_t n_clips, int64_t n_samples,
                              int64_t warm, float* d_rel, ofp_onset* d_records, int64_t cap_per_clip,
                              int64_t* d_counts, void* d_ws, int64_t ws_bytes, int64_t* h_info, void* stream);

/* _finish without its synchronisation (complete it with ofp_detect_offline_complete after synchronising). */
int ofp_detect_offline_finish_enqueue(ofp_detector* det, const float* d_x, int64_t n_clips, int64_t n_samples,
                                      int64_t warm, float* d_rel, ofp_onset* d_records, int64_t cap_per_clip,
                                      int64_t* d_counts, void* d_ws, int64_t ws_bytes, void* stream);

/* _begin in two calls: _begin_input enqueues the planar copy of the input only (ofp_detect_planar_input
 * is valid once it has run), _begin_iir the IIR candidate launch. */
int ofp_detect_offline_begin_input(ofp_detector* det, const float* d_x, int64_t n_clips, int64_t n_samples,
                                   int64_t warm, void* d_ws, int64_t ws_bytes, void* stream);
int ofp_detect_offline_begin_iir(ofp_detector* det, const float* d_x, int64_t n_clips, int64_t n_samples,
                                 int64_t warm, void* d_ws, int64_t ws_bytes, void* stream);
/* Collation block of a rank (the all-gather of onset indices): the onset records of a batch of
 * clips, d_records [n_clips][cap_per_clip] with d_counts [n_clips] as ofp_detect_offline leaves them, compacted
 * in clip order into d_block [1 + cap_total] without a host round trip (no data-dependent shape).  Record 0 is
 * the header {clip: 0, channel: 1 if some clip counted more onsets than cap_per_clip (lost records), sample:
 * total number of records}; records 1.. follow with clip_offset added to their clip ids; records that do not
 * fit cap_total are dropped (the header still carries the true total, so the reader can tell); rows beyond
 * the total are not written.  One launch. */
int ofp_pack_records(const ofp_onset* d_records, const int64_t* d_counts, int64_t n_clips, int64_t cap_per_clip,
                     int64_t cap_total, int32_t clip_offset, ofp_onset* d_block, void* stream);

/* Streaming form: AmplitudeOnsetDetector.__call__ (detection.py:727-798) on
 * n_blocks consecutive blocks with the detector state carried in d_state
 * (ofp_stream_state_bytes() bytes, initialised by ofp_stream_state_init).
 * One launch, no host synchronisation: capturable into a hipGraph.
 *   d_x [n_blocks*B][C]; d_rel same shape or NULL; d_records [cap]; d_count [1]
 *   (int64, ACCUMULATED: zero it to start a new list); record.sample is relative to
 *   sample_base + the first sample of this call.
 *   warmup != 0: run init_minmax_tracker (detection.py:827-840) over the n_rows
 *   rows of d_x instead (n_blocks is then ignored; rows beyond the last full block
 *   pass through the high-pass filter only). */
int64_t ofp_stream_state_bytes(const ofp_detector* det);
int ofp_stream_state_init(ofp_detector* det, void* d_state, void* stream);
int ofp_stream_process(ofp_detector* det, void* d_state, const float* d_x, int64_t n_blocks,
                       int64_t n_rows, int32_t warmup, int64_t sample_base, float* d_rel,
                       ofp_onset* d_records, int64_t cap, int64_t* d_count, void* stream);

/* AmplitudeOnsetDetector.init (detection.py:842-888), the passes over the samples, with the state in
 * d_state: high-pass over the n_rows rows of d_x (:849-850), unclipped rectified dB (:852), the
 * followers over rows [r0, r1) (the settling blocks, :855-860), over all rows with
 * d_rel[n_rows][C] = fast - slow in dB (:862-867), and over rows n_rev-1 .. 0 (:883-888).
 * d_scratch [n_rows][C].  Tracker and hysteresis state are untouched.  The thresholds init derives
 * from statistics of d_rel (:869-872) are installed with ofp_detector_set_thresholds.  Only enqueues. */
int ofp_stream_calibrate(ofp_detector* det, void* d_state, const float* d_x, int64_t n_rows, int64_t r0,
                         int64_t r1, int64_t n_rev, float* d_scratch, float* d_rel, void* stream);
/* Replaces the per-channel thresholds given to ofp_detector_create (host arrays of C doubles);
 * synchronous.  Applies to launches enqueued afterwards. */
int ofp_detector_set_thresholds(ofp_detector* det, const double* on_threshold, const double* off_threshold);

/* ---- framing + STFT (data.py:55-120, 581-654) ------------------------------------ */
/* Dense power spectra, the metric's frame definition (one frame per hop):
 *   d_x [n_clips][n_samples][C] interleaved;  frame h of channel c covers samples
 *   [h*hop, h*hop + n_fft);  H = 1 + (n_samples - n_fft)/hop.
 *   d_power [n_clips][C][H][n_fft/2+1] float32 = |rfft(hann_periodic(n_fft) * frame)|^2
 *   n_fft in {256, 512, 1024, 2048, 4096}.  d_power may be NULL when only the mel
 *   output below is wanted. */
int ofp_stft_power(const float* d_x, int64_t n_clips, int64_t n_samples, int32_t n_channels,
                   int32_t n_fft, int32_t hop, float* d_power, void* stream);
/* The same with the mel filterbank of ofp_mel applied while a frame's power spectrum is still on
 * chip: d_mel [n_clips][C][H][n_mels] (same values as ofp_mel on d_power); d_power may be NULL
 * when only the mel fingerprint is wanted.  fb_nnz = number of weights in d_fb_w. */
int ofp_stft_power_mel(const float* d_x, int64_t n_clips, int64_t n_samples, int32_t n_channels, int32_t n_fft,
                       int32_t hop, float* d_power, int32_t n_mels, const int32_t* d_fb_lo,
                       const int32_t* d_fb_len, const int32_t* d_fb_off, const float* d_fb_w, int32_t fb_nnz,
                       float* d_mel, int64_t planar_stride, void* stream);
/* ... and the classifier on top (the fingerprint path of SURVEY.md 8a rows a9 -> a11 -> a13 in one
 * launch): the 40 band sums of 16 frames at a time go through the whole FCNN while they are still in
 * LDS.  d_logits [n_clips][C][H][mlp outputs]; d_power and d_mel may each be NULL (nothing but the
 * logits then leaves the chip: 32 B per frame instead of 2 052 + 160).  mlp inputs == n_mels. */
typedef struct ofp_mlp ofp_mlp;
int ofp_stft_power_mel_mlp(const float* d_x, int64_t n_clips, int64_t n_samples, int32_t n_channels, int32_t n_fft,
                           int32_t hop, float* d_power, int32_t n_mels, const int32_t* d_fb_lo,
                           const int32_t* d_fb_len, const int32_t* d_fb_off, const float* d_fb_w, int32_t fb_nnz,
                           float* d_mel, int64_t planar_stride, const ofp_mlp* mlp, float* d_logits, void* stream);
/* planar_stride != 0: d_x points at one series per (clip, channel), planar_stride floats apart, e.g.
 * the planar copy the detector keeps in its work space between ofp_detect_offline_begin and the next
 * begin (each series there is preceded by the warm-up part of the detector's stream): */
const float* ofp_detect_planar_input(const ofp_detector* det, int64_t n_clips, int64_t n_samples, int64_t warm,
                                     const void* d_ws);
int64_t ofp_detect_planar_stride(const ofp_detector* det, int64_t n_clips, int64_t n_samples, int64_t warm);
/* (both return 0 / NULL when the detector's layout for these sizes works on the caller's interleaved array and makes no
 *  planar copy -- tuning `interleaved` 2 / 3; pass planar_stride 0 and the caller's array then: with 4 or 8 channels and
 *  hop = n_fft / 4 the dense STFT reads it almost as efficiently.) */

/* Gathered complex STFT frames (data.py:593-654 semantics are built on this by
 * the Python layer): for each of n_frames (clip, channel, start) triples,
 * d_spec[f][n_fft/2+1] complex64 = rfft(window * pad_center(x[start : start+frame_length]))
 * with samples outside [0, n_samples) read as zero.  d_window is float32[n_fft]
 * (the zero-padded periodic Hann, data.py:627-629).  starts may be negative. */
int ofp_stft_frames(const float* d_x, int64_t n_clips, int64_t n_samples, int32_t n_channels,
                    const int32_t* d_clip, const int32_t* d_channel, const int64_t* d_start,
                    const int64_t* d_valid_lo, const int64_t* d_valid_hi, int64_t n_frames,
                    int32_t frame_length, int32_t n_fft, const float* d_window, float* d_spec,
                    void* stream);

/* FrameExtractor gather (data.py:90-120): d_out[o][c][w] = x[clip][start[o][c] + w][c] */
int ofp_extract_frames(const float* d_x, int64_t n_samples, int32_t n_channels,
                       const int64_t* d_start /* [O][C] */, int64_t n_onsets, int32_t width,
                       float* d_out, void* stream);

/* ---- fingerprint: mel + dB + DCT (data.py:657-680) ------------------------------- */
/* d_power [n_rows][n_bins] -> d_mel [n_rows][n_mels] = power @ fb^T with the sparse
 * triangular filterbank given in CSR-by-band form (band b covers bins
 * [d_fb_lo[b], d_fb_lo[b]+d_fb_len[b]) with weights d_fb_w[d_fb_off[b] ...]). */
int ofp_mel(const float* d_power, int64_t n_rows, int32_t n_bins, int32_t n_mels,
            const int32_t* d_fb_lo, const int32_t* d_fb_len, const int32_t* d_fb_off,
            const float* d_fb_w, float* d_mel, void* stream);
/* power_to_db (ref 1, amin, top_db relative to the max over the n values) then
 * DCT-II ortho over the mel axis keeping n_mfcc: d_mel [n_rows][n_mels] ->
 * d_mfcc [n_rows][n_mfcc].  d_dct is float32 [n_mfcc][n_mels].  top_db < 0: no floor.
 * d_scratch: >= 4 bytes. */
int ofp_mfcc(const float* d_mel, int64_t n_rows, int32_t n_mels, int32_t n_mfcc, float amin,
             float top_db, const float* d_dct, float* d_mfcc, float* d_scratch, void* stream);

/* ---- classifier forward ---------------------------------------------------------- */
#define OFP_ACT_IDENTITY 0
#define OFP_ACT_RELU 1
#define OFP_ACT_SILU 2
#define OFP_ACT_LEAKYRELU 3
#define OFP_ACT_ELU 4
#define OFP_ACT_TANH 5
/* One fused dense layer: d_y[n][out] = act((d_x[n][in] @ W^T + b) * scale + shift)
 * (W [out][in] row-major as torch stores Linear.weight; scale/shift fold an eval-mode
 * BatchNorm1d, NULL = identity; b NULL = no bias).  fp32 MFMA (v_mfma_f32_16x16x4_f32). */
int ofp_dense(const float* d_x, int64_t n, int32_t in, int32_t out, const float* d_w,
              const float* d_b, const float* d_scale, const float* d_shift, int32_t act,
              float* d_y, void* stream);
/* The whole FCNN (calibration.py:463-527, eval mode) as one handle and ONE launch: n_layers fused
 * dense layers (1..8) with the activations kept in LDS; per output element the arithmetic is that of
 * the ofp_dense chain, so results are bit-identical to it.  All arrays are HOST pointers and are copied:
 * dims [n_layers+1] (dims[0] inputs ... dims[n_layers] outputs), act [n_layers] (OFP_ACT_*), and per
 * layer h_w[L] [dims[L+1]][dims[L]] (torch Linear.weight), h_b[L] / h_scale[L] / h_shift[L]
 * [dims[L+1]] or NULL (h_b, h_scale, h_shift themselves may be NULL).  ofp_mlp_forward: d_x [n][dims[0]]
 * -> d_y [n][dims[n_layers]]; fails with OFP_ERR_INVALID when ofp_mlp_lds_bytes() exceeds the 160 KiB of
 * LDS (run such a network layer by layer with ofp_dense). */
typedef struct ofp_mlp ofp_mlp; /* opaque */
int ofp_mlp_create(int32_t n_layers, const int32_t* dims, const int32_t* act, const float* const* h_w,
                   const float* const* h_b, const float* const* h_scale, const float* const* h_shift,
                   ofp_mlp** out);
int ofp_mlp_destroy(ofp_mlp* mlp);
int64_t ofp_mlp_lds_bytes(const ofp_mlp* mlp);
int ofp_mlp_forward(const ofp_mlp* mlp, const float* d_x, int64_t n, float* d_y, void* stream);

/* Conv1d (stride 1, groups 1) + bias + activation: d_x [n][cin][w] ->
 * d_y [n][cout][wout], wout = w + 2*padding - dilation*(k-1)   (model.py:84-95) */
int ofp_conv1d(const float* d_x, int64_t n, int32_t cin, int32_t w, const float* d_w /*[cout][cin/groups][k]*/,
               const float* d_b, int32_t cout, int32_t k, int32_t padding, int32_t dilation, int32_t groups,
               int32_t stride, int32_t act, const float* d_bn_scale /*[cout] or NULL*/, const float* d_bn_shift,
               int32_t pool /* MaxPool1d(2, 2) after the affine */, float* d_y, void* stream);
/* nn.GroupNorm(1, K) over items d_x [n][K][V] (CCCNN with batch_norm=True, model.py:497-501), then
 * optionally MaxPool1d(2, 2): d_y [n][K][V or V/2]; d_gamma / d_beta [K] or NULL.  Not in place. */
int ofp_groupnorm1(const float* d_x, int64_t n, int32_t K, int32_t V, const float* d_gamma, const float* d_beta,
                   float eps, int32_t pool, float* d_y, void* stream);

/* CCCNN correlation head (model.py:524-534): d_x [n][K][V] feature maps -> d_out [n][2V-1]:
 * full auto-correlation of every map, summed over the K maps, soft-maxed over the lags. */
int ofp_autocorr_softmax(const float* d_x, int64_t n, int32_t K, int32_t V, float* d_out, void* stream);

/* ---- per-hop streaming session (BASELINE config 5) -------------------------------------------
 * The reference's realtime pattern -- PortAudio callback: ring-buffer write (realtime/audio.py:97),
 * AmplitudeOnsetDetector on the hop (:62-74), classifier (multilateration.py:555-557 ->
 * calibration.py:552-560), one spectral frame of the trailing n_fft samples per hop
 * (realtime/recording.py:273-280) -- as ONE hipGraph per hop, captured at creation:
 *   H2D hop -> detector (state in HBM) -> per channel: ring write, Hann x audio[-n_fft:], rFFT,
 *   |X|^2, mel bands, FCNN -> D2H of {count, records, logits, mel, rel}.
 * The ring buffer ([ring_samples][C] float32, zeros at start: realtime/config.py:45,59) and all
 * detector state stay on the device.  The detector handle is borrowed and must outlive the session;
 * hop length and channel count are the detector's block_size and n_channels.  All h_ pointers are
 * HOST memory.  One hop may be in flight per session (submit -> collect; push = both). */
typedef struct ofp_hop_config {
    int32_t n_fft;          /* 256, 512, 1024, 2048 or 4096; periodic Hann of n_fft (data.py:627) */
    int64_t ring_samples;   /* rows of the ring buffer, >= max(n_fft, block_size) */
    int32_t n_mels;         /* mel filterbank, band-CSR as for ofp_mel; HOST arrays, copied */
    const int32_t* fb_lo;
    const int32_t* fb_len;
    const int32_t* fb_off;
    const float* fb_w;
    int32_t fb_nnz;
    const ofp_mlp* mlp;     /* classifier on the n_mels bands, or NULL; parameters are copied */
    int32_t want_rel;       /* != 0: the hop's relative envelope [B][C] is copied back too */
    /* Per-hop onset strength of the channel mean (realtime/recording.py:273-311, RecAnalysis.fft +
     * onset_strength): symmetric float32 Hann x audio[-n_fft:].mean(-1), rFFT, dB floored 80 dB below a tracked
     * maximum, positive flux against the previous frame averaged over the bins, normalised by a tracked
     * min / max, moving max / mean over the last max_length / avg_length entries (the reference reads these
     * two from config.MAX_LENGTH / AVG_LENGTH, which its config.py does not define).  The trackers are
     * loopmate.EMA_MinMaxTracker objects (:251-256); loopmate is absent, their update is ASSUMED to be
     * envelope_follower.c:27-57 with a single alpha: PARITY UNPINNED. */
    int32_t strength;       /* != 0: enabled */
    int32_t strength_ring;  /* entries of the onset-envelope ring (the reference's n_stft) */
    int32_t max_length, avg_length;
    float ls_max0, ls_minmax, ls_alpha;            /* EMA_MinMaxTracker(max0=10, minmax=0, alpha=0.0005) */
    float oe_min0, oe_minmin, oe_max0, oe_alpha;   /* EMA_MinMaxTracker(min0=0, minmin=0, max0=1, alpha=0.001) */
    int32_t tg_win_length;  /* > 0: the tempogram frame of the hop as well (realtime/recording.py:313-327, config.py:55:
                               TG_WIN_LENGTH): autocorrelation of the Hann-windowed last tg_win_length entries of the
                               normalised onset envelope, lags 0 .. tg_win_length - 1, divided by (its maximum + 1e-10);
                               needs strength, tg_win_length <= strength_ring.  PARITY UNPINNED like the envelope. */
} ofp_hop_config;
typedef struct ofp_hop_session ofp_hop_session; /* opaque */
int ofp_hop_create(ofp_detector* det, const ofp_hop_config* cfg, ofp_hop_session** out);
int ofp_hop_destroy(ofp_hop_session* s);
/* back to the state after creation (zero ring, fresh detector state, hop counter 0) */
int ofp_hop_reset(ofp_hop_session* s);
/* AmplitudeOnsetDetector.init_minmax_tracker (detection.py:827-840) over h_x [n_rows][C] */
int ofp_hop_warmup(ofp_hop_session* s, const float* h_x, int64_t n_rows);
/* One hop, h_hop [B][C].  Outputs (each may be NULL): *n_onsets; h_records [C] with .sample =
 * hop_index * B + delta (audio.py:65) and .clip = 0, in channel order; h_logits [C][mlp outputs];
 * h_mel [C][n_mels]; h_rel [B][C] (needs want_rel); h_strength [4 + tg_win_length] = {flux, normalised, moving max,
 * moving mean, then the tempogram frame} (needs strength). */
int ofp_hop_submit(ofp_hop_session* s, const float* h_hop);
int ofp_hop_collect(ofp_hop_session* s, int64_t* n_onsets, ofp_onset* h_records, float* h_logits, float* h_mel,
                    float* h_rel, float* h_strength);
int ofp_hop_push(ofp_hop_session* s, const float* h_hop, int64_t* n_onsets, ofp_onset* h_records, float* h_logits,
                 float* h_mel, float* h_rel, float* h_strength);
/* audio[-n_rows:] of the ring buffer (oldest row first), h_out [n_rows][C]; n_rows <= ring_samples */
int ofp_hop_ring_read(ofp_hop_session* s, int64_t n_rows, float* h_out);

/* ---- onset groups and their windows --------------------------------
 * find_onset_groups (detection.py:131-189) per clip, straight from the records
 * ofp_detect_offline wrote, without a host round trip.
 *   d_records [n_clips][cap_per_clip], d_counts [n_clips]   as ofp_detect_offline leaves them
 *   n_channels      row width (the reference uses max(channels)+1, detection.py:158,170);
 *                   every record's channel must be in [0, n_channels)
 *   max_distance, min_channels   detection.py:134-135
 *   close_channel   detection.py:136,185; < 0 for None
 *   d_groups  [n_clips][cap_groups][n_channels] int64 rows, -1 where a channel has no onset,
 *             in the reference's order; d_n_groups [n_clips] kept groups per clip (may exceed
 *             cap_groups: only cap_groups rows are stored)
 *   d_ws      ofp_group_workspace_bytes(n_clips, cap_per_clip) bytes */
int64_t ofp_group_workspace_bytes(int64_t n_clips, int64_t cap_per_clip);
int ofp_group_onsets(const ofp_onset* d_records, int64_t cap_per_clip, const int64_t* d_counts, int64_t n_clips,
                     int32_t n_channels, int64_t max_distance, int32_t min_channels, int32_t close_channel,
                     int64_t* d_groups, int64_t cap_groups, int64_t* d_n_groups, void* d_ws, int64_t ws_bytes,
                     void* stream);
/* FrameExtractor.__call__ (data.py:90-120, max_shift = 0) over those rows: window c of a
 * group starts at min_c(row) - pre_samples (use_min_onset != 0) or row[c] - pre_samples.
 *   d_x [n_clips][n_samples][n_channels] interleaved
 *   d_offsets [n_clips + 1] (out): first output row of each clip; [n_clips] = total rows
 *   d_out [cap_total][n_channels][width]: rows of all clips back to back (rows beyond
 *   cap_total are dropped).  Samples outside the clip read as 0. */
int ofp_group_windows(const float* d_x, int64_t n_clips, int64_t n_samples, int32_t n_channels,
                      const int64_t* d_groups, int64_t cap_groups, const int64_t* d_n_groups, int32_t pre_samples,
                      int32_t use_min_onset, int32_t width, float* d_out, int64_t cap_total, int64_t* d_offsets,
                      void* stream);

/* ---- cross-correlation lag and onset fixing --------------------------
 * cross_correlation_lag (detection.py:195-268) for a batch of pairs.
 *   d_x, d_y [n_pairs][n_in] float32; d: difference order applied first (np.diff, :238-239);
 *   take_abs (:240-242); cutoff = normalization_cutoff (:247-250).  With n = n_in - d,
 *   d_lo/d_hi [n_pairs] give the slice [lo, hi) of the normalised full correlation (length
 *   2n-1) that is searched, i.e. what Python's slicing at :257 / :263 selects.
 *   d_argmax [n_pairs]: np.argmax over the slice (first maximum), -1 for an empty slice; the
 *   lag is max_adjust - argmax (:268).  d_cc (optional) [n_pairs][cc_stride]: the slice values.
 * Dot products are accumulated in fp64 and rounded once to fp32 (see csrc/ofp_xcorr.hip).
 * n_in <= 4096, d <= 4. */
int ofp_xcorr_lag(const float* d_x, const float* d_y, int64_t n_pairs, int32_t n_in, int32_t d, int32_t take_abs,
                  int32_t cutoff, const int32_t* d_lo, const int32_t* d_hi, int32_t* d_argmax, float* d_cc,
                  int32_t cc_stride, void* stream);
/* adjust_onset (detection.py:299-352) for a batch of pairs, one wave each: d_x, d_y [n_pairs][n] float32,
 * d_onsets [n_pairs][2] (onset in x, onset in y), d_new_lag [n_pairs] -> d_moves [n_pairs][2], the amounts
 * the reference returns to be added to the two onsets.  Weighted sums in fp64 as in ofp_fix_onsets. */
int ofp_adjust_onset(const float* d_x, const float* d_y, int64_t n_pairs, int32_t n, const int32_t* d_onsets,
                     const int32_t* d_new_lag, int32_t* d_moves, void* stream);
/* filter_data (detection.py:355-370): d_y[t][c] = d_x[t][c], or 0 where the first difference along time is
 * negative (direction 1, "up") / positive (direction 2, "down"); row 0 is kept.  Not in place. */
int ofp_filter_direction(const float* d_x, int64_t n, int32_t n_channels, int32_t direction, float* d_y, void* stream);
/* detect_onset_region (detection.py:454-484) for n_onsets onsets of one 1-D signal d_audio [n_audio]:
 * region = audio[onset - n/2 : onset + n/2] (clipped), |.|, scipy medfilt (zero-padded, odd size <= 33),
 * threshold_factor * max, binary_opening with ones(5), first True -> d_out [n_onsets] absolute indices.
 * n/2*2 <= 4096. */
int ofp_onset_region(const float* d_audio, int64_t n_audio, const int64_t* d_onsets, int64_t n_onsets, int32_t n,
                     int32_t median_filter_size, float threshold_factor, int64_t* d_out, void* stream);
/* StretchFrameExtractor's resampling (data.py:212-222): scipy.signal.resample (Fourier method, real input) of
 * the windows audio[d_start[i] : d_start[i] + d_nx[i], c] to `num` samples each: d_out [n_items][C][num].
 * d_audio [n_samples][C]; samples outside the clip read as 0; d_nx[i] <= max_nx <= 2048, num <= 2048. */
int ofp_resample_windows(const float* d_audio, int64_t n_samples, int32_t n_channels, const int64_t* d_start,
                         const int32_t* d_nx, int64_t n_items, int32_t max_nx, int32_t num, float* d_out, void* stream);
/* Full cross-correlation of row pairs (batch_cc, data.py:226-230; paired_xcorr, model.py:12-45):
 * d_out[i][j] = sum_t a[i][t + j - (L-1)] b[i][t], j in [0, 2L-1); rows i of d_a / d_b start at
 * i*a_stride / i*b_stride floats.  mean_k > 1: output row i is the mean of input rows
 * [i*mean_k, (i+1)*mean_k) (paired_xcorr's mean over feature maps).  length <= 4096. */
int ofp_xcorr_full(const float* d_a, const float* d_b, int64_t n_rows, int32_t length, int64_t a_stride,
                   int64_t b_stride, int32_t mean_k, float* d_out, void* stream);
/* fix_onsets (detection.py:373-451) for every onset group, one workgroup per group:
 * median filter (scipy.ndimage 'reflect') over audio[a-look : b+look], d-th difference,
 * rectification by direction (0 none, 1 "up", 2 "down"), abs, then for every channel after the
 * earliest one cross_correlation_lag + adjust_onset (:299-352), in the reference's order.
 *   d_audio [n_clips][n_samples][n_channels]; d_onsets [n_clips][cap_groups][n_channels] int64
 *   (the layout ofp_group_onsets writes), updated in place (shift is added first, :414);
 *   d_n_groups [n_clips] rows in use per clip (clamped to cap_groups) or NULL for all rows;
 *   max_section: longest b - a + 2*(cutoff + tol) the work space holds (<= 4096);
 *   d_status [n_clips][cap_groups]: 0 fixed, 1 = group outside the clip, with a missing
 *   channel (-1) or longer than max_section (only shifted), 2 = row not in use.
 *   d_ws: ofp_fix_onsets_workspace_bytes(n_clips * cap_groups, ...) bytes. */
int64_t ofp_fix_onsets_workspace_bytes(int64_t n_groups, int32_t n_channels, int32_t max_section);
int ofp_fix_onsets(const float* d_audio, int64_t n_clips, int64_t n_samples, int32_t n_channels, int64_t* d_onsets,
                   int64_t cap_groups, const int64_t* d_n_groups, int32_t filter_size, int32_t d, int32_t direction, int32_t take_abs, int32_t zero_left,
                   int32_t cutoff, int32_t tol, int32_t shift, int32_t max_section, int32_t* d_status, void* d_ws,
                   int64_t ws_bytes, void* stream);

/* ---- spectral-flux onset detector (detection.py:89-128) --------------
 * The STFT is ofp_stft_power on the centre-padded signal (librosa.stft, center=True).
 *   ofp_spectral_flux: d_power [n_frames][n_bins] -> d_oe [n_frames-1] =
 *     mean_k max(0, w_k sqrt(P[t+1][k]) - w_k sqrt(P[t][k]))   (detection.py:106-110; d_weight is
 *     the normalised A-weighting of :105-106)
 *   ofp_select_rank: the value of ascending rank `rank` among n non-negative floats (the order
 *     statistics np.percentile interpolates, :111); d_out [1]
 *   ofp_scale_inverse: d_x[i] /= d_scale[0]
 *   ofp_peak_pick: librosa.util.peak_pick (:113-121): i is a peak iff x[i] == max(x[i-pre_max :
 *     i+post_max]), x[i] >= mean(x[i-pre_avg : i+post_avg]) + delta and i - previous peak > wait;
 *     d_peaks [cap] int64 indices, d_count [1] (may exceed cap), d_flags [n] scratch.
 * librosa is not available to pin these against: restated from its published definition. */
int ofp_spectral_flux(const float* d_power, int64_t n_frames, int32_t n_bins, const float* d_weight, float* d_oe,
                      void* stream);
int ofp_select_rank(const float* d_v, int64_t n, int64_t rank, float* d_out, void* stream);
int ofp_scale_inverse(float* d_x, int64_t n, const float* d_scale, void* stream);
int ofp_peak_pick(const float* d_x, int64_t n, int32_t pre_max, int32_t post_max, int32_t pre_avg, int32_t post_avg,
                  float delta, int64_t wait, int64_t* d_peaks, int64_t cap, int64_t* d_count, uint8_t* d_flags,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ONSETFP_H */
