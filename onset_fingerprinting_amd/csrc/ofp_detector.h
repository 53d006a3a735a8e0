// The opaque detector handle of include/onsetfp.h (host side).
#pragma once
#include <vector>

#include "ofp_common.h"

// what a call that was only enqueued (ofp_detect_offline_enqueue) leaves for its completion
struct ofp_detect_pending {
    bool valid = false;        // an enqueued call awaits ofp_detect_offline_complete
    bool empty = false;        // fewer samples than one block: nothing ran
    bool timed = false;        // the stage events were recorded (not while capturing a graph)
    bool hp_timed = false;
    bool staged = false;       // IIR candidates in stages: the run counts arrive with the final copy
    bool sm_flag = false;      // the segmented state machine's last change counter decides about the fall-back
    bool ahead = false;        // follower / tracker passes enqueued ahead: their change counters arrive with the final copy
    int ar_nv = 0, mm_nv = 0;  // verifying passes enqueued ahead (0: single chunk / manual thresholds / host-verified)
    int hp_rounds = 0;         // IIR verification rounds enqueued ahead (0: no high-pass)
    int n_cuts = 0;
    int64_t cuts[16] = {};
    int64_t info[16] = {};
};

struct ofp_detector {
    ofp_detector_params p;
    float b[5], a[5];  // normalised by a[0] in fp32, as scipy's lfilter does
    float ialpha_min, ialpha_max;
    std::vector<double> on, off;
    float* d_on_f = nullptr;   // [C] on threshold (manual) or factor (relative), fp32
    float* d_off_f = nullptr;  // [C]
    double* d_on_d = nullptr;  // [C] the Python double, used for row 0 in manual mode
    ofp_detect_tuning t;
    hipEvent_t ev[10] = {};     // stage timing (ofp_detect_offline h_info)
    int n_cus = 256;            // compute units of the device the detector was created on
    ofp_detect_pending pend;
    int ar_pass_hint = 3, mm_pass_hint = 3;  // likewise for the verifying passes of the follower / tracker stage (+1)
    int hp_rounds_hint = 5;     // IIR verification rounds the next call enqueues ahead: what recent calls needed, + 2
    int* h_flags = nullptr;     // pinned host words the pass loops copy their counters into (truly async D2H)
};
