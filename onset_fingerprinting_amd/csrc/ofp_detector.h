// The opaque detector handle of include/onsetfp.h (host side).
#pragma once
#include <vector>

#include "ofp_common.h"

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
    int* h_flags = nullptr;     // pinned host words the pass loops copy their counters into (truly async D2H)
};
