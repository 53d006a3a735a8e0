"""CPU oracle for the onset-fingerprinting hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product (``onset_fingerprinting_amd``) never
does.  It restates the reference's algorithm on the CPU (plain C for the
detector, numpy for framing / STFT / mel / classifier forward) and is pinned
against golden vectors captured from the reference itself
(``tests/golden/make_golden.py``).  Parity status per function is in each
docstring; reference paths are under ``/root/reference/onset_fingerprinting/``.
"""
from .detector import (  # noqa: F401
    OracleDetector,
    init_ranges,
    ar_envelope,
    backtrack_onsets,
    butter_hp_f32,
    detect_onsets_amplitude,
    exp10f,
    host_math_probe,
    lfilter4,
    lib,
    log10f,
    minmax_envelope,
    rect_db,
    rel_linear,
)
from .spectral import (  # noqa: F401
    cspec_to_mfcc,
    dense_power_frames,
    fourier_resample,
    frame_extract,
    hann_periodic,
    mel_filterbank,
    power_to_db,
    stft,
    stft_frame,
    stretch_frames,
    window_contribution_weights,
)
from .classifier import cccnn_forward, cnn_forward, fcnn_forward  # noqa: F401
from .groups import find_onset_groups, group_windows  # noqa: F401
from .xcorr import (adjust_onset, adjust_onset_rel, cross_correlation_lag, detect_onset_region,  # noqa: F401
                    filter_data, fix_onsets, lag_window, xcorr_slice)
from .spectral_onsets import EMAMinMax, HopStrength, detect_onsets_spectral, librosa_stft_mag, peak_pick  # noqa: F401
