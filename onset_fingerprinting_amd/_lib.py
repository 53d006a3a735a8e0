"""ctypes binding of libonsetfp.so (include/onsetfp.h).

The shared library is built in-tree by ``csrc/Makefile`` (hipcc, gfx950).  There
is NO CPU fallback: if the library is missing, or a call fails, an exception is
raised.  PyTorch is used only to own device memory and streams; every pointer
handed to the library is a raw device address.
"""
import ctypes
import subprocess
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libonsetfp.so"


class OnsetFPError(RuntimeError):
    pass


class DetectorParams(ctypes.Structure):
    _fields_ = [
        ("n_channels", ctypes.c_int32),
        ("block_size", ctypes.c_int32),
        ("floor_db", ctypes.c_float),
        ("hp_enabled", ctypes.c_int32),
        ("hp_b", ctypes.c_float * 5),
        ("hp_a", ctypes.c_float * 5),
        ("fast_attack", ctypes.c_float),
        ("fast_release", ctypes.c_float),
        ("slow_attack", ctypes.c_float),
        ("slow_release", ctypes.c_float),
        ("alpha_min", ctypes.c_float),
        ("alpha_max", ctypes.c_float),
        ("minmin", ctypes.c_float),
        ("min0", ctypes.c_float),
        ("max0", ctypes.c_float),
        ("manual", ctypes.c_int32),
        ("cooldown", ctypes.c_int64),
        ("backtrack", ctypes.c_int32),
        ("backtrack_buffer_size", ctypes.c_int64),
        ("backtrack_alpha", ctypes.c_float),
        ("backtrack_tol", ctypes.c_float),
    ]


class DetectTuning(ctypes.Structure):
    _fields_ = [
        ("hp_chunk", ctypes.c_int64), ("hp_warm", ctypes.c_int64),
        ("ar_chunk", ctypes.c_int64), ("ar_warm", ctypes.c_int64),
        ("mm_chunk", ctypes.c_int64), ("mm_warm", ctypes.c_int64),
        ("max_passes", ctypes.c_int32),
        ("ar_coarse_warm", ctypes.c_int64),
        ("hp_candidates", ctypes.c_int64),
        ("hp_candidate_offset", ctypes.c_int64),
        ("ar_guess", ctypes.c_int64),
        ("hp_span", ctypes.c_int64),
        ("ar_span", ctypes.c_int64),
        ("mm_span", ctypes.c_int64),
        ("verify_group", ctypes.c_int64),
        ("hp_dedupe", ctypes.c_int64),
        ("hp_early", ctypes.c_int64),
        ("lane_merge", ctypes.c_int64),
        ("fuse_db_sums", ctypes.c_int64),
        ("sm_segments", ctypes.c_int64),
        ("concurrent_calls", ctypes.c_int64),
        ("scan_skip", ctypes.c_int64),
        ("host_verify", ctypes.c_int64),
        ("interleaved", ctypes.c_int64),
        ("walk_through", ctypes.c_int64),
        ("line_stores", ctypes.c_int64),
    ]


class HopConfig(ctypes.Structure):
    _fields_ = [
        ("n_fft", ctypes.c_int32),
        ("ring_samples", ctypes.c_int64),
        ("n_mels", ctypes.c_int32),
        ("fb_lo", ctypes.c_void_p),
        ("fb_len", ctypes.c_void_p),
        ("fb_off", ctypes.c_void_p),
        ("fb_w", ctypes.c_void_p),
        ("fb_nnz", ctypes.c_int32),
        ("mlp", ctypes.c_void_p),
        ("want_rel", ctypes.c_int32),
        ("strength", ctypes.c_int32),
        ("strength_ring", ctypes.c_int32),
        ("max_length", ctypes.c_int32),
        ("avg_length", ctypes.c_int32),
        ("ls_max0", ctypes.c_float), ("ls_minmax", ctypes.c_float), ("ls_alpha", ctypes.c_float),
        ("oe_min0", ctypes.c_float), ("oe_minmin", ctypes.c_float), ("oe_max0", ctypes.c_float),
        ("oe_alpha", ctypes.c_float),
        ("tg_win_length", ctypes.c_int32),
    ]


_vp, _i32, _i64, _f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
_f32p_h = ctypes.POINTER(ctypes.c_float)
_long_p = ctypes.POINTER(ctypes.c_long)

# name -> (restype, argtypes); every name here must be declared in include/onsetfp.h
SIGNATURES = {
    "ofp_abi_version": (ctypes.c_int, []),
    "ofp_last_error": (ctypes.c_char_p, []),
    "ofp_device_count": (ctypes.c_int, []),
    "ofp_device_check": (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_int]),
    "ar_envelope": (None, [_f32p_h, _f32p_h, _f32, _f32, ctypes.c_int, ctypes.c_int]),
    "minmax_envelope": (None, [_f32p_h, _f32p_h, _f32p_h, _f32, _f32, _f32, ctypes.c_int, ctypes.c_int]),
    "backtrack_onsets": (None, [_f32p_h, _long_p, _long_p, _f32, _f32, ctypes.c_long, ctypes.c_long,
                                ctypes.c_long, ctypes.c_long]),
    "ofp_lfilter": (ctypes.c_int, [_f32p_h, _f32p_h, _f32p_h, _f32p_h, ctypes.c_int, _f32p_h, ctypes.c_long,
                                   ctypes.c_int]),
    "ofp_detector_create": (ctypes.c_int, [ctypes.POINTER(DetectorParams), ctypes.POINTER(ctypes.c_double),
                                           ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_vp)]),
    "ofp_detector_destroy": (ctypes.c_int, [_vp]),
    "ofp_detector_set_tuning": (ctypes.c_int, [_vp, ctypes.POINTER(DetectTuning)]),
    "ofp_pack_records": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, ctypes.c_int32, _vp, _vp]),
    "ofp_detect_workspace_bytes": (_i64, [_vp, _i64, _i64, _i64]),
    "ofp_detect_offline": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _i64,
                                          ctypes.POINTER(_i64), _vp]),
    "ofp_detect_offline_enqueue": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp]),
    "ofp_detect_offline_complete": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _i64,
                                                   ctypes.POINTER(_i64), _vp]),
    "ofp_detect_offline_finish_enqueue": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp]),
    "ofp_detect_offline_begin": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp]),
    "ofp_detect_offline_finish": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _i64,
                                                 ctypes.POINTER(_i64), _vp]),
    "ofp_detect_offline_begin_input": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp]),
    "ofp_detect_offline_begin_iir": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp]),
    "ofp_stream_state_bytes": (_i64, [_vp]),
    "ofp_stream_state_init": (ctypes.c_int, [_vp, _vp, _vp]),
    "ofp_stream_process": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i64, _i32, _i64, _vp, _vp, _i64, _vp, _vp]),
    "ofp_stream_calibrate": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    "ofp_detector_set_thresholds": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "ofp_stft_power": (ctypes.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp]),
    "ofp_stft_power_mel": (ctypes.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp,
                                          _i64, _vp]),
    "ofp_detect_planar_input": (_vp, [_vp, _i64, _i64, _i64, _vp]),
    "ofp_detect_planar_stride": (_i64, [_vp, _i64, _i64, _i64]),
    "ofp_stft_frames": (ctypes.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32,
                                       _vp, _vp, _vp]),
    "ofp_extract_frames": (ctypes.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp]),
    "ofp_mel": (ctypes.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ofp_mfcc": (ctypes.c_int, [_vp, _i64, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _vp]),
    "ofp_dense": (ctypes.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "ofp_mlp_create": (ctypes.c_int, [_i32, ctypes.POINTER(_i32), ctypes.POINTER(_i32), ctypes.POINTER(_vp),
                                      ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                      ctypes.POINTER(_vp)]),
    "ofp_mlp_destroy": (ctypes.c_int, [_vp]),
    "ofp_mlp_lds_bytes": (_i64, [_vp]),
    "ofp_mlp_forward": (ctypes.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "ofp_stft_power_mel_mlp": (ctypes.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32,
                                              _vp, _i64, _vp, _vp, _vp]),
    "ofp_hop_create": (ctypes.c_int, [_vp, ctypes.POINTER(HopConfig), ctypes.POINTER(_vp)]),
    "ofp_hop_destroy": (ctypes.c_int, [_vp]),
    "ofp_hop_reset": (ctypes.c_int, [_vp]),
    "ofp_hop_warmup": (ctypes.c_int, [_vp, _vp, _i64]),
    "ofp_hop_submit": (ctypes.c_int, [_vp, _vp]),
    "ofp_hop_collect": (ctypes.c_int, [_vp, ctypes.POINTER(_i64), _vp, _vp, _vp, _vp, _vp]),
    "ofp_hop_push": (ctypes.c_int, [_vp, _vp, ctypes.POINTER(_i64), _vp, _vp, _vp, _vp, _vp]),
    "ofp_hop_ring_read": (ctypes.c_int, [_vp, _i64, _vp]),
    "ofp_autocorr_softmax": (ctypes.c_int, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "ofp_conv1d": (ctypes.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp,
                                  _i32, _vp, _vp]),
    "ofp_groupnorm1": (ctypes.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _f32, _i32, _vp, _vp]),
    "ofp_group_workspace_bytes": (_i64, [_i64, _i64]),
    "ofp_group_onsets": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i32, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _i64,
                                        _vp]),
    "ofp_group_windows": (ctypes.c_int, [_vp, _i64, _i64, _i32, _vp, _i64, _vp, _i32, _i32, _i32, _vp, _i64, _vp,
                                         _vp]),
    "ofp_xcorr_lag": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp]),
    "ofp_adjust_onset": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "ofp_filter_direction": (ctypes.c_int, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "ofp_onset_region": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _f32, _vp, _vp]),
    "ofp_resample_windows": (ctypes.c_int, [_vp, _i64, _i32, _vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "ofp_xcorr_full": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i64, _i64, _i32, _vp, _vp]),
    "ofp_spectral_flux": (ctypes.c_int, [_vp, _i64, _i32, _vp, _vp, _vp]),
    "ofp_select_rank": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp]),
    "ofp_scale_inverse": (ctypes.c_int, [_vp, _i64, _vp, _vp]),
    "ofp_peak_pick": (ctypes.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _f32, _i64, _vp, _i64, _vp, _vp, _vp]),
    "ofp_fix_onsets_workspace_bytes": (_i64, [_i64, _i32, _i32]),
    "ofp_fix_onsets": (ctypes.c_int, [_vp, _i64, _i64, _i32, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                      _i32, _i32, _vp, _vp, _i64, _vp]),
}

_lib = None


def build(force=False):
    """Compile csrc/*.hip into libonsetfp.so for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", str(_HERE / "csrc"), "-j4"]
    if force:
        subprocess.check_call(cmd + ["clean"])
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    """The loaded library with argtypes set; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise OnsetFPError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)")
        L = ctypes.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here means header and library disagree
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status, what=""):
    if status != 0:
        msg = lib().ofp_last_error().decode(errors="replace")
        raise OnsetFPError(f"{what or 'libonsetfp'} failed with status {status}: {msg}")


def last_error():
    return lib().ofp_last_error().decode(errors="replace")


_checked = {}


def require_gpu(device=0):
    """Fail loudly unless `device` is a gfx950 GPU (checked once per device and process)."""
    if device in _checked:
        return _checked[device]
    L = lib()
    n = L.ofp_device_count()
    if n <= 0:
        raise OnsetFPError(f"no GPU visible to HIP ({last_error()}); onset_fingerprinting_amd has no CPU path")
    buf = ctypes.create_string_buffer(64)
    check(L.ofp_device_check(device, buf, 64), "ofp_device_check")
    _checked[device] = buf.value.decode()
    return _checked[device]
