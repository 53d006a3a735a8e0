"""CPU sanitizer runs (the reference has none; the build compiles its host C with
-fsanitize=address,undefined in tests).

* the oracle's C restatement (oracle/ofp_oracle.c) as an ASan + UBSan build (UB is fatal) drives every exported
  function on the golden inputs: the whole oracle-vs-reference golden suite runs against it in a subprocess with the
  sanitizer runtime preloaded -- plus a negative control (a deliberately short buffer must be caught: the sanitizer
  is really armed);
* the host side of the three legacy symbols (csrc/ofp_core.hip: argument handling and staging of the caller's
  arrays) as a host-only ASan + UBSan build (hipcc -fsanitize=address,undefined -fno-gpu-sanitize); without a GPU
  every call must come back cleanly at its first device call, with any argument combination, touching nothing.
GPU AddressSanitizer is not available on this pool; the kernels are covered by the differential tests instead."""
import os
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]
BAD = ("AddressSanitizer", "runtime error:", "LeakSanitizer")


def _gcc_asan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if p and Path(p).exists() else None


def _env(preload, **extra):
    env = dict(os.environ, LD_PRELOAD=preload, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", PYTHONPATH=str(REPO))
    env.update(extra)
    return env


def test_oracle_c_restatement_under_asan_and_ubsan():
    asan = _gcc_asan()
    if asan is None:
        pytest.skip("gcc's libasan is not installed")
    so = REPO / "oracle" / "libofp_oracle_asan.so"
    subprocess.check_call(["make", "-C", str(REPO / "oracle"), str(so)])
    env = _env(asan, OFP_ORACLE_SO=str(so))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "tests/test_oracle_golden.py"],
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert " passed" in out and not any(b in out for b in BAD), out[-3000:]
    # negative control: the same library must trap a buffer that is 8 floats too short
    code = ("import numpy as np, oracle\n"
            "x = np.ones(64, np.float32); y = np.empty(64, np.float32)\n"
            "oracle.lib.oracle_rect_db(x, y, 72, -70.0)\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=REPO, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "AddressSanitizer" in r.stderr, (r.returncode, r.stderr[-1500:])


DRIVE_LEGACY = r'''
import ctypes, sys
import numpy as np
L = ctypes.CDLL(sys.argv[1])
f32p = ctypes.POINTER(ctypes.c_float); lp = ctypes.POINTER(ctypes.c_long)
L.ar_envelope.argtypes = [f32p, f32p, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int]
L.minmax_envelope.argtypes = [f32p, f32p, f32p, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int]
L.backtrack_onsets.argtypes = [f32p, lp, lp, ctypes.c_float, ctypes.c_float, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_long]
L.ofp_lfilter.argtypes = [f32p, f32p, f32p, f32p, ctypes.c_int, f32p, ctypes.c_long, ctypes.c_int]
L.ofp_last_error.restype = ctypes.c_char_p
P = lambda a: a.ctypes.data_as(f32p)
x = np.zeros((16, 4), np.float32); y = np.full((16, 4), -70, np.float32); y0 = y.copy()
for size, n in ((4, 16), (0, 16), (4, 0), (-1, 5), (1, 1)):
    L.ar_envelope(P(x), P(y), 0.3, 0.003, size, n)
mn = np.zeros(4, np.float32); mx = np.full(4, 10, np.float32)
for n, C in ((16, 4), (0, 4), (16, 0), (1, 1)):
    L.minmax_envelope(P(x), P(mn), P(mx), 1e-4, 1e-5, 2.0, n, C)
ch = np.zeros(3, np.int64); de = np.zeros(3, np.int64)
for k in (3, 0, -2):
    L.backtrack_onsets(P(x), ch.ctypes.data_as(lp), de.ctypes.data_as(lp), 0.3, 1e-6, 16, k, 4, 8)
b = np.array([1, -2, 1], np.float32); a = np.array([1, -1.8, 0.9], np.float32); zi = np.zeros((2, 4), np.float32)
codes = [L.ofp_lfilter(P(x), P(y), P(b), P(a), order, P(zi), n, C) for order, n, C in ((2, 16, 4), (9, 16, 4), (0, 0, 4), (2, 16, 0))]
bad = L.ofp_lfilter(None, P(y), P(b), P(a), 2, P(zi), 16, 4)
assert bad != 0 and codes[1] != 0 and codes[2] == 0 and codes[3] == 0, (bad, codes)
assert np.array_equal(y, y0) and (mn == 0).all() and (mx == 10).all() and (de == 0).all()   # no GPU: nothing was touched
print("abi", L.ofp_abi_version(), "devices", L.ofp_device_count(), "last error:", L.ofp_last_error().decode()[:80])
'''


def test_legacy_symbol_host_side_under_asan_without_a_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the calls would not stop at their first device call")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    rts = sorted(Path("/opt/rocm/lib/llvm/lib/clang").glob("*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not Path(hipcc).exists() or not rts:
        pytest.skip("hipcc or its ASan runtime is not installed")
    so = tmp_path / "libofp_core_hostasan.so"
    subprocess.check_call([hipcc, "-O1", "-g", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
                           "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-gpu-sanitize", "-shared",
                           str(REPO / "onset_fingerprinting_amd" / "csrc" / "ofp_core.hip"), "-o", str(so)])
    r = subprocess.run([sys.executable, "-c", DRIVE_LEGACY, str(so)], env=_env(str(rts[-1])), capture_output=True, text=True,
                       timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "abi 3" in r.stdout, out[-3000:]
    assert not any(b in out for b in BAD), out[-3000:]
