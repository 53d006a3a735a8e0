"""CPU-only: the C-ABI library loads and exports every symbol include/onsetfp.h
declares (no compute calls without a GPU), and fails loudly without one."""
import ctypes
import re
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]


def header_functions():
    src = (REPO / "include" / "onsetfp.h").read_text()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith("#"))
    return sorted(set(re.findall(r"\b([a-z_][a-z0-9_]*)\s*\(", src)))


def test_header_and_binding_agree():
    from onset_fingerprinting_amd import _lib
    names = header_functions()
    assert "ofp_detect_offline" in names and "ar_envelope" in names and len(names) >= 20
    assert sorted(_lib.SIGNATURES) == names, set(names) ^ set(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    from onset_fingerprinting_amd import _lib
    if not _lib.LIB_PATH.exists():
        _lib.build()
    L = ctypes.CDLL(str(_lib.LIB_PATH))
    for name in header_functions():
        assert hasattr(L, name), f"{name} declared in onsetfp.h but not exported"
    assert _lib.lib().ofp_abi_version() == 3


def test_no_gpu_means_loud_failure_not_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import numpy as np

    from onset_fingerprinting_amd import _lib, detection
    with pytest.raises(_lib.OnsetFPError):
        _lib.require_gpu(0)
    with pytest.raises(Exception):
        detection.detect_onsets_amplitude(np.zeros((1024, 2), np.float32))


def test_product_never_imports_the_oracle():
    for p in (REPO / "onset_fingerprinting_amd").rglob("*.py"):
        text = p.read_text()
        assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), p
    for p in (REPO / "onset_fingerprinting_amd" / "csrc").glob("*"):
        if p.suffix in (".hip", ".h"):
            assert "oracle/" not in p.read_text(), p


def test_struct_mirrors_follow_the_header_field_for_field():
    """The ctypes mirrors of the header's structs list the same fields in the same order (a field added on one side
    only shifts every later one silently)."""
    from onset_fingerprinting_amd import _lib
    src = (REPO / "include" / "onsetfp.h").read_text()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)

    def fields(struct):
        body = re.search(r"typedef struct " + struct + r"\s*\{(.*?)\}\s*" + struct + r"\s*;", src, flags=re.S).group(1)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            # "int64_t a, b" / "const float* fb_w" / "float x"
            first, *rest = decl.split(",")
            names.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", first)[-1])
            names += [re.findall(r"[A-Za-z_][A-Za-z0-9_]*", r)[-1] for r in rest]
        return names

    assert fields("ofp_detect_tuning") == [f[0] for f in _lib.DetectTuning._fields_]
    assert fields("ofp_hop_config") == [f[0] for f in _lib.HopConfig._fields_]
    assert fields("ofp_detector_params") == [f[0] for f in _lib.DetectorParams._fields_]
