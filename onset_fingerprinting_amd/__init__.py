"""onset_fingerprinting_amd -- the onset-fingerprinting hot path on MI355X (gfx950).

Drop-in surface of the reference package for the accelerated path:
``detection`` (amplitude onset detector), ``data`` (framing / STFT / MFCC),
``calibration.FCNN`` and ``model.CNN`` forward; plus ``pipeline`` (batched,
HBM-resident detect -> FFT -> fingerprint -> classify) and ``distributed``
(clip sharding + RCCL all-gather of onset records).

Everything runs in hand-written HIP kernels behind the C ABI of
``libonsetfp.so`` (include/onsetfp.h).  There is no CPU fallback.
"""
__all__ = ["detection", "data", "calibration", "model", "pipeline", "distributed", "synth"]
