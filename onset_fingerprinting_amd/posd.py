"""POSD sessions on disk: ``<session>.json`` + ``<session>.wav`` (or one
``<session>_<channel>.wav`` per channel), the layout of the reference's dataset draft
(notebooks/dataset_spec_draft.org:240-268, 275-291, 333-397) that its loaders read
(data.py:285-311 ``MCPOSD.from_file``: hits[].onset_start + hits[].location; data.py:385-442
``POSD``: every ``*.json`` with a "meta" key, audio in ``<stem>_<channel>.wav``).

Host-side I/O only (no kernel: JSON and RIFF headers).  The reference reads audio with
``soundfile``, which is not installed; the small RIFF/WAVE reader below covers what it would be
given here: PCM 16/24/32-bit and IEEE float 32/64, plain or WAVE_FORMAT_EXTENSIBLE headers.
"""
import json
import struct
from pathlib import Path

import numpy as np


def write_wav(path, audio, sr):
    """audio [N] or [N, C] float32 -> IEEE-float32 RIFF/WAVE (what ``sf.read(dtype=float32)``
    returns unchanged)."""
    a = np.ascontiguousarray(audio, dtype="<f4")
    if a.ndim == 1:
        a = a[:, None]
    n, c = a.shape
    data = a.tobytes()
    fmt = struct.pack("<HHIIHH", 3, c, int(sr), int(sr) * c * 4, c * 4, 32)
    fact = struct.pack("<I", n)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"fact" + struct.pack("<I", 4) + fact
    body += b"data" + struct.pack("<I", len(data)) + data + (b"\x00" if len(data) % 2 else b"")
    Path(path).write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)


def read_wav(path):
    """-> (audio float32 [N] or [N, C], sample rate).  Integer PCM is scaled by 2^(bits-1), as
    libsndfile does for float reads."""
    b = Path(path).read_bytes()
    if b[:4] != b"RIFF" or b[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(b):
        cid, size = b[pos:pos + 4], struct.unpack("<I", b[pos + 4:pos + 8])[0]
        chunk = b[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag, ch, sr, _, _, bits = struct.unpack("<HHIIHH", chunk[:16])
            if tag == 0xFFFE and len(chunk) >= 26:  # WAVE_FORMAT_EXTENSIBLE: the sub-format GUID starts with the tag
                tag = struct.unpack("<H", chunk[24:26])[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            data = chunk
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    tag, ch, sr, bits = fmt
    if tag == 3 and bits in (32, 64):
        x = np.frombuffer(data, dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
    elif tag == 1 and bits == 16:
        x = np.frombuffer(data, dtype="<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 32:
        x = (np.frombuffer(data, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    elif tag == 1 and bits == 24:
        raw = np.frombuffer(data[:len(data) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = raw[:, 0] | (raw[:, 1] << 8) | (raw[:, 2] << 16)
        v = np.where(v & 0x800000, v - (1 << 24), v)
        x = (v / 8388608.0).astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported WAVE format tag {tag} with {bits} bits")
    x = x[:len(x) // ch * ch]
    return (x.reshape(-1, ch) if ch > 1 else x), sr


def write_session(folder, name, audio, sr, onsets, channels, meta=None, hits=None, per_channel_files=False):
    """Write one session.
      audio     [N, C] float32
      onsets    [G, C] int: hits[i].onset_start, one index per channel ("must not miss channels",
                spec :246-248; -1 is kept as the sentinel the spec suggests)
      channels  list of C names, or dict name -> {"location": ..., "coordinate_system": ...}
                (session meta "channels", spec :339-348)
      meta      further session metadata (instrument, ...)
      hits      optional list of G dicts merged into the hit records (zone, location, velocity, ...)
      per_channel_files  write <name>_<channel>.wav per channel (what data.py:400-402 opens)
                instead of one multi-channel <name>.wav (what data.py:297 opens)."""
    folder = Path(folder)
    folder.mkdir(parents=True, exist_ok=True)
    audio = np.asarray(audio, dtype=np.float32)
    if audio.ndim == 1:
        audio = audio[:, None]
    onsets = np.asarray(onsets, dtype=np.int64).reshape(-1, audio.shape[1])
    names = list(channels)
    if len(names) != audio.shape[1]:
        raise ValueError(f"{len(names)} channel names for {audio.shape[1]} audio channels")
    ch_meta = channels if isinstance(channels, dict) else {n: {} for n in names}
    doc = {"meta": dict(meta or {}, channels=ch_meta), "hits": []}
    for i, row in enumerate(onsets):
        h = {"i": i, "onset_start": [int(v) for v in row] if audio.shape[1] > 1 else int(row[0])}
        if hits is not None:
            h.update(hits[i])
        doc["hits"].append(h)
    (folder / f"{name}.json").write_text(json.dumps(doc, indent=4))
    if per_channel_files:
        for c, n in enumerate(names):
            write_wav(folder / f"{name}_{n}.wav", audio[:, c], sr)
    else:
        write_wav(folder / f"{name}.wav", audio, sr)
    return folder / f"{name}.json"


def read_session(folder, name, channels=None):
    """-> dict(audio [N, C] float32, sr, meta, hits (list of dicts), onsets int64 [G, C],
    locations (array or None)).  Reads <name>.wav, or the per-channel files of `channels`
    (default: every channel of the session meta)."""
    folder = Path(folder)
    doc = json.loads((folder / f"{name}.json").read_text())
    if "meta" not in doc:
        raise ValueError(f"{name}.json holds no session (no 'meta' key; data.py:393-396 skips such files)")
    multi = folder / f"{name}.wav"
    if multi.exists() and channels is None:
        audio, sr = read_wav(multi)
    else:
        cols = []
        for n in (channels or list(doc["meta"]["channels"])):
            a, sr = read_wav(folder / f"{name}_{n}.wav")
            cols.append(a)
        audio = np.stack(cols, axis=1)
    if audio.ndim == 1:
        audio = audio[:, None]
    hits = doc["hits"]
    onsets = np.array([np.atleast_1d(h["onset_start"]) for h in hits], dtype=np.int64).reshape(len(hits), -1)
    loc = np.array([h["location"] for h in hits]) if hits and all("location" in h for h in hits) else None
    return dict(audio=audio, sr=sr, meta=doc["meta"], hits=hits, onsets=onsets, locations=loc)


def find_sessions(path):
    """Every session below `path`: (json path, parsed document) for each *.json with a "meta"
    key, visiting sub-directories (spec :286-289; data.py:389-396)."""
    out = []
    for f in sorted(Path(path).rglob("*.json")):
        doc = json.loads(f.read_text())
        if isinstance(doc, dict) and "meta" in doc:
            out.append((f, doc))
    return out


def session_from_groups(folder, name, audio, sr, groups, channels, **kw):
    """Write the onset groups `find_onset_groups` / `fix_onsets` produced (int [G, C], -1 where a
    channel has no onset) as a session: the end of detect -> group -> fix on real recordings."""
    g = np.zeros((0, np.asarray(audio).shape[1]), np.int64) if groups is None else np.asarray(groups, np.int64)
    return write_session(folder, name, audio, sr, g, channels, **kw)
