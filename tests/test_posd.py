"""POSD session files: round trips, the reference draft's own example
document (notebooks/dataset_spec_draft.org:333-397, re-typed here as data), PCM decoding.
No reference fixture exists (no recordings ship upstream): the spec example is the pin."""
import json
import struct
import wave

import numpy as np
import pytest

from onset_fingerprinting_amd import posd


def test_round_trip_multichannel_and_per_channel(tmp_path):
    rng = np.random.default_rng(0)
    audio = (0.3 * rng.standard_normal((5000, 3))).astype(np.float32)
    onsets = np.array([[100, 110, 95], [2000, -1, 2010]], np.int64)
    ch = {"SP": {"location": [0.95, 0], "coordinate_system": "polar"},
          "OP": {"location": [0.95, 30], "coordinate_system": "polar"}, "C": {}}
    hits = [dict(zone="center", location=[0.1, 0.2], velocity=0.5), dict(zone="edge", location=[0.9, 1.0], velocity=1.0)]
    posd.write_session(tmp_path, "s1", audio, 48000, onsets, ch, meta=dict(instrument="snare"), hits=hits)
    posd.write_session(tmp_path / "sub", "s2", audio, 96000, onsets, ["a", "b", "c"], per_channel_files=True)
    (tmp_path / "instruments.json").write_text(json.dumps({"snare": {"zones": ["center", "edge"]}}))
    s1 = posd.read_session(tmp_path, "s1")
    assert s1["sr"] == 48000 and np.array_equal(s1["audio"].view(np.uint32), audio.view(np.uint32))
    assert np.array_equal(s1["onsets"], onsets) and np.allclose(s1["locations"], [[0.1, 0.2], [0.9, 1.0]])
    assert s1["meta"]["instrument"] == "snare" and list(s1["meta"]["channels"]) == ["SP", "OP", "C"]
    assert s1["hits"][1]["i"] == 1 and s1["hits"][1]["onset_start"] == [2000, -1, 2010]
    s2 = posd.read_session(tmp_path / "sub", "s2")
    assert s2["sr"] == 96000 and np.array_equal(s2["audio"], audio) and s2["locations"] is None
    assert np.array_equal(posd.read_session(tmp_path / "sub", "s2", channels=["c", "a"])["audio"], audio[:, [2, 0]])
    found = posd.find_sessions(tmp_path)  # instruments.json has no "meta": skipped (data.py:393-396)
    assert [f.name for f, _ in found] == ["s1.json", "s2.json"]
    # what the reference's MCPOSD.from_file extracts (data.py:298-301)
    doc = json.loads((tmp_path / "s1.json").read_text())
    assert np.array_equal(np.array([x["onset_start"] for x in doc["hits"]]), onsets)


def test_reads_the_spec_example_and_pcm(tmp_path):
    doc = {"meta": {"channels": {"SP": {"location": [0.95, 0], "coordinate_system": "polar"},
                                 "OP": {"location": [0.95, 30], "coordinate_system": "polar"}},
                    "instrument": "snare", "tuning": "low"},
           "hits": [{"i": 0, "zone": "center", "onset_start": [0, 2], "velocity": 0.0, "isolated": True,
                     "pitch": 220, "conditions": {"wires": "on"}},
                    {"i": 1, "zone": "edge", "onset_start": [48000, 47900], "velocity": 1.0, "isolated": True,
                     "pitch": 219, "conditions": {"wires": "off"}}]}
    (tmp_path / "session1.json").write_text(json.dumps(doc))
    pcm = (np.arange(2000) % 200 - 100).astype("<i2") * 300
    for n in ("SP", "OP"):  # 16-bit PCM files as a recorder would write them
        with wave.open(str(tmp_path / f"session1_{n}.wav"), "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(44100)
            w.writeframes(pcm.tobytes())
    s = posd.read_session(tmp_path, "session1")
    assert s["sr"] == 44100 and s["audio"].shape == (2000, 2) and s["onsets"].tolist() == [[0, 2], [48000, 47900]]
    assert np.array_equal(s["audio"][:, 0], pcm.astype(np.float32) / 32768.0)
    # 24-bit and extensible headers
    v = np.array([0, 1, -1, 8388607, -8388608], np.int32)
    raw = b"".join(struct.pack("<i", int(x))[:3] for x in v)
    fmt = struct.pack("<HHIIHH", 0xFFFE, 1, 8000, 24000, 3, 24) + struct.pack("<HHI", 22, 24, 4) + \
        struct.pack("<H", 1) + b"\x00" * 14
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(raw)) + raw + b"\x00"
    (tmp_path / "x.wav").write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)
    a, sr = posd.read_wav(tmp_path / "x.wav")
    assert sr == 8000 and np.array_equal(a, (v / 8388608.0).astype(np.float32))
    with pytest.raises(ValueError):
        posd.write_session(tmp_path, "bad", np.zeros((10, 2), np.float32), 48000, np.zeros((0, 2)), ["only_one"])


@pytest.mark.gpu
def test_detect_group_fix_to_session(tmp_path):
    """The whole post-detection chain ends in a session file the reference's loader reads."""
    from onset_fingerprinting_amd import detection, synth
    audio, true_on = synth.sensor_hits(31, n_channels=3, n=60000, hits=10)
    audio = (audio * 0.5).astype(np.float32)
    ch, on, _ = detection.detect_onsets_amplitude(audio, block_size=128, sr=48000)
    groups = detection.find_onset_groups(on, ch, max_distance=400, min_channels=3)
    assert groups is not None and len(groups) >= 8
    fixed = detection.fix_onsets(audio, groups, d=1, take_abs=True)
    posd.session_from_groups(tmp_path, "cal", audio, 48000, fixed, ["a", "b", "c"], meta=dict(instrument="pad"))
    s = posd.read_session(tmp_path, "cal")
    assert np.array_equal(s["onsets"], fixed) and np.array_equal(s["audio"], audio)
