"""CPU oracle, TEST INFRASTRUCTURE ONLY: onset grouping and group windows.

Restates ``find_onset_groups`` (reference detection.py:131-189).  Parity: pinned bit-for-bit
by tests/golden/g5_groups.npz (captured from the reference, 39 cases).
"""
import numpy as np


def find_onset_groups(onsets, channels, max_distance=1000, min_channels=3, close_channel=None, width=None):
    """detection.py:156-189.  Greedy scan: a group is anchored at its first onset and takes every
    following onset within ``max_distance`` of the anchor (detection.py:165); the first onset
    outside closes it and becomes the next anchor (:174).  A closed group is kept when it holds
    at least ``min_channels`` distinct channels (:168-169); its row is filled with -1 and then
    written in list order, so the LAST onset of a channel wins (:170-172).  ``close_channel``
    keeps rows whose entry for that channel is <= every entry, the -1 ones included (:185).
    Returns int64 [G, width] or None (:186-189); width = max(channels) + 1 (:158, 170)."""
    onsets = [int(s) for s in onsets]
    channels = [int(c) for c in channels]
    W = max(channels) + 1 if width is None else width
    rows, cur = [], []

    def close():
        if len({c for _, c in cur}) >= min_channels:
            r = np.full(W, -1, np.int64)
            for s, c in cur:
                r[c] = s
            rows.append(r)

    for s, c in zip(onsets, channels):
        if cur and abs(s - cur[0][0]) > max_distance:
            close()
            cur = []
        cur.append((s, c))
    close()
    if close_channel is not None:
        rows = [r for r in rows if np.all(r[close_channel] <= r)]
    return np.array(rows, dtype=np.int64) if rows else None


def group_windows(audio, groups, frame_length, pre_samples, use_min_onset=True):
    """FrameExtractor on group rows (data.py:90-120 with max_shift = 0): window c of group g
    starts at min_c(groups[g]) - pre_samples (use_min_onset) or groups[g, c] - pre_samples.
    Samples outside the clip read as 0 (the device form's documented guard; the reference's
    strided view would raise or wrap there)."""
    audio = np.asarray(audio, np.float32)
    N, C = audio.shape
    G = len(groups)
    out = np.zeros((G, C, frame_length), np.float32)
    for g in range(G):
        for c in range(C):
            st = int(groups[g].min() if use_min_onset else groups[g, c]) - pre_samples
            lo, hi = max(st, 0), min(st + frame_length, N)
            if hi > lo:
                out[g, c, lo - st:hi - st] = audio[lo:hi, c]
    return out
