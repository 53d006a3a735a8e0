"""Replays the differential sweep of tests/test_gpu_fuzz.py up to its first failing case and bisects it:
prints the full case, where `rel` differs, and which tuning overrides make the difference go away.

    python tools/fuzz_find.py <seed> [max_cases]"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402

import oracle  # noqa: E402  (a tool, like the tests: the oracle is the checker)
from onset_fingerprinting_amd import detection  # noqa: E402
from tests.test_gpu_fuzz import bits, random_case  # noqa: E402


def compare(x, kw, tuning):
    recs, rel, info = detection.detect_batch(x[None], tuning=tuning or None, **kw)
    c, o, orel = oracle.detect_onsets_amplitude(x, **kw)
    same_idx = np.array_equal(recs[0]["channel"], np.array(c, np.int64)) and np.array_equal(recs[0]["sample"], np.array(o, np.int64))
    d = bits(rel[0]) != bits(orel)
    return same_idx, d, rel[0], orel, info


def main():
    seed = int(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    rng = np.random.default_rng(seed)
    for case in range(n):
        x, kw, tuning = random_case(rng)
        same_idx, d, rel, orel, info = compare(x, kw, tuning)
        if same_idx and not d.any():
            continue
        rows = np.nonzero(d.any(axis=1))[0]
        print(json.dumps(dict(case=case, shape=x.shape, kw={k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()},
                              tuning=tuning, same_idx=bool(same_idx), n_diff=int(d.sum()), first_row=int(rows[0]) if len(rows) else -1,
                              last_row=int(rows[-1]) if len(rows) else -1, chans=np.nonzero(d.any(axis=0))[0].tolist(),
                              passes=[info["hp_passes"], info["ar_passes"], info["mm_passes"], info["repaired"]])))
        if len(rows):
            r = rows[0]
            print("first differing row", r, "gpu", rel[r].tolist(), "oracle", orel[r].tolist())
        for name, over in (("verify_group=1", dict(verify_group=1)), ("ar_span=1", dict(ar_span=1)), ("mm_span=1", dict(mm_span=1)),
                           ("hp_span=1", dict(hp_span=1)), ("no tuning", None)):
            t = dict(tuning, **over) if over is not None else {}
            s2, d2, *_ = compare(x, kw, t)
            print("  with", name, "->", "ok" if (s2 and not d2.any()) else f"still differs ({int(d2.sum())})")
        for rep in range(3):
            s2, d2, *_ = compare(x, kw, tuning)
            print("  repeat", rep, "->", "ok" if (s2 and not d2.any()) else f"differs ({int(d2.sum())})")
        return
    print("no failing case in", n)


if __name__ == "__main__":
    main()
