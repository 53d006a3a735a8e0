"""CPU-only: the N > 1 path (clip sharding + all-gather of onset records) over
gloo with world_size 2."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pytest

from onset_fingerprinting_amd.distributed import (ONSET_DTYPE, all_gather_blocks, all_gather_onsets,
                                                  all_gather_onsets_padded, flatten_records, pack_block, pack_clips,
                                                  records_to_numpy, shard_range, unpack_gathered)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_records(rank):
    """Synthetic per-rank detector output: [n_clips, cap] structured records + counts."""
    rng = np.random.default_rng(100 + rank)
    n_clips, cap = 3, 8
    counts = np.array([2 + rank, 0, 5], dtype=np.int64)
    recs = np.zeros((n_clips, cap), dtype=ONSET_DTYPE)
    for c in range(n_clips):
        recs["clip"][c] = c
        recs["channel"][c] = rng.integers(0, 4, cap)
        recs["sample"][c] = np.sort(rng.integers(0, 480000, cap))
    return recs, counts


def _as_u8(recs):
    return torch.from_numpy(recs.view(np.uint8).reshape(recs.shape[0], recs.shape[1], 16).copy())


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    recs, counts = _make_records(rank)
    lo, hi = shard_range(6, rank, world)
    flat = flatten_records(_as_u8(recs), torch.from_numpy(counts), 8, clip_offset=lo)
    out = all_gather_onsets(flat)
    # the single-collective form bench.py uses gives the same records
    padded = unpack_gathered(all_gather_onsets_padded(flat, 16))
    assert torch.equal(out, padded)
    try:
        unpack_gathered(all_gather_onsets_padded(flat, 3))  # rank 1 holds 8 records: truncated
        truncated_detected = False
    except RuntimeError:
        truncated_detected = True
    assert truncated_detected
    q.put((rank, [tuple(int(v) for v in r) for r in records_to_numpy(out).tolist()]))
    dist.destroy_process_group()


def test_shard_range_partitions_everything():
    for n in (0, 1, 7, 512):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_all_gather_onsets_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # expected: rank-ordered concatenation of the valid records, clip ids offset by the shard start
    exp = []
    for rank in range(2):
        recs, counts = _make_records(rank)
        lo, _ = shard_range(6, rank, 2)
        for c in range(3):
            for k in range(counts[c]):
                r = recs[c, k]
                exp.append((int(r["clip"]) + lo, int(r["channel"]), int(r["sample"])))
    assert got[0] == got[1] == exp


def test_single_process_is_identity():
    recs, counts = _make_records(0)
    flat = flatten_records(_as_u8(recs), torch.from_numpy(counts), 8)
    assert all_gather_onsets(flat) is flat and flat.shape == (7, 16)
    assert torch.equal(unpack_gathered(all_gather_onsets_padded(flat, 8)), flat)
    # one clip per rank and step: the block is built without flattening (what bench.py does)
    one = _as_u8(recs)[2:3]
    blk = pack_block(one, torch.from_numpy(counts[2:3]), 6, clip_offset=4)
    want = flatten_records(one, torch.from_numpy(counts[2:3]), 8, clip_offset=4)
    assert torch.equal(unpack_gathered(all_gather_blocks(blk)), want)



def test_pack_clips_compacts_without_data_dependent_shapes():
    """The block bench.py exchanges for a batch of clips: same records, same order as flatten_records,
    fixed shape; overflow of the block or of a clip's record capacity is an error at unpack time."""
    rng = np.random.default_rng(0)
    n_clips, cap = 5, 7
    recs = np.zeros((n_clips, cap), ONSET_DTYPE)
    recs["clip"] = np.arange(n_clips)[:, None]
    recs["channel"] = rng.integers(0, 4, (n_clips, cap))
    recs["sample"] = rng.integers(0, 1000, (n_clips, cap))
    r8 = _as_u8(recs)
    counts = torch.tensor([3, 0, 7, 2, 1])
    blk = pack_clips(r8, counts, 32, clip_offset=10)
    assert blk.shape == (33, 16)
    assert torch.equal(unpack_gathered(all_gather_blocks(blk)), flatten_records(r8, counts, cap, clip_offset=10))
    assert torch.equal(unpack_gathered(all_gather_blocks(pack_clips(r8, counts, 13))), flatten_records(r8, counts, cap))
    with pytest.raises(RuntimeError):
        unpack_gathered(all_gather_blocks(pack_clips(r8, counts, 12)))          # 13 records do not fit 12
    with pytest.raises(RuntimeError):
        unpack_gathered(all_gather_blocks(pack_clips(r8, torch.tensor([3, 0, 9, 2, 1]), 32)))  # clip 2 lost 2 records
    empty = pack_clips(r8, torch.zeros(5, dtype=torch.int64), 4)
    assert unpack_gathered(all_gather_blocks(empty)).shape == (0, 16)
