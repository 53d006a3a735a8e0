"""Multi-GPU form of the path: clips shard across ranks (one process per GPU),
no data-path collective; one exchange at the end collates the onset records
(SURVEY.md section 8e): an all-gather of the per-rank record counts followed by
an all-gather of the records padded to the largest count.  On ROCm the "nccl"
backend of torch.distributed is RCCL over xGMI; the payload is KBs, so the
exchange is latency-bound and a ring all-reduce is never needed.

Channels of one detector instance are coupled (detection.py:790), so a clip is
never split across ranks; a single long stream does not shard at all ("replicas
only").
"""
import numpy as np
import torch
import torch.distributed as dist

ONSET_DTYPE = np.dtype([("clip", np.int32), ("channel", np.int32), ("sample", np.int64)])


def shard_range(n_items, rank, world_size):
    """Contiguous block partition (first `n_items % world_size` ranks get one more)."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def flatten_records(records, counts, cap, clip_offset=0):
    """[n_clips, cap, 16] uint8 + counts -> [total, 16] uint8 with global clip ids
    (on the tensors' device; no host round trip)."""
    n_clips = counts.numel()
    idx = torch.arange(cap, device=records.device)[None, :] < counts.clamp(max=cap)[:, None]
    flat = records.reshape(n_clips, cap, 16)[idx]
    if clip_offset and flat.numel():
        clip = flat[:, :4].contiguous().view(torch.int32)
        clip += int(clip_offset)
        flat[:, :4] = clip.view(torch.uint8)
    return flat.contiguous()


def all_gather_onsets(local_records, group=None):
    """local_records: [n_local, 16] uint8 (ofp_onset structs) on this rank's device.
    Returns the concatenation over ranks, in rank order, on every rank."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_records
    world = dist.get_world_size(group)
    dev = local_records.device
    n_local = torch.tensor([local_records.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(counts)
    if m == 0:
        return local_records[:0]
    padded = torch.zeros((m, 16), dtype=torch.uint8, device=dev)
    padded[: local_records.shape[0]] = local_records
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def pack_block(records, counts, cap, clip_offset=0):
    """Detector output of ONE clip -> the fixed block `all_gather_onsets_padded` exchanges, built
    without a host round trip: records [1, cap_det, 16] uint8, counts [1] int64 -> [1 + cap, 16]
    (record 0 carries the count; rows beyond it are don't-care)."""
    assert records.shape[0] == 1, "pack_block handles one clip per rank and step; use flatten_records otherwise"
    dev = records.device
    block = torch.zeros((1 + cap, 16), dtype=torch.uint8, device=dev)
    m = min(cap, records.shape[1])
    block[1:1 + m] = records[0, :m]
    if clip_offset:
        clip = block[1:, :4].contiguous().view(torch.int32)
        clip += int(clip_offset)
        block[1:, :4] = clip.view(torch.uint8)
    block[0, 8:16] = counts[:1].to(torch.int64).view(torch.uint8)
    return block


def pack_clips(records, counts, cap_total, clip_offset=0):
    """Detector output of a batch of clips -> ONE fixed block for `all_gather_blocks`, compacted on the
    device without a host round trip (no data-dependent shape): records [n_clips, cap, 16] uint8,
    counts [n_clips] int64 -> [1 + cap_total, 16]; record 0 carries the total count (decoded and checked
    against cap_total by `unpack_gathered`), records 1.. are the valid records of clip 0, clip 1, ... in
    order with `clip_offset` added to their clip ids; rows beyond the count are don't-care."""
    n_clips, cap = records.shape[0], records.shape[1]
    dev = records.device
    if records.is_cuda:
        # one launch (ofp_pack_records) instead of a dozen small tensor operations; rows beyond the count stay
        # unwritten (don't-care for `unpack_gathered`)
        from . import _lib
        from .detection import _stream_ptr
        records = records.contiguous()
        counts = counts.to(device=dev, dtype=torch.int64).contiguous()
        block = torch.empty((cap_total + 1, 16), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().ofp_pack_records(records.data_ptr(), counts.data_ptr(), n_clips, cap, int(cap_total),
                                                   int(clip_offset), block.data_ptr(), _stream_ptr(dev)), "ofp_pack_records")
        return block
    c = counts.to(torch.int64).clamp(max=cap)
    offs = torch.cumsum(c, 0) - c                                   # first output row of each clip
    k = torch.arange(cap, device=dev, dtype=torch.int64)[None, :]
    pos = torch.where(k < c[:, None], 1 + offs[:, None] + k, torch.full_like(k, cap_total + 1))
    pos = pos.clamp(max=cap_total + 1).reshape(-1)                  # overflow and padding rows -> the dump row
    block = torch.zeros((cap_total + 2, 16), dtype=torch.uint8, device=dev)
    src = records.reshape(n_clips * cap, 16)
    if clip_offset:
        src = src.clone()
        clip = src[:, :4].contiguous().view(torch.int32)
        clip += int(clip_offset)
        src[:, :4] = clip.view(torch.uint8)
    block.index_copy_(0, pos, src)
    block[0] = 0
    block[0, 8:16] = c.sum().reshape(1).view(torch.uint8)
    # record 0's channel field flags a clip with more onsets than its record capacity (a lost record is an
    # error, not a truncation: `unpack_gathered` raises)
    block[0, 4:8] = (counts.to(torch.int64) > cap).any().to(torch.int32).reshape(1).view(torch.uint8)
    return block[: cap_total + 1]


def all_gather_blocks(block, group=None):
    """All-gather of one fixed block [1 + cap, 16] per rank (see `pack_block`) -> [world, 1 + cap, 16]:
    one collective, no host round trip."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return block[None]
    world = dist.get_world_size(group)
    out = torch.empty((world,) + tuple(block.shape), dtype=torch.uint8, device=block.device)
    dist.all_gather_into_tensor(out.view(world * block.shape[0], 16), block, group=group)
    return out


def all_gather_onsets_padded(local_records, cap, group=None):
    """`all_gather_blocks` for already flattened records [n, 16]: every rank contributes 1 + cap
    records (record 0 carries its true count in the `sample` field).  Decode with `unpack_gathered`
    (which checks for truncation)."""
    dev = local_records.device
    n = local_records.shape[0]
    block = torch.zeros((1 + cap, 16), dtype=torch.uint8, device=dev)
    block[0, 8:16] = torch.tensor([n], dtype=torch.int64, device=dev).view(torch.uint8)
    m = min(n, cap)
    block[1:1 + m] = local_records[:m]
    return all_gather_blocks(block, group)


def unpack_gathered(blocks):
    """[world, 1 + cap, 16] from `all_gather_onsets_padded` -> concatenated records [total, 16]."""
    cap = blocks.shape[1] - 1
    head = blocks[:, 0].cpu()
    counts = head[:, 8:16].contiguous().view(torch.int64).reshape(-1).tolist()
    if int(head[:, 4:8].contiguous().view(torch.int32).max()) != 0:
        raise RuntimeError("a clip produced more onsets than its record capacity (cap_per_clip)")
    if max(counts, default=0) > cap:
        raise RuntimeError(f"{max(counts)} onset records on a rank exceed the gather capacity {cap}")
    return torch.cat([blocks[r, 1:1 + c] for r, c in enumerate(counts)], dim=0)


def records_to_numpy(flat):
    return flat.cpu().numpy().reshape(-1, 16).view(ONSET_DTYPE).reshape(-1)
