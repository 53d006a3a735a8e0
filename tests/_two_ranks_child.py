"""Child process of tests/test_gpu_pipeline.py::test_two_ranks_on_one_gpu...: ONE rank of a two-rank gloo job whose
ranks share cuda:0.  It runs the real detector on its shard of 16 C4 clips, packs the collation block on the device
(`pack_clips` -> ofp_pack_records), exchanges it with `all_gather_blocks` and decodes with `unpack_gathered`; rank 0
saves what it gathered.  Started fresh by the test (nothing here runs in the pytest process).

    python tests/_two_ranks_child.py <rank> <world> <port> <out.npy>
"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist

    from onset_fingerprinting_amd import synth
    from onset_fingerprinting_amd.detection import BatchDetector
    from onset_fingerprinting_amd.distributed import (all_gather_blocks, pack_clips, records_to_numpy, shard_range,
                                                      unpack_gathered)
    dist.init_process_group("gloo", rank=rank, world_size=world, init_method=f"tcp://127.0.0.1:{port}")
    torch.cuda.set_device(0)
    n_total, C, secs, sr = 16, 4, 3.0, 48000
    lo, hi = shard_range(n_total, rank, world)
    x = torch.from_numpy(np.stack([synth.c4_clip(i, secs, C, sr) for i in range(lo, hi)])).cuda().contiguous()
    bd = BatchDetector(C, 256, sr=sr)
    bd.set_tuning(lane_merge=1, hp_dedupe=1)  # (the bench's settings for steps in flight)
    det = bd.detect(x, cap_per_clip=512)
    blk = pack_clips(det["records"], det["counts"], 8 * 256, clip_offset=lo)   # device block, one launch
    gathered = all_gather_blocks(blk.cpu())   # gloo exchanges host tensors; the block itself was built on the device
    recs = records_to_numpy(unpack_gathered(gathered))
    if rank == 0:
        np.save(out, recs)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
