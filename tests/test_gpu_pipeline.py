"""GPU parity of the composed path at the shapes of the other BASELINE configs
(C3: 64 channels, 2048/512 + mel + FCNN; C4: a batch of 4-channel clips; C5:
hipGraph-captured per-hop streaming), and size-independent properties at larger
sizes (idempotence, independence from the time-parallel tuning, batch == single)."""
import numpy as np
import pytest
import torch

import oracle
from onset_fingerprinting_amd import synth

pytestmark = pytest.mark.gpu
SR = 48000


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def mods():
    from onset_fingerprinting_amd import detection, pipeline
    return detection, pipeline


def test_c3_shape_64_channels_2048_512_with_mel_and_fcnn(mods):
    detection, pipeline = mods
    C, F, H = 64, 2048, 512
    x = synth.drum_hits(C, 3.0, SR, seed=2, amp_log_uniform=(0.05, 0.9))
    pipe = pipeline.FingerprintPipeline(C, F, H, SR, 40, device=0)
    out = pipe.run(torch.from_numpy(x).cuda().unsqueeze(0).contiguous())
    torch.cuda.synchronize()
    recs = detection.BatchDetector.records_to_numpy(out)[0]
    ch, on, rel = oracle.detect_onsets_amplitude(x, block_size=H, sr=SR)
    assert np.array_equal(recs["channel"], np.array(ch)) and np.array_equal(recs["sample"], np.array(on))
    assert len(ch) > 100
    assert np.array_equal(bits(out["rel"][0].cpu().numpy()), bits(rel))
    P = oracle.dense_power_frames(x, F, H)
    mel = P @ oracle.mel_filterbank(SR, F, 40).astype(np.float64).T
    assert np.abs(out["mel"][0].cpu().numpy() - mel).max() / mel.max() < 1e-4
    sd = {k: v.numpy() for k, v in pipe.classifier.state_dict().items()}
    ref = oracle.fcnn_forward(sd, mel.reshape(-1, 40)).reshape(C, -1, 8)
    assert np.abs(out["logits"][0].cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-4


def test_c4_shape_batch_of_clips_equals_clip_by_clip(mods):
    detection, pipeline = mods
    xs = np.stack([synth.c4_clip(i, 2.5, 4, SR) for i in range(12)])
    bd = detection.BatchDetector(4, 256, sr=SR)
    out = bd.detect(torch.from_numpy(xs).cuda())
    recs = detection.BatchDetector.records_to_numpy(out)
    rel = out["rel"].cpu().numpy()
    total = 0
    for i in range(len(xs)):
        ch, on, orel = oracle.detect_onsets_amplitude(xs[i], block_size=256, sr=SR)
        assert np.array_equal(recs[i]["channel"], np.array(ch)) and np.array_equal(recs[i]["sample"], np.array(on))
        assert (recs[i]["clip"] == i).all()
        assert np.array_equal(bits(rel[i]), bits(orel))
        total += len(ch)
    assert total > 50
    # the same clip alone gives the same answer as inside the batch
    one = detection.BatchDetector.records_to_numpy(bd.detect(torch.from_numpy(xs[5:6]).cuda()))[0]
    assert np.array_equal(one["sample"], recs[5]["sample"]) and np.array_equal(one["channel"], recs[5]["channel"])


def test_result_is_independent_of_time_parallel_tuning_and_idempotent(mods):
    """20 s x 8 ch: every chunking / warm-up / candidate setting must give the same bytes
    (the time-parallel passes are exact), and a second run repeats the first."""
    detection, _ = mods
    x = synth.c2_drums(20.0, 8, SR, seed=9)
    xd = torch.from_numpy(x).cuda().unsqueeze(0).contiguous()
    ref = None
    for tuning in (None,
                   dict(hp_chunk=4096, hp_warm=16384, hp_candidates=3, ar_chunk=8192, ar_warm=30000, mm_chunk=4096, mm_warm=20000),
                   dict(hp_chunk=50000, hp_candidates=1, ar_chunk=100000, ar_coarse_warm=-1, mm_chunk=30000, mm_warm=100000),
                   dict(concurrent_calls=8),   # the layout of a call that has an eighth of the GPU
                   dict(hp_dedupe=1),          # staged candidates with duplicate runs removed (the batch setting)
                   dict(hp_dedupe=1, hp_chunk=4096, hp_warm=20000, hp_candidates=5),
                   dict(hp_dedupe=1, hp_candidates=16, hp_candidate_offset=-1),
                   dict(hp_dedupe=-1, concurrent_calls=8),
                   dict(hp_early=1),           # whole re-runs that stop where they join a candidate
                   dict(hp_early=1, hp_candidates=2, hp_warm=6000, hp_chunk=4096),   # many breaks
                   dict(hp_early=1, hp_dedupe=1, hp_candidates=3, hp_warm=9000, hp_chunk=32768),
                   dict(hp_early=-1, hp_dedupe=1, hp_chunk=32768),
                   dict(fuse_db_sums=-1),      # dB and the closed-form guess's sums as two passes
                   dict(sm_segments=1),        # the hysteresis machine time-parallel over the visit list
                   dict(sm_segments=2),        # ... and the sequential machine deciding after it (the fallback path)
                   dict(sm_segments=-1),
                   dict(lane_merge=1),         # fast/slow follower and min/max as one lane per chunk
                   dict(lane_merge=1, ar_chunk=2048, ar_warm=9000, mm_chunk=2048, mm_warm=6000, ar_span=4, mm_span=4),
                   dict(lane_merge=1, mm_chunk=1024, mm_warm=-1),   # no tracker warm-up: the repair passes do the work
                   dict(lane_merge=1, line_stores=-1),   # lane-private stores in the output walks (default here: complete lines)
                   dict(lane_merge=1, hp_early=1, hp_candidates=2, hp_warm=6000, hp_chunk=4096),   # ... with many breaks
                   dict(lane_merge=1, walk_through=-1),  # every chunk through the chunk pass (default: walk-through chunks are pass 0)
                   dict(lane_merge=1, ar_span=8, mm_span=4, ar_chunk=8192, mm_chunk=8192),
                   dict(lane_merge=1, interleaved=-1),   # planar copies of `rel` and the input throughout
                   dict(lane_merge=1, hp_dedupe=1, interleaved=3),  # ... and none at all: every stage on the caller's arrays
                   dict(hp_dedupe=1, interleaved=2, hp_chunk=8192, hp_warm=12000, hp_candidates=4),
                   dict(concurrent_calls=64),
                   None):
        bd = detection.BatchDetector(8, 256, sr=SR)
        if tuning:
            bd.set_tuning(**tuning)
        out = bd.detect(xd)
        got = (out["records"].cpu().numpy().copy()[:, :int(out["counts"][0])], out["rel"].cpu().numpy().copy(),
               int(out["counts"][0]))
        if ref is None:
            ref = got
            ch, on, orel = oracle.detect_onsets_amplitude(x, block_size=256, sr=SR)
            assert got[2] == len(ch) and np.array_equal(bits(got[1][0]), bits(orel))
        else:
            assert got[2] == ref[2] and np.array_equal(got[0], ref[0]) and np.array_equal(bits(got[1]), bits(ref[1]))


def test_pipeline_without_planar_copies_gives_the_same_bytes(mods):
    """With the detector on the caller's interleaved audio (tuning interleaved 3) there is no planar copy for the
    spectral branch either: `planar_input` is None and the STFT takes its interleaved sliding form.  Records, `rel`,
    |X|^2, mel and logits equal the default pipeline's, byte for byte (4 and 8 channels)."""
    from onset_fingerprinting_amd.pipeline import FingerprintPipeline
    for C, gen in ((8, lambda i: synth.c2_drums(4.0, 8, SR, seed=40 + i)), (4, lambda i: synth.c4_clip(i, 4.0, 4, SR))):
        x = torch.from_numpy(np.stack([gen(i) for i in range(3)])).cuda().contiguous()
        outs = []
        for tuning in (dict(lane_merge=1, hp_dedupe=1), dict(lane_merge=1, hp_dedupe=1, interleaved=3)):
            pipe = FingerprintPipeline(C, 1024, 256, SR, 40)
            pipe.detector.set_tuning(**tuning)
            assert (pipe.detector.planar_input(x) is None) == (tuning.get("interleaved") == 3)
            o = pipe.run(x)
            torch.cuda.synchronize()
            outs.append({k: o[k].clone() for k in ("records", "counts", "rel", "power", "mel", "logits")})
        n = outs[0]["counts"]
        assert torch.equal(n, outs[1]["counts"]) and int(n.min()) > 20
        for i in range(3):
            assert torch.equal(outs[0]["records"][i, :int(n[i])], outs[1]["records"][i, :int(n[i])])
        for k in ("rel", "power", "mel", "logits"):
            assert torch.equal(outs[0][k], outs[1][k]), k


def test_c2_at_full_size_matches_the_oracle_bit_for_bit(mods):
    """BASELINE config C2 at its full size (8 ch x 60 s, block 256): onset records and the whole
    relative envelope against the oracle (1.2 s of CPU)."""
    detection, _ = mods
    x = synth.c2_drums(60.0, 8, SR, seed=1)
    xd = torch.from_numpy(x).cuda().unsqueeze(0).contiguous()
    bd = detection.BatchDetector(8, 256, sr=SR)
    out = bd.detect(xd)
    n = int(out["counts"][0])
    recs = detection.BatchDetector.records_to_numpy(out)[0]
    ch, on, orel = oracle.detect_onsets_amplitude(x, block_size=256, sr=SR)
    assert n == len(ch) and n > 900
    assert np.array_equal(recs["channel"], np.asarray(ch)) and np.array_equal(recs["sample"], np.asarray(on))
    assert np.array_equal(bits(out["rel"][0].cpu().numpy()), bits(orel))


def test_c3_size_properties_tuning_independence_and_channel_subsets(mods):
    """C3's channel count and block size at 64 ch x 100 s (the oracle would need seconds per
    channel-minute here): the result does not depend on the time-parallel tuning (the batch-adaptive
    layout picks 8 candidates and long chunks at this size; a second detector is forced to the
    small-batch layout), a second run repeats the first, and a clip made of the first 3 channels
    padded with silent ones gives channel-for-channel the same envelope for those 3 (`rel` of a
    channel depends on that channel alone; only the hysteresis couples channels)."""
    detection, _ = mods
    x = synth.c2_drums(100.0, 64, SR, seed=2)
    xd = torch.from_numpy(x).cuda().unsqueeze(0).contiguous()
    ref = None
    for tuning in (None, dict(hp_chunk=8192, hp_candidates=16, ar_chunk=4096, mm_chunk=4096), None,
                   dict(sm_segments=1),  # 64 channels' hysteresis machine in segments (automatic from 16 384 blocks)
                   dict(sm_segments=1, lane_merge=1, hp_dedupe=1)):
        bd = detection.BatchDetector(64, 512, sr=SR)
        if tuning:
            bd.set_tuning(**tuning)
        out = bd.detect(xd)
        n = int(out["counts"][0])
        got = (out["records"].cpu().numpy()[:, :n].copy(), out["rel"].cpu().numpy().copy())
        if ref is None:
            ref = got
            assert n > 5000
        else:
            assert np.array_equal(got[0], ref[0]) and np.array_equal(bits(got[1]), bits(ref[1]))
    sub = np.zeros_like(x[:, :8])
    sub[:, :3] = x[:, :3]
    bd = detection.BatchDetector(8, 512, sr=SR)
    out = bd.detect(torch.from_numpy(sub).cuda().unsqueeze(0).contiguous())
    assert np.array_equal(bits(out["rel"][0, :, :3].cpu().numpy()), bits(ref[1][0, :, :3]))
    # ... and the oracle agrees on a 5 s prefix of 4 of the 64 channels' envelopes
    ch, on, orel = oracle.detect_onsets_amplitude(np.ascontiguousarray(x[: 5 * SR, :4]), block_size=512, sr=SR)
    assert np.array_equal(bits(ref[1][0, : orel.shape[0], :4]), bits(orel))


def test_c5_streaming_per_hop_in_a_hip_graph(mods):
    """2-channel stream fed hop by hop (B = 256) through ONE captured graph per hop
    (copy-in -> k_stream); the onsets and the relative envelope equal the oracle's."""
    detection, _ = mods
    B, C = 256, 2
    x = synth.drum_hits(C, 2.0, SR, seed=4, period=0.21)
    nb = len(x) // B
    od = detection.AmplitudeOnsetDetector(C, B, sr=SR)
    oo = oracle.OracleDetector(C, B, sr=SR)
    od.init_minmax_tracker(x[: int(0.1 * SR)])
    oo.init_minmax_tracker(x[: int(0.1 * SR)])
    xd = torch.from_numpy(x[: nb * B]).cuda()
    hop_in = torch.empty((B, C), dtype=torch.float32, device="cuda")
    rel = torch.empty((B, C), dtype=torch.float32, device="cuda")
    rec = torch.empty((4096, 16), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        od.process(hop_in, 1, 0, rel, rec, cnt)  # warm the launch path outside capture
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    # reset what the warm call consumed
    od2 = detection.AmplitudeOnsetDetector(C, B, sr=SR)
    od2.init_minmax_tracker(x[: int(0.1 * SR)])
    cnt.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        od2.process(hop_in, 1, 0, rel, rec, cnt)
    rels = []
    for i in range(nb):
        hop_in.copy_(xd[i * B:(i + 1) * B])
        g.replay()
        rels.append(rel.clone())
    torch.cuda.synchronize()
    n = int(cnt.item())
    got = rec.cpu().numpy().view(detection.ONSET_DTYPE).reshape(-1)[:n]
    exp_c, exp_d, exp_rel = [], [], []
    for i in range(nb):
        c, d, r = oo(x[i * B:(i + 1) * B])
        exp_rel.append(r)
        exp_c += [int(v) for v in c]
        exp_d += [int(v) for v in d]  # deltas relative to the hop, as __call__ returns them
    assert n == len(exp_c) and n > 5
    assert [int(v) for v in got["channel"]] == exp_c and [int(v) for v in got["sample"]] == exp_d
    assert np.array_equal(bits(torch.stack(rels).cpu().numpy().reshape(-1, C)), bits(np.concatenate(exp_rel)))


def test_pipelines_in_flight_from_several_threads_give_the_same_bytes():
    """bench.py keeps several steps in flight, each pipeline instance driven by its own host thread on
    its own streams: the library must be re-entrant per handle.  Three different clips, processed
    concurrently several times over, must reproduce what each gives alone, byte for byte."""
    from concurrent.futures import ThreadPoolExecutor
    import torch
    from onset_fingerprinting_amd import synth
    from onset_fingerprinting_amd.pipeline import FingerprintPipeline
    C, sr = 4, 48000
    clips = [torch.from_numpy(synth.drum_hits(C, 6.0, sr, seed=50 + i, period=0.21 + 0.04 * i)).cuda().unsqueeze(0)
             for i in range(3)]
    pipes = [FingerprintPipeline(C, 1024, 256, sr, 40) for _ in range(3)]
    streams = [torch.cuda.Stream() for _ in range(3)]

    def snapshot(out):
        n = int(out["counts"][0])
        return (out["records"][0, :n].cpu().numpy().copy(), out["rel"].cpu().numpy().copy(),
                out["mel"].cpu().numpy().copy(), out["logits"].cpu().numpy().copy())

    alone = [snapshot(pipes[i].run(clips[i])) for i in range(3)]

    def work(i):
        torch.cuda.set_device(0)
        res = []
        with torch.cuda.stream(streams[i]):
            for _ in range(4):
                out = pipes[i].run(clips[i])
                streams[i].synchronize()
                res.append(snapshot(out))
        return res

    with ThreadPoolExecutor(3) as ex:
        results = list(ex.map(work, range(3)))
    for i in range(3):
        assert len(alone[i][0]) > 10
        for got in results[i]:
            for a, b in zip(alone[i], got):
                assert a.shape == b.shape and a.tobytes() == b.tobytes()


def test_collation_block_kernel_equals_the_tensor_form():
    """ofp_pack_records (one launch) against the device-independent tensor form of `pack_clips` (the one the gloo
    tests run on the CPU): header, order, clip offset, overflow flag, records that do not fit."""
    from onset_fingerprinting_amd.distributed import all_gather_blocks, flatten_records, pack_clips, unpack_gathered
    rng = np.random.default_rng(5)
    for n_clips, cap, cap_total, off in ((5, 8, 32, 10), (1, 4, 4, 0), (300, 16, 2000, 7), (64, 3, 50, 0)):
        rec = np.zeros((n_clips, cap), dtype=np.dtype([("clip", np.int32), ("channel", np.int32), ("sample", np.int64)]))
        rec["clip"] = np.arange(n_clips)[:, None]
        rec["channel"] = rng.integers(0, 8, (n_clips, cap))
        rec["sample"] = rng.integers(0, 1 << 40, (n_clips, cap))
        r8 = torch.from_numpy(rec.view(np.uint8).reshape(n_clips, cap, 16).copy())
        counts = torch.from_numpy(rng.integers(0, cap + 1, n_clips).astype(np.int64))
        if n_clips == 64:
            counts[3] = cap + 2  # a clip that lost records: flagged in the header
        want = pack_clips(r8, counts, cap_total, clip_offset=off)          # CPU tensors: the tensor form
        got = pack_clips(r8.cuda(), counts.cuda(), cap_total, clip_offset=off).cpu()   # ofp_pack_records
        total = int(counts.clamp(max=cap).sum())
        n = 1 + min(total, cap_total)
        assert torch.equal(got[:n], want[:n]), (n_clips, cap, cap_total)
        if total <= cap_total and n_clips != 64:
            assert torch.equal(unpack_gathered(all_gather_blocks(got)), flatten_records(r8, counts, cap) if off == 0 else
                               unpack_gathered(all_gather_blocks(want)))


def test_two_ranks_on_one_gpu_gather_what_one_process_computes(mods, tmp_path):
    """The N > 1 path end to end with real device buffers: two FRESH child processes (gloo, both on cuda:0, started
    before they touch the GPU) each run the detector on their shard of 16 C4 clips, build the collation block on the
    device (ofp_pack_records) and exchange it; what rank 0 gathers equals the records of ONE process over all 16
    clips, and the oracle's for two of them (bench.py's N > 1 step without the timing)."""
    import socket
    import subprocess
    import sys
    from pathlib import Path
    detection, _ = mods
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tmp_path / "gathered.npy"
    child = str(Path(__file__).with_name("_two_ranks_child.py"))
    procs = [subprocess.Popen([sys.executable, child, str(r), "2", str(port), str(out)]) for r in range(2)]
    try:
        codes = [pr.wait(timeout=300) for pr in procs]
    finally:
        for pr in procs:  # exactly the children started above
            if pr.poll() is None:
                pr.kill()
    assert codes == [0, 0]
    got = np.load(out)
    xs = np.stack([synth.c4_clip(i, 3.0, 4, SR) for i in range(16)])
    one = detection.BatchDetector.records_to_numpy(detection.BatchDetector(4, 256, sr=SR).detect(torch.from_numpy(xs).cuda()))
    want = np.concatenate(one)
    assert len(got) == len(want) > 100
    for f in ("clip", "channel", "sample"):
        assert np.array_equal(got[f], want[f]), f
    for i in (3, 12):  # one clip of each rank's shard against the oracle
        ch, on, _ = oracle.detect_onsets_amplitude(xs[i], block_size=256, sr=SR)
        mine = got[got["clip"] == i]
        assert np.array_equal(mine["channel"], np.array(ch)) and np.array_equal(mine["sample"], np.array(on))


@pytest.mark.slow
def test_bench_two_ranks_with_several_steps_per_call(tmp_path):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one rank per process), both ranks on this
    one GPU over gloo: each rank's 256-clip shard of C4 goes through the library two steps per call, every step's
    records are exchanged on their own, and rank 0 prints the one JSON line."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from pathlib import Path
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, OFP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    bench_py = str(Path(__file__).resolve().parents[1] / "bench.py")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), bench_py, "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--inflight", "2", "--no-cpu", "--no-extras"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    c = j["config"]
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["scaling"] == "strong" and j["value"] > 0
    assert c["ranks_in_exchange"] == 2 and c["steps_per_call"] == 2 and c["clips_per_gpu_per_step"] == 256
    assert c["onsets_gathered"] > 512 * 50  # (a C4 clip has ~140 onsets: both ranks' records arrived)


@pytest.mark.slow
def test_c4_at_its_stated_size_512_clips_on_one_gpu(mods):
    """BASELINE configs[3] at full size on one GPU: 512 clips x 4 ch x 10 s in ONE call (2 048 chains) in the bench's
    throughput settings == the same clips one call each in the default settings (every 8th clip), and 16 sampled
    clips == the oracle (indices exact, `rel` bit for bit)."""
    detection, _ = mods
    import bench
    xs = np.stack(bench.synth_batch("c4", list(range(512)), 10.0, 4, 16))
    xd = torch.from_numpy(xs).cuda().contiguous()
    bd = detection.BatchDetector(4, 256, sr=SR)
    bd.set_tuning(lane_merge=1, hp_dedupe=1)
    out = bd.detect(xd, cap_per_clip=1024)
    # A fresh detector enqueues few verifying passes ahead; these 2 048 sparse-hit chains need more for the tracker, so the
    # first call repeats that stage host-verified (info 15: 1 + 4) and the detector enqueues more from then on: the
    # second call converges within what it enqueued.  Both give the same bytes.
    first = (bd.last_info["repeated_host_verified"], out["rel"].clone(), out["counts"].clone())
    assert first[0] in (0, 5)
    out = bd.detect(xd, out=out, cap_per_clip=1024)
    assert bd.last_info["repeated_host_verified"] == 0
    assert torch.equal(out["rel"], first[1]) and torch.equal(out["counts"], first[2])
    recs = detection.BatchDetector.records_to_numpy(out)
    single = detection.BatchDetector(4, 256, sr=SR)
    for i in range(0, 512, 8):
        o1 = single.detect(xd[i:i + 1], cap_per_clip=1024)
        r1 = detection.BatchDetector.records_to_numpy(o1)[0]
        assert np.array_equal(r1["sample"], recs[i]["sample"]) and np.array_equal(r1["channel"], recs[i]["channel"]), i
        assert torch.equal(o1["rel"][0], out["rel"][i]), i
    total = 0
    for i in range(5, 512, 32):
        ch, on, orel = oracle.detect_onsets_amplitude(xs[i], block_size=256, sr=SR)
        assert np.array_equal(recs[i]["channel"], np.array(ch)) and np.array_equal(recs[i]["sample"], np.array(on)), i
        assert (recs[i]["clip"] == i).all()
        assert np.array_equal(bits(out["rel"][i].cpu().numpy()), bits(orel)), i
        total += len(ch)
    assert total > 1500


@pytest.mark.slow
def test_c3_at_its_stated_size_64_channels_600_s(mods):
    """BASELINE configs[2] at full size: ONE clip of 64 ch x 600 s @ 48 kHz (1.84 G samples), 2048/512, 40 mel + FCNN with
    |X|^2 kept on the chip (want_power=False): onset indices and the relative envelope equal the oracle's over the
    WHOLE clip (exact; the oracle needs most of a minute), mel / logits on two channels within 1e-4."""
    detection, pipeline = mods
    C, F, H = 64, 2048, 512
    x = synth.c3_stream(600.0, C, SR, seed=2)
    pipe = pipeline.FingerprintPipeline(C, F, H, SR, 40, device=0, want_power=False)
    out = pipe.run(torch.from_numpy(x).cuda().unsqueeze(0).contiguous())
    torch.cuda.synchronize()
    assert out["power"] is None and not out["info"]["repeated_host_verified"]
    n = int(out["counts"][0])
    assert 0 < n <= out["cap"]
    rec = out["records"][0, :n].cpu().numpy().view(np.dtype([("clip", np.int32), ("channel", np.int32),
                                                            ("sample", np.int64)])).reshape(-1)
    ch, on, rel = oracle.detect_onsets_amplitude(x, block_size=H, sr=SR)
    assert len(ch) > 20000
    assert np.array_equal(rec["channel"], np.array(ch)) and np.array_equal(rec["sample"], np.array(on))
    g_rel = out["rel"][0].cpu().numpy()
    assert np.array_equal(g_rel.view(np.uint32), rel.view(np.uint32))
    del rel, g_rel
    sd = {k: v.numpy() for k, v in pipe.classifier.state_dict().items()}
    fb = oracle.mel_filterbank(SR, F, 40).astype(np.float64)
    Hn = 2000
    ns = F + (Hn - 1) * H
    for c in (0, C - 1):
        P = oracle.dense_power_frames(np.ascontiguousarray(x[:ns, c:c + 1]), F, H)[0]
        mel = P @ fb.T
        big = mel >= 1e-5 * mel.max()
        gm = out["mel"][0, c, :Hn].cpu().numpy()
        assert (np.abs(gm - mel)[big] / mel[big]).max() < 1e-4 and np.abs(gm - mel).max() / mel.max() < 1e-4
        lg = oracle.fcnn_forward(sd, mel)
        assert np.abs(out["logits"][0, c, :Hn].cpu().numpy() - lg).max() / np.abs(lg).max() < 1e-4
