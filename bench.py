#!/usr/bin/env python3
"""Benchmark of the onset-fingerprinting hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of detect -> rFFT |X|^2 -> 40-mel -> FCNN over one batch of
synthetic audio already resident in HBM, plus (N > 1) the RCCL all-gather that
collates onset records.  The detector is a chain of latency-bound recurrences that
fills a fraction of the chip, so `--inflight` steps (default 8) are in flight at a
time on each GPU, each on its own pipeline instance (work space, buffers, streams,
host thread); every step is still one complete pass over one clip, steps complete and
are gathered in order, and `config.latency_ms_per_step` reports what one step takes
(`--inflight 1`: strictly one after the other).  In flight, the detector uses its throughput
setting (`hp_span = 2`, `mm_chunk = 8192`, reported as `config.detector_tuning`; results are
identical for every tuning).  Workload at every N: BASELINE.json configs[1] ("C2":
8 ch x 60 s @ 48 kHz, 1024-point frames, hop 256) per GPU -- each rank owns an
independent 8-channel clip (channels of one detector are coupled and a stream does
not shard in time), so scaling is weak.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os

# Steps in flight live on separate HIP streams; by default the runtime multiplexes all streams of a
# process onto 4 hardware queues, which serialises unrelated steps behind each other's 2 ms kernels.
# Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

from onset_fingerprinting_amd import synth  # noqa: E402
from onset_fingerprinting_amd.distributed import (all_gather_blocks, pack_block,  # noqa: E402
                                                  records_to_numpy, unpack_gathered)
from onset_fingerprinting_amd.pipeline import FingerprintPipeline  # noqa: E402

SR, C, SECONDS, NFFT, HOP, NMELS = 48000, 8, 60.0, 1024, 256, 40
GATHER_CAP = 4096  # onset records per rank and step in the all-gather block (C2 has 952)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
VALU_NOFMA_PEAK_TFLOPS = 157.3 / 2  # the guide's fp32 vector peak counts an FMA as two operations
# algorithmic bytes per frame: every input sample read once, every
# required output written once
BYTES_DETECT = 2 * 4 * HOP                 # 4 B read + 4 B rel write per sample
BYTES_STFT = 4 * HOP + 4 * (NFFT // 2 + 1)  # new samples in, |X|^2 out
BYTES_MEL = 4 * (NFFT // 2 + 1) + 4 * NMELS
BYTES_MLP = 4 * NMELS + 4 * 8
# HBM bytes per launch of the dominant kernel from rocprofv3 PMC counters (FETCH_SIZE doubled +
# WRITE_SIZE, /opt/skills/guides/MI355X_MICROARCH.md section HBM); filled from profiles/, else null
TRAFFIC_BYTES_PER_LAUNCH = {}
try:  # measured with `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes), see profiles/README.md
    _t = json.load(open(REPO / "profiles" / "r01" / "v14_pmc_traffic_per_kernel.json"))
    TRAFFIC_BYTES_PER_LAUNCH = {"hp": _t["k_hp_candidates"]["hbm_mb_per_launch"] * 1e6,
                                "stft_mel": _t["k_stft_power"]["hbm_mb_per_launch"] * 1e6}
except Exception:
    pass


def cpu_baseline(x, seconds):
    """The CPU oracle (a port, 1 thread) on the first `seconds` of the same clip."""
    import oracle
    n = int(seconds * SR)
    xs = np.ascontiguousarray(x[:n])
    from onset_fingerprinting_amd.pipeline import seeded_fcnn
    sd = {k: v.numpy() for k, v in seeded_fcnn(NMELS, 8).state_dict().items()}
    fb = oracle.mel_filterbank(SR, NFFT, NMELS).astype(np.float64)
    t0 = time.perf_counter()
    ch, on, rel = oracle.detect_onsets_amplitude(xs, block_size=HOP, sr=SR)
    P = oracle.dense_power_frames(xs, NFFT, HOP)          # [C, H, bins]
    mel = P @ fb.T
    logits = oracle.fcnn_forward(sd, mel.reshape(-1, NMELS))
    dt = time.perf_counter() - t0
    frames = C * synth.n_frames(n, NFFT, HOP)
    return dict(value=frames / dt, seconds=dt, frames=frames, ch=np.array(ch), on=np.array(on), rel=rel,
                mel=mel, logits=logits.reshape(C, -1, 8))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--cpu-seconds", type=float, default=60.0, help="audio seconds given to the CPU baseline")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--inflight", type=int, default=8, help="steps (clips) processed concurrently per GPU")
    ap.add_argument("--tuning", type=str, default="", help="JSON dict of ofp_detect_tuning fields (experiments)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL ("nccl" on ROCm); OFP_BENCH_BACKEND=gloo lets two ranks share one GPU to rehearse the
        # control flow (RCCL refuses two ranks on one device)
        backend = os.environ.get("OFP_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
            local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    x = synth.c2_drums(SECONDS, C, SR, seed=1 + rank)
    xd = torch.from_numpy(x).to(dev).unsqueeze(0).contiguous()
    # The detector is a chain of latency-bound recurrences that fills a fraction of the chip, so
    # `--inflight` steps are processed concurrently, each by its own pipeline instance (work space,
    # output buffers, HIP streams) driven by its own host thread; a step is still one full pass
    # over one clip, and steps complete (and are all-gathered) in order.
    D = max(1, args.inflight)
    pipes = [FingerprintPipeline(C, NFFT, HOP, SR, NMELS, device=local) for _ in range(D)]
    streams = [torch.cuda.Stream(dev) for _ in range(D)]
    # With several steps in flight the detector runs in its throughput setting: ofp_detect_tuning.hp_span = 2
    # (one speculative IIR run serves two chunks: 2/3 of the work in half the waves, a slightly longer
    # launch) and mm_chunk = 8192 (tracker chunks twice as long: half the waves and half the overlapping
    # window reads, 0.1 ms more for a lone step); the one-step-at-a-time figure below uses a pipeline with
    # the library defaults.
    tuning = json.loads(args.tuning) if args.tuning else ({"hp_span": 2, "mm_chunk": 8192} if D > 1 else {})
    if tuning:
        for pp in pipes:
            pp.detector.set_tuning(**tuning)
    pipe = pipes[0]
    frames_per_rank = C * pipe.n_frames(x.shape[0])

    stage_acc = {}
    lat_acc = []
    gather_acc = []

    def run_step(w, timed):
        torch.cuda.set_device(local)
        t_in = time.perf_counter()
        with torch.cuda.stream(streams[w]):
            out = pipes[w].run(xd, timed=timed)
            # this rank's onset records as the fixed block the exchange uses, in a tensor of its own
            # (no host round trip; the pipeline's buffers are free for its next step on return)
            flat = pack_block(out["records"], out["counts"], GATHER_CAP, clip_offset=rank)
        streams[w].synchronize()
        return out, flat, time.perf_counter() - t_in

    def finish(out, flat, lat, timed):
        # the exchange: ONE all-gather of fixed-size blocks (count + records), no host round trip
        t_g = time.perf_counter()
        gathered = all_gather_blocks(flat)
        if timed:
            gather_acc.append(time.perf_counter() - t_g)  # host time of the exchange call (asynchronous under RCCL)
            st = dict(out["info"]["stage_ms"])
            st.pop("total")
            st.update(out["spectral_ms"])  # runs concurrently with the detector on a second stream
            for k, v in st.items():
                stage_acc[k] = stage_acc.get(k, 0.0) + v
            lat_acc.append(lat)
        return out, out["power"], out["mel"], out["logits"].reshape(-1, 8), gathered

    from concurrent.futures import ThreadPoolExecutor
    workers = [ThreadPoolExecutor(1) for _ in range(D)]  # worker w runs steps w, w+D, ... in order

    def run_steps(n, timed):
        # D steps in flight: worker w runs steps w, w+D, ... one after the other
        futs = [workers[i % D].submit(run_step, i % D, timed) for i in range(n)]
        res = None
        for f in futs:  # complete in step order
            out, flat, lat = f.result()
            res = finish(out, flat, lat, timed)
        return res

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    run_steps(max(args.warmup, D), False)
    barrier()
    t0 = time.perf_counter()
    res = run_steps(args.steps, True)
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * frames_per_rank / (ms_per_step / 1e3)

    # the same steps strictly one after the other (outside the timed region above), so that the
    # line also says what a single step costs when nothing else is in flight
    single_ms = ms_per_step
    if D > 1:
        n1 = max(1, min(args.steps, 10))
        w1 = D  # a pipeline of its own (default tuning), not one that holds a timed result
        pipes.append(FingerprintPipeline(C, NFFT, HOP, SR, NMELS, device=local))
        streams.append(torch.cuda.Stream(dev))
        run_step(w1, False)
        barrier()
        t1 = time.perf_counter()
        for _ in range(n1):
            _, flat1, _ = run_step(w1, False)
            all_gather_blocks(flat1)
        barrier()
        t = torch.tensor([(time.perf_counter() - t1) / n1 * 1e3], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        single_ms = float(t.item())

    if rank == 0:
        out, power, mel, logits, gathered = res
        gathered = unpack_gathered(gathered)  # decode (and check) the last step's exchange
        stage_ms = {k: v / args.steps for k, v in stage_acc.items()}
        stage_bytes = dict(hp=BYTES_DETECT, db=BYTES_DETECT, ar=BYTES_DETECT, rel=BYTES_DETECT, mm=BYTES_DETECT // 2,
                           logic=BYTES_DETECT // 2, stft_mel=BYTES_STFT + 4 * NMELS, mlp=BYTES_MLP)
        cand_ms = stage_ms.pop("hp_candidates", 0.0)  # a part of the hp stage, reported separately
        dom = max(stage_ms, key=stage_ms.get)
        passes = out["info"]
        launches = 1
        dom_ms = stage_ms[dom]
        if dom == "hp" and cand_ms > 0:
            # the dominant KERNEL is the single k_hp_candidates launch of the IIR stage: it reads
            # every input sample once (4 B) and keeps only chunk-boundary states
            dom_ms, dom_bytes = cand_ms, 4 * HOP
        else:
            dom_bytes = stage_bytes[dom]
        achieved = dom_bytes * frames_per_rank / (dom_ms / 1e3) / 1e9
        stage_ms["hp_candidates(part of hp)"] = cand_ms
        result = {
            "metric": "frames/sec (1024-pt, hop 256, 48 kHz) detect+FFT+classify at 1/2/4/8 MI355X",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2 per GPU: 8 ch x 60 s @ 48 kHz drum hits, 1024/256, "
                                   "detect + rFFT |X|^2 + 40-mel + FCNN(40-10-10-10-8); RCCL all-gather of onsets",
                       "frames_per_gpu": frames_per_rank, "onsets_gathered": int(gathered.shape[0]),
                       "parallelism": f"clips x{world}", "steps_in_flight_per_gpu": D, "detector_tuning": tuning,
                       "latency_ms_per_step": round(1e3 * float(np.mean(lat_acc)), 3),
                       "exchange_call_ms_per_step": round(1e3 * float(np.mean(gather_acc)), 3),
                       "one_step_at_a_time": {"ms_per_step": round(single_ms, 3),
                                              "frames_per_s": round(world * frames_per_rank / (single_ms / 1e3))}},
            "roofline": {"bound": "hbm", "kernel": {"hp": "k_hp_candidates", "ar": "k_ar_sym_local+k_ar_sym_combine+k_ar_warm2+k_ar_chunk",
                                                    "mm": "k_mm_warm2+k_mm_chunk", "db": "k_rect_db",
                                                    "rel": "k_rel_out", "logic": "k_block_scan+k_state_machine",
                                                    "stft_mel": "k_stft_power<1024> (mel fused)", "mlp": "k_dense"}[dom],
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": TRAFFIC_BYTES_PER_LAUNCH.get(dom), "launches_per_step": launches,
                         "avg_launch_ms": dom_ms / launches,
                         "algorithmic_bytes_per_frame": dom_bytes,
                         "note": "latency-bound sequential recurrence (8 chains): see DESIGN.md section 5"},
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "detector_passes": {k: passes[k] for k in ("hp_passes", "ar_passes", "mm_passes", "repaired")},
        }
        if dom == "hp" and cand_ms > 0 and passes.get("hp_candidate_steps"):
            # what actually bounds k_hp_candidates: the fp32 operations of its speculative IIR steps (17 each,
            # no FMA: every operation is rounded as the reference rounds it) against the vector fp32 rate
            # without FMA (157.3 TFLOP/s counts an FMA as 2)
            flop = 17.0 * passes["hp_candidate_steps"]
            tf = flop / (cand_ms / 1e3) / 1e12
            result["roofline"]["issue"] = {"bound": "valu fp32 without fma", "flop_per_launch": flop,
                                           "achieved": tf, "peak": VALU_NOFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                           "frac": tf / VALU_NOFMA_PEAK_TFLOPS,
                                           # the launches of all steps in flight together: one launch per ms_per_step
                                           "achieved_chip": flop / (ms_per_step / 1e3) / 1e12,
                                           "frac_chip": flop / (ms_per_step / 1e3) / 1e12 / VALU_NOFMA_PEAK_TFLOPS}
        if world == 1 and not args.no_cpu:
            cb = cpu_baseline(x, min(args.cpu_seconds, SECONDS))
            # parity of the timed GPU result against the oracle on the same sample
            n = int(min(args.cpu_seconds, SECONDS) * SR)
            recs = records_to_numpy(gathered)
            k = recs["sample"] < (n // HOP) * HOP
            # onsets of the prefix are identical only up to the last block boundary effects: compare strictly
            # on records whose block lies inside the sample
            ok_idx = np.array_equal(recs["channel"][k], cb["ch"]) and np.array_equal(recs["sample"][k], cb["on"])
            nb = (n // HOP) * HOP
            ok_rel = np.array_equal(out["rel"][0, :nb].cpu().numpy().view(np.uint32), cb["rel"].view(np.uint32))
            Hs = cb["mel"].shape[1]
            gm = mel[0, :, :Hs].cpu().numpy()
            mel_err = float(np.abs(gm - cb["mel"]).max() / cb["mel"].max())
            gl = logits.reshape(C, -1, 8)[:, :Hs].cpu().numpy()
            log_err = float(np.abs(gl - cb["logits"]).max() / np.abs(cb["logits"]).max())
            result["cpu_baseline"] = {"value": cb["value"], "unit": "frames/s", "cores": 1, "kind": "port",
                                      "sample": f"the first {min(args.cpu_seconds, SECONDS):.0f} s of the same 60 s clip "
                                                f"({cb['frames']} frames) through oracle/ (C detector + numpy "
                                                f"rFFT/mel/FCNN), {cb['seconds']:.2f} s wall"}
            result["parity"] = {"onset_indices_exact": bool(ok_idx), "rel_bit_exact": bool(ok_rel),
                                "mel_max_rel_err": mel_err, "logits_max_rel_err": log_err}
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
