#!/usr/bin/env python3
"""Benchmark of the onset-fingerprinting hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c2|c4] [--clips n] [--inflight D]

One step = one pass of detect -> rFFT |X|^2 -> 40-mel -> FCNN (the last three in ONE kernel) over one
batch of synthetic clips already resident in HBM, followed by the exchange that collates the onset
records (an RCCL all-gather when N > 1).  Rank 0 prints ONE JSON line.

Launch.  `--gpus N` with N > 1 and no RANK in the environment makes THIS process a launcher: before
anything touches a GPU it starts N ranks of itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one
process per GPU), waits for them, relays rank 0's JSON line and exits non-zero if any rank failed.
Under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks are already there;
`--gpus` must then equal WORLD_SIZE (anything else is refused, loudly).

Workloads (BASELINE.json configs):
  c2  (default at N = 1)  `--clips` (default 48) DISTINCT 8-channel 60 s clips per GPU and step
      (configs[1], the configuration the metric is quoted on, as a batch: one library call processes
      the clips side by side -- 384 chains, 4.4 GB of audio; rounds 1-2 used 16: the larger batch lets the
      detector's speculative passes use longer chunks, 145 -> 173 M frames/s with the same command; 212 M on the last build of round 3).
      Every rank owns its own clips: scaling "weak".
      `config.one_clip_per_step` is the same path on ONE clip per step (8 chains, latency-bound).
  c4  (default at N > 1)  512 clips x 4 ch x 10 s in total, sharded contiguously over the ranks
      (configs[3]): total work fixed, scaling "strong".  `config.whole_batch_on_one_gpu` (rank 0, after
      the timed region, four steps in flight) is the same 512 clips on one GPU -- the base the strong-scaling
      ratio refers to.
`--inflight D` keeps D steps in flight per GPU (default 6 for c2; 6 / 12 for C4 shards of > 128 / <= 128 clips), each on its own pipeline instance with its own DISTINCT clips, streams and host thread: while one
batch sits in the latency-bound verification rounds of its detector the others keep the chip busy.  With steps
overlapping the detector runs in its throughput settings (`config.detector_tuning`: one lane per chunk for both
followers / both tracker words, IIR candidates in stages; small C4 shards also the layout hint
`concurrent_calls`) -- every setting gives the same bytes (tests/test_gpu_pipeline.py).  The K timed steps are bracketed by barriers as the
contract asks (the pipeline's ramp-up and drain are inside the window); `config.latency_ms_per_step` is what
one step takes meanwhile.
`--steps-per-call G` (C4 shards of at most 256 clips: G x clips <= 640 by default, G a divisor of K): the resident
batches of G consecutive steps go through ONE library call side by side; every step's records are still packed and
exchanged on their own, in step order (`config.steps_per_call`, `config.calls_in_flight_per_gpu`).  `--inflight`
then counts calls.

`--rehearsal` (no GPU needed): the launcher, rendezvous, sharding, exchange, max-over-ranks timing and
JSON line with fabricated onset records instead of GPU work -- the control flow of the N > 1 path for
CPU tests (gloo).  Its `value` is not a measurement and says so.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

# Steps in flight live on separate HIP streams; by default the runtime multiplexes all streams of a
# process onto 4 hardware queues.  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

SR, NFFT, HOP, NMELS, NOUT = 48000, 1024, 256, 40, 8
C2 = dict(C=8, seconds=60.0)
C4 = dict(C=4, seconds=10.0, clips=512)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
VALU_NOFMA_PEAK_TFLOPS = 157.3 / 2  # the guide's fp32 vector peak counts an FMA as two operations
# algorithmic bytes per frame: every input sample read once, every required output written once
BYTES_DETECT = 2 * 4 * HOP                  # 4 B read + 4 B rel write per sample
BYTES_SPECTRUM = 4 * (NFFT // 2 + 1)        # |X|^2 out (the samples are the detector's: read once)
BYTES_FINGERPRINT = 4 * NMELS + 4 * NOUT    # mel + logits out
BYTES_E2E = BYTES_DETECT + BYTES_SPECTRUM + BYTES_FINGERPRINT  # 4 100 + 192
# PMC traffic per launch of the dominant kernels (FETCH_SIZE / WRITE_SIZE passes, corrected as the guide's
# HBM section prescribes) is measured by tools/profile_round.sh, not in this run: the line names its source
TRAFFIC_FILE = REPO / "profiles" / "r03" / "pmc_traffic_per_kernel.json"
SQ_FILE = REPO / "profiles" / "r03" / "pmc_sq_per_kernel.json"   # instruction-level counters per kernel (tools/profile_r3.sh)
MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_16x16x4_f32 / 32x32x2: the FP32 matrix peak of the guide


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n):
    """The launcher: N children, one per GPU; relays rank 0's JSON line; non-zero exit if any rank fails
    (the remaining ranks are then stopped -- they would wait for the lost one at the next barrier)."""
    import tempfile
    port = free_port()
    procs = []
    with tempfile.TemporaryFile() as out0:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        deadline = time.time() + float(os.environ.get("OFP_BENCH_LAUNCH_TIMEOUT", "1500"))
        failed = []
        while True:
            codes = [p.poll() for p in procs]
            failed = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if failed or all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                failed = [(r, "timeout") for r, c in enumerate(codes) if c is None]
                break
            time.sleep(0.05)
        for p in procs:  # exactly the children started above
            if p.poll() is None:
                p.kill()
                p.wait()
        out0.seek(0)
        lines = out0.read().decode(errors="replace").splitlines()
    relayed = False
    for line in lines:
        if line.startswith("{") and not failed:
            print(line)
            relayed = True
        elif line.strip():
            sys.stderr.write(line + "\n")  # library chatter of rank 0 is not part of the contract's ONE line
    sys.stdout.flush()
    if failed or not relayed:
        sys.stderr.write(f"bench.py: ranks failed (rank, exit code): {failed}\n")
        return 1
    return 0


def usable_cpus():
    """Worker processes this run may start: the CPUs it is allowed on (affinity mask and cgroup CPU quota), at most
    16 (a GPU box hands one GPU's job 16 of its cores whatever os.cpu_count() says; every worker also holds its own
    copy of a clip)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:  # cgroup v2: "<quota> <period>" or "max <period>"
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(16, n))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def fcnn_state():
    from onset_fingerprinting_amd.pipeline import seeded_fcnn
    return {k: v.numpy() for k, v in seeded_fcnn(NMELS, NOUT).state_dict().items()}


def oracle_pass(x, sd):
    """The CPU oracle (a port, 1 thread) on one clip: detect + dense |X|^2 + mel + FCNN."""
    import numpy as np

    import oracle
    fb = oracle.mel_filterbank(SR, NFFT, NMELS).astype(np.float64)
    t0 = time.perf_counter()
    ch, on, rel = oracle.detect_onsets_amplitude(x, block_size=HOP, sr=SR)
    P = oracle.dense_power_frames(x, NFFT, HOP)          # [C, H, bins]
    mel = P @ fb.T
    logits = oracle.fcnn_forward(sd, mel.reshape(-1, NMELS))
    dt = time.perf_counter() - t0
    return dict(seconds=dt, frames=x.shape[1] * P.shape[1], ch=np.array(ch), on=np.array(on), rel=rel, mel=mel,
                logits=logits.reshape(x.shape[1], -1, NOUT))


def _fanout_worker(job):
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):  # one core per worker, as stated
        os.environ[v] = "1"
    kind, seed, seconds, sd = job
    from onset_fingerprinting_amd import synth
    x = synth.c2_drums(seconds, C2["C"], SR, seed=seed) if kind == "c2" else synth.c4_clip(seed, seconds, C4["C"], SR)
    r = oracle_pass(x, sd)
    return r["frames"], r["seconds"]


def _synth_worker(job):
    kind, ident, seconds, C = job
    from onset_fingerprinting_amd import synth
    return synth.c2_drums(seconds, C, SR, seed=ident) if kind == "c2" else synth.c4_clip(ident, seconds, C, SR)


def under_profiler():
    """rocprofv3 preloads its tool library into this process AND into every child; with --pmc that library has
    initialised the GPU before main() runs, and a process that has done so must not start (exec) others on this pool.
    Round 2's first counter pass of tools/pmc_detect.sh sat in exactly that: eight spawned synthesis workers, each
    with the counter service attached, were SIGTERMed by Pool.__exit__ while the tool was finalising in them, and the
    parent then never came out of its own counter-service initialisation (no "HSA version ... initialized" line in
    gpurun_out/pmc_detect/SQ_INSTS_VALU...log before the 200 s limit).  Under a profiler nothing is spawned."""
    e = os.environ
    return any("rocprof" in e.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")) or \
        any(k.startswith("ROCPROF") for k in e)


def synth_batch(kind, idents, seconds, C, workers):
    """The clips of one batch (host synthesis, several processes: a C2 clip takes seconds of numpy).  OFP_SYNTH_CACHE
    names a directory of per-clip .npy files (filled on first use): a profiled run loads its clips from there instead
    of synthesising them in child processes (see under_profiler)."""
    import numpy as np
    jobs = [(kind, int(i), seconds, C) for i in idents]
    cache = os.environ.get("OFP_SYNTH_CACHE")
    if cache:
        os.makedirs(cache, exist_ok=True)
        paths = [Path(cache) / f"{k}_{i}_{s:g}_{c}.npy" for k, i, s, c in jobs]
        missing = [j for j, p in zip(jobs, paths) if not p.exists()]
        if missing:
            made = synth_batch_uncached(missing, 1 if under_profiler() else workers)
            for j, x in zip(missing, made):
                np.save(Path(cache) / f"{j[0]}_{j[1]}_{j[2]:g}_{j[3]}.npy", x)
        return [np.load(p) for p in paths]
    return synth_batch_uncached(jobs, 1 if under_profiler() else workers)


def synth_batch_uncached(jobs, workers):
    if workers <= 1 or len(jobs) == 1:
        return [_synth_worker(j) for j in jobs]
    import multiprocessing as mp
    pool = mp.get_context("spawn").Pool(min(workers, len(jobs)))
    try:
        return pool.map(_synth_worker, jobs, chunksize=max(1, len(jobs) // (4 * workers)))
    finally:  # close + join: the workers leave by themselves (terminate() signals them while a profiler's preloaded
        pool.close()  # library is finalising in them, which fills its log with "Aborted")
        pool.join()


_CAL = {}


def _spin_worker(seconds):
    """The fan-out's own work on a SHORT clip (oracle detector + numpy rFFT / mel / FCNN, single-threaded BLAS): what
    the calibration times side by side.  (A pure-Python spin loop calibrated to 16 on a box whose 16 oracle workers
    then ran 12 x slower than one alone: the work is bound by the memory system, not by issue slots.)  -> its seconds."""
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = "1"
    if "x" not in _CAL:
        from onset_fingerprinting_amd import synth
        _CAL["x"] = synth.c2_drums(float(seconds), C2["C"], SR, seed=77)
        _CAL["sd"] = fcnn_state()
        oracle_pass(_CAL["x"][: SR // 2], _CAL["sd"])   # (imports, table builds)
    return oracle_pass(_CAL["x"], _CAL["sd"])["seconds"]


def calibrated_cores(limit):
    """How many processes this job can really run side by side: the fan-out's own work (a 4 s clip through the oracle) in
    k = 1, 2, 4 ... processes until the slowest of them takes 20 % longer than one alone (an affinity mask or cpu_count() says what the box has, not
    what this job is given).  -> (cores, {k: seconds of the slowest})"""
    import multiprocessing as mp
    curve, cores, base = {}, 1, None
    k = 1
    while k <= limit:
        pool = mp.get_context("spawn").Pool(k)
        try:
            pool.map(_spin_worker, [4.0] * k, chunksize=1)             # (processes up, clip synthesised, imports done)
            worst = max(pool.map(_spin_worker, [4.0] * k, chunksize=1))
        finally:
            pool.close()
            pool.join()
        curve[k] = round(worst, 3)
        if base is None:
            base = worst
        if worst > 1.2 * base:
            break
        cores = k
        k *= 2
    return cores, curve


def cpu_fanout(kind, seconds, sd):
    """One process per core this job really has (calibrated_cores), one clip each (the process-per-core
    fan-out over clips)."""
    import multiprocessing as mp
    n, curve = calibrated_cores(usable_cpus())
    jobs = [(kind, 1000 + i, seconds, sd) for i in range(n)]
    t0 = time.perf_counter()
    pool = mp.get_context("spawn").Pool(n)
    try:
        res = pool.map(_fanout_worker, jobs, chunksize=1)
    finally:
        pool.close()
        pool.join()
    wall = time.perf_counter() - t0
    busy = max(s for _, s in res)  # the clips run side by side: the slowest one bounds the compute time
    return dict(value=sum(f for f, _ in res) / busy, cores=n, wall_s=round(wall, 2), spin_seconds_by_processes=curve,
                sample=f"{n} processes (the count at which the oracle on a 4 s clip still runs within 20 % of its lone speed; the mask "
                       f"allows {usable_cpus()}) x one {seconds:.0f} s clip each through oracle/, {busy:.2f} s for the "
                       "slowest (process start-up and clip synthesis excluded)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["c2", "c4"], default=None)
    ap.add_argument("--clips", type=int, default=48, help="c2: distinct clips per GPU and step (48: 4.4 GB of audio per step, six steps in flight use ~63 %% of the 288 GB)")
    ap.add_argument("--inflight", type=int, default=0, help="steps in flight per GPU (0: 6 for c2; 6 / 12 for c4 shards of > 128 / <= 128 clips; with two or more in flight the detector runs with its throughput settings lane_merge and hp_dedupe unless --tuning says otherwise)")
    ap.add_argument("--steps-per-call", type=int, default=0, help="c4: consecutive steps whose (distinct) batches ONE library call "
                    "processes side by side; every step keeps its own exchange (0: small shards are grouped up to ~384 clips per "
                    "call, by a divisor of --steps; 1: never)")
    ap.add_argument("--cpu-seconds", type=float, default=60.0, help="audio seconds of one clip given to the CPU baseline")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extra figures (one clip per step, whole batch)")
    ap.add_argument("--tuning", type=str, default="", help="JSON dict of ofp_detect_tuning fields (experiments)")
    ap.add_argument("--exp", type=str, default="", help="EXPERIMENTS (the line says so in config.experiment and is not the "
                    "metric): JSON dict with detector kwargs (hipass_freq, on_threshold, ...) and/or \"no_stft\": true")
    ap.add_argument("--rehearsal", action="store_true", help="control flow only, no GPU work (CPU tests)")
    ap.add_argument("--shard-of", type=int, default=0, help="c4 on ONE GPU: process only rank 0's shard of a world of this "
                    "size (what one rank of an N-GPU run does per step, without the exchange partners); the line is "
                    "marked as such and its value counts this shard's frames only")
    args = ap.parse_args()

    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))  # nothing in this process has touched a GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: refusing to print a line whose n_gpus "
                         "is not what was asked for (launch with --nproc-per-node equal to --gpus, or let bench.py "
                         "start the ranks itself)\n")
        sys.exit(2)
    if os.environ.get("OFP_BENCH_TEST_FAIL_RANK") == str(rank):  # test hook: a rank that dies at start-up
        sys.exit(3)
    workload = args.workload or ("c2" if world == 1 else "c4")
    rehearsal = args.rehearsal or os.environ.get("OFP_BENCH_REHEARSAL") == "1"

    import numpy as np
    import torch
    import torch.distributed as dist

    from onset_fingerprinting_amd import synth
    from onset_fingerprinting_amd.distributed import (all_gather_blocks, pack_clips, records_to_numpy, shard_range,
                                                      unpack_gathered)

    backend = os.environ.get("OFP_BENCH_BACKEND", "gloo" if rehearsal else "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":  # RCCL on ROCm
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:  # gloo: the ranks may share one GPU (or none, in a rehearsal)
            dist.init_process_group(backend, rank=rank, world_size=world)
            local = 0
    dev = torch.device("cpu") if rehearsal else torch.device("cuda", local)
    if not rehearsal:
        torch.cuda.set_device(local)

    # ---- the rank's share of the workload
    if workload == "c2":
        C, seconds = C2["C"], C2["seconds"]
        n_local, clip_lo, total_clips = max(1, args.clips), rank * max(1, args.clips), world * max(1, args.clips)
        scaling = "weak"
    else:
        C, seconds = C4["C"], C4["seconds"]
        clip_lo, clip_hi = shard_range(C4["clips"], rank, args.shard_of if (world == 1 and args.shard_of > 1) else world)
        n_local, total_clips = clip_hi - clip_lo, (C4["clips"] if not (world == 1 and args.shard_of > 1) else clip_hi - clip_lo)
        scaling = "strong"
    N = int(seconds * SR)
    H = synth.n_frames(N, NFFT, HOP)
    frames_local = n_local * C * H
    frames_total = total_clips * C * H
    cap_clip = 4096 if workload == "c2" else 1024      # onset records per clip (C2 has ~950, a C4 clip ~160)
    # records per rank in the exchanged block: the SAME size on every rank (an all-gather of equal blocks),
    # so it is sized for the largest shard
    n_local_max = n_local if workload == "c2" else -(-C4["clips"] // (args.shard_of if (world == 1 and args.shard_of > 1) else world))
    cap_block = n_local_max * (1024 if workload == "c2" else 256)
    # steps in flight (measured on one GPU, tools/share_sweep*.sh, ms per step, this build): C2 x 16: 11.5 / 10.6 / 10.6 at
    # 4 / 6 / 8 in flight (C2 x 32: 21.8 / 20.4 at 3 / 4); C4, all 512 clips: 24.9 / 23.8 at 4 / 6; a rank's share
    # alone: 256 clips 14.4 / 13.7 at 4 / 6; 128 clips 7.4 / 7.0 at 8 / 12; 64 clips 4.1 at 12, 4.7 at 16 (9.0 / 7.3 / 6.1 /
    # 5.7 at 1 / 2 / 4 / 8 before the layout hint below).
    if args.inflight > 0:
        D = args.inflight
    elif workload == "c4":
        D = 12 if n_local <= 128 else 6
    else:
        D = 6
    # Steps per call (strong scaling, small shards): a rank's 64 clips are a small call -- twelve of them in flight reach
    # 123 M frames/s where 512 clips per call reach 184 M.  The steps are independent and their batches
    # are resident before the timed region, so G consecutive steps' batches go through ONE call (G x n_local clips side
    # by side, ~512-640 clips); each step's records are then packed and exchanged on their own, in step order.  G divides
    # K so that the timed region is whole calls.
    G = 1
    if workload == "c4" and not rehearsal and args.steps_per_call != 1:
        n_ref = -(-C4["clips"] // (args.shard_of if (world == 1 and args.shard_of > 1) else world))  # the same on every rank
        lim = args.steps_per_call if args.steps_per_call > 1 else (640 // n_ref if n_ref <= 256 else 1)
        G = max([g for g in range(1, max(lim, 1) + 1) if args.steps % g == 0] or [1])
        if G > 1 and args.inflight <= 0:
            D = 4 if n_local <= 128 else 6   # CALLS in flight (each G steps)
    auto_tuning = {}
    if D >= 2:
        # steps overlap: the detector's throughput setting (fast/slow follower and min/max as one lane per chunk:
        # half the reads of those passes; 16 x C2, four in flight: 116 -> 127 M frames/s; a lone call is slower with it)
        auto_tuning["lane_merge"] = 1
        auto_tuning["hp_dedupe"] = 1   # IIR candidates in stages, duplicate runs removed (a third of the speculative steps)
    if workload == "c4" and n_local <= 128 and D >= 4 and G == 1:
        # small shards, many calls in flight: each call lays its speculative passes out for a quarter of the
        # GPU instead of as if it were alone (ofp_detect_tuning.concurrent_calls; results do not change)
        auto_tuning["concurrent_calls"] = 4

    def barrier():
        if not rehearsal:
            torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    def make_clips(slot):
        """[n_local, N, C] on the device.  c2: one DISTINCT clip per (rank, slot, clip): seed 1 + rank + 97*slot
        + 7919*clip.  c4: clip ids clip_lo.. from the recipe for slot 0; further slots (steps in flight) are
        distinct derived batches (channels rotated, gain changed) -- the host-side synthesis of another 512
        clips would take longer than the whole bench."""
        workers = max(1, usable_cpus() // (world if world <= 2 else 1))
        if workload == "c2":
            xs = synth_batch("c2", [1 + rank + 97 * slot + 7919 * i for i in range(n_local)], seconds, C, workers)
        else:
            xs = synth_batch("c4", [clip_lo + i for i in range(n_local)], seconds, C, workers)
        x = torch.from_numpy(np.stack(xs)).to(dev)
        return x.contiguous(), xs[0]

    def derive(slot0, slot):
        """c4, steps in flight: a distinct batch per slot derived on the device (channels rotated, gain changed)."""
        return (torch.roll(slot0[0], slot, dims=2) * (1.0 - 0.07 * slot)).contiguous(), None

    tuning = json.loads(args.tuning) if args.tuning else dict(auto_tuning)
    result_extra = {}
    if rehearsal:
        # fabricated detector output: 3 onsets per clip of this rank's shard
        def fake_step():
            rec = np.zeros((n_local, 4), dtype=np.dtype([("clip", np.int32), ("channel", np.int32), ("sample", np.int64)]))
            rec["clip"] = np.arange(n_local)[:, None]
            rec["channel"] = np.arange(4)[None, :] % C
            rec["sample"] = 1000 * (clip_lo + np.arange(n_local))[:, None] + np.arange(4)[None, :]
            r8 = torch.from_numpy(rec.view(np.uint8).reshape(n_local, 4, 16).copy())
            return pack_clips(r8, torch.full((n_local,), 3, dtype=torch.int64), cap_block, clip_offset=clip_lo)

        def run_steps(n, timed):
            g = None
            for _ in range(n):
                g = all_gather_blocks(fake_step())
            return g
    else:
        from onset_fingerprinting_amd.pipeline import FingerprintPipeline
        if workload == "c2":
            slots = [make_clips(s) for s in range(D)]
        else:
            slot0 = make_clips(0)
            if G == 1:
                slots = [slot0] + [derive(slot0, s) for s in range(1, D)]
            else:  # a call's slot holds the batches of its G steps back to back (all distinct)
                slots = [(torch.cat([slot0[0] if w * G + i == 0 else derive(slot0, w * G + i)[0] for i in range(G)]).contiguous(),
                          slot0[1] if w == 0 else None) for w in range(D)]
        exp = json.loads(args.exp) if args.exp else {}
        no_stft = bool(exp.pop("no_stft", False))
        pipes = [FingerprintPipeline(C, NFFT, HOP, SR, NMELS, device=local, cap_per_clip=cap_clip, **exp) for _ in range(D)]
        for pp in pipes:
            pp.skip_spectral = no_stft
        if args.exp:
            result_extra["experiment"] = args.exp
        if tuning:
            for pp in pipes:
                pp.detector.set_tuning(**tuning)
        streams = [torch.cuda.Stream(dev) for _ in range(D)]
        stage_acc, lat_acc, gather_acc = {}, [], []

        def run_step(w, timed):
            torch.cuda.set_device(local)
            t_in = time.perf_counter()
            with torch.cuda.stream(streams[w]):
                out = pipes[w].run(slots[w][0], timed=timed)
                # this rank's onset records as the fixed block the exchange uses (compacted on the device); with G steps
                # per call one block per step, from that step's clips
                if G == 1:
                    blk = pack_clips(out["records"], out["counts"], cap_block, clip_offset=clip_lo)
                else:
                    blk = [pack_clips(out["records"][i * n_local:(i + 1) * n_local], out["counts"][i * n_local:(i + 1) * n_local],
                                      cap_block, clip_offset=clip_lo) for i in range(G)]
            streams[w].synchronize()
            return out, blk, time.perf_counter() - t_in

        def finish(out, blk, lat, timed):
            t_g = time.perf_counter()
            if G == 1:
                gathered = all_gather_blocks(blk)  # ONE collective of fixed-size blocks, no host round trip
            else:
                gathered = None
                for b_i in blk:                    # one per step of the call, in step order
                    g_i = all_gather_blocks(b_i)
                    if gathered is None:           # (the checks below read the call's FIRST step: its clip 0 is out[...][0])
                        gathered = g_i
            if timed:
                gather_acc.append((time.perf_counter() - t_g) / G)
                st = dict(out["info"]["stage_ms"])
                st.pop("total")
                st.update(out["spectral_ms"])
                for k, v in st.items():
                    stage_acc[k] = stage_acc.get(k, 0.0) + v
                lat_acc.append(lat)
            return out, gathered

        if D == 1:
            def run_steps(n, timed):
                res = None
                for _ in range(-(-n // G)):
                    res = finish(*run_step(0, timed), timed)
                return res
        else:
            from concurrent.futures import ThreadPoolExecutor
            workers = [ThreadPoolExecutor(1) for _ in range(D)]  # worker w runs steps w, w+D, ... in order

            def run_steps(n, timed):
                n_calls = -(-n // G)   # (n is a multiple of G in the timed region)
                futs = [workers[i % D].submit(run_step, i % D, timed) for i in range(n_calls)]
                res = None
                for f in futs:  # complete (and exchange) in step order
                    res = finish(*f.result(), timed)
                return res

    # ---- W warm-up steps, then EXACTLY K timed steps between barriers; max over ranks
    run_steps(max(args.warmup, D * G if not rehearsal else 1), False)
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
    value = frames_total / (ms_per_step / 1e3)

    if rehearsal:
        gathered = unpack_gathered(res)
        recs = records_to_numpy(gathered)
        if rank == 0:
            print(json.dumps({
                "metric": "frames/sec (1024-pt, hop 256, 48 kHz) detect+FFT+classify at 1/2/4/8 MI355X",
                "value": 0.0, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
                "dtype": "f32", "data": "rehearsal: fabricated onset records, no GPU work (control flow only, not a measurement)",
                "config": {"workload": workload, "clips_total": total_clips, "clips_per_rank": n_local,
                           "ranks_in_exchange": int(res.shape[0]), "backend": backend,
                           "onsets_gathered": int(len(recs)), "clips_seen": int(len(np.unique(recs["clip"])))}}))
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- untimed extras on the same process: what one clip per step costs (c2) / the whole batch on one GPU (c4)
    extras = {}
    alone_ms = {}
    copy_gbs = None
    host_clip0 = None
    parity_out = None
    if not rehearsal:
        host_clip0 = slots[(args.steps // G - 1) % D][1]  # host copy of clip 0 of the batch the last timed step ran on (c4: slot 0 only)
        if rank == 0 and world == 1 and not args.no_cpu and res is not None:
            # clip 0 of the last TIMED step, taken to the host before anything else reuses the pipelines' output buffers
            o_t = res[0]
            parity_out = {k: o_t[k][0].cpu() for k in ("rel", "mel", "logits")}
        if rank == 0:
            # (a) what a plain device-to-device copy reaches on THIS GPU (2 GiB in, 2 GiB out: far beyond the caches):
            # the practical ceiling next to the 8 TB/s specification
            try:
                ca = torch.empty(1 << 29, dtype=torch.float32, device=dev)
                cb = torch.empty_like(ca)
                cb.copy_(ca)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(dev)
                e0.record()
                for _ in range(5):
                    cb.copy_(ca)
                e1.record()
                torch.cuda.synchronize(dev)
                copy_gbs = 5 * 2 * ca.numel() * 4 / (e0.elapsed_time(e1) / 1e3) / 1e9
                del ca, cb
            except Exception:
                copy_gbs = None
    if D > 1 and not rehearsal:
        # the SAME batch step with nothing else on the GPU (two repetitions after the timed region): what each launch
        # takes when it does not share the chip with the other steps in flight -- next to the in-flight durations
        barrier()
        for _ in range(2):
            o_a, _b, _l = run_step(0, True)
            st = dict(o_a["info"]["stage_ms"])
            st.pop("total")
            st.update(o_a["spectral_ms"])
            alone_ms = {k: float(v) for k, v in st.items()}
        barrier()
    if not args.no_extras:
        if workload == "c2" and (n_local > 1 or D > 1):
            p1 = FingerprintPipeline(C, NFFT, HOP, SR, NMELS, device=local, cap_per_clip=cap_clip)
            x1 = slots[0][0][:1].contiguous()
            s1 = torch.cuda.Stream(dev)
            with torch.cuda.stream(s1):
                p1.run(x1)
                s1.synchronize()
                barrier()
                n1 = 10
                t1 = time.perf_counter()
                for _ in range(n1):
                    o1 = p1.run(x1)
                    pack_clips(o1["records"], o1["counts"], 1024)
                    s1.synchronize()
                ms1 = (time.perf_counter() - t1) / n1 * 1e3
            extras["one_clip_per_step"] = {"ms_per_step": round(ms1, 3), "frames_per_s": round(C * H / (ms1 / 1e3)),
                                           "note": "the same path on ONE 8-channel clip per step (8 chains), steps one "
                                                   "after the other, this rank only"}
            del p1
        if world == 1 and D > 1:
            # (b) the same steps with the batch uploaded from pinned host memory before every step (the uploads of the
            # steps in flight share the PCIe link and overlap the other steps' compute): never `value`
            try:
                n_up = max(D, min(args.steps, 2 * D))
                pinned = slots[0][0].cpu().pin_memory()
                copy_streams = [torch.cuda.Stream(dev) for _ in range(D)]

                def step_with_upload(w):
                    torch.cuda.set_device(local)
                    with torch.cuda.stream(copy_streams[w]):
                        slots[w][0].copy_(pinned, non_blocking=True)
                    streams[w].wait_stream(copy_streams[w])
                    return run_step(w, False)

                for f in [workers[i % D].submit(step_with_upload, i % D) for i in range(D)]:
                    f.result()
                barrier()
                tu = time.perf_counter()
                for f in [workers[i % D].submit(step_with_upload, i % D) for i in range(n_up)]:
                    f.result()
                barrier()
                ms_up = (time.perf_counter() - tu) / n_up * 1e3
                t_c = time.perf_counter()
                for _ in range(3):
                    slots[0][0].copy_(pinned, non_blocking=True)
                torch.cuda.synchronize(dev)
                h2d_gbs = 3 * pinned.numel() * 4 / (time.perf_counter() - t_c) / 1e9
                extras["including_h2d"] = {"frames_per_s": round(frames_total / (ms_up / 1e3)), "ms_per_step": round(ms_up, 3),
                                           "h2d_GBps_alone": round(h2d_gbs, 1), "batch_MB": round(pinned.numel() * 4 / 1e6),
                                           "note": "the same steps in flight, each preceded by the upload of its batch from "
                                                   "pinned host memory (PCIe-bound: not the metric's value)"}
                del pinned
            except Exception as e:  # (an extra figure must not cost the line)
                extras["including_h2d"] = {"error": repr(e)[:200]}
        if world == 1 and workload == "c2":
            # (d) BASELINE configs[3] on this one GPU, same settings: the base a 1 -> N curve of the C4 workload needs in
            # the N = 1 record (the N > 1 lines run C4 sharded; `whole_batch_on_one_gpu` there is the same measurement)
            try:
                del pipes[:]
                slots.clear()
                torch.cuda.empty_cache()
                Dc, Cc, Nc = 6, C4["C"], int(C4["seconds"] * SR)
                xs4 = synth_batch("c4", list(range(C4["clips"])), C4["seconds"], Cc, max(1, usable_cpus()))
                x4 = torch.from_numpy(np.stack(xs4)).to(dev).contiguous()
                del xs4
                xs_slots = [x4] + [(torch.roll(x4, s, dims=2) * (1.0 - 0.07 * s)).contiguous() for s in range(1, Dc)]
                p4 = [FingerprintPipeline(Cc, NFFT, HOP, SR, NMELS, device=local, cap_per_clip=1024) for _ in range(Dc)]
                for pp in p4:
                    pp.detector.set_tuning(**(tuning or dict(lane_merge=1, hp_dedupe=1)))
                s4 = [torch.cuda.Stream(dev) for _ in range(Dc)]

                def c4_step(w):
                    torch.cuda.set_device(local)
                    with torch.cuda.stream(s4[w]):
                        o4 = p4[w].run(xs_slots[w])
                        pack_clips(o4["records"], o4["counts"], C4["clips"] * 256)
                    s4[w].synchronize()

                from concurrent.futures import ThreadPoolExecutor as _TPE4
                pool4 = [_TPE4(1) for _ in range(Dc)]
                for f in [pool4[i % Dc].submit(c4_step, i % Dc) for i in range(Dc)]:
                    f.result()
                n4 = 2 * Dc
                torch.cuda.synchronize(dev)
                t4 = time.perf_counter()
                for f in [pool4[i % Dc].submit(c4_step, i % Dc) for i in range(n4)]:
                    f.result()
                torch.cuda.synchronize(dev)
                ms4 = (time.perf_counter() - t4) / n4 * 1e3
                fr4 = C4["clips"] * Cc * synth.n_frames(Nc, NFFT, HOP)
                extras["c4_on_one_gpu"] = {"ms_per_step": round(ms4, 3), "frames_per_s": round(fr4 / (ms4 / 1e3)),
                                           "frames_per_step": fr4, "steps_in_flight": Dc, "steps": n4,
                                           "note": "BASELINE configs[3] (512 clips x 4 ch x 10 s) on this one GPU with the bench's "
                                                   "detector settings: the N = 1 point of the C4 strong-scaling curve"}
                for e4 in pool4:
                    e4.shutdown()
                del p4, xs_slots, x4
            except Exception as e:
                extras["c4_on_one_gpu"] = {"error": repr(e)[:200]}
        if workload == "c4" and world > 1 and rank == 0:
            # the base of the strong-scaling ratio: all 512 clips on ONE GPU with the same settings, steps in flight
            # as the single-GPU bench runs them (four deep here; rank 0's shard tiled to 512 clips: the same shapes
            # and amount of work as the whole batch)
            from concurrent.futures import ThreadPoolExecutor as _TPE
            Dw = 4
            xw = slots[0][0].repeat((total_clips + n_local - 1) // n_local, 1, 1)[:total_clips].contiguous()
            pws = [FingerprintPipeline(C, NFFT, HOP, SR, NMELS, device=local, cap_per_clip=cap_clip) for _ in range(Dw)]
            for pw in pws:
                pw.detector.set_tuning(lane_merge=1, hp_dedupe=1)
            sws = [torch.cuda.Stream(dev) for _ in range(Dw)]

            def whole_step(w):
                torch.cuda.set_device(local)
                with torch.cuda.stream(sws[w]):
                    ow = pws[w].run(xw)
                    pack_clips(ow["records"], ow["counts"], total_clips * 256)
                sws[w].synchronize()

            pool = [_TPE(1) for _ in range(Dw)]
            for f in [pool[i % Dw].submit(whole_step, i % Dw) for i in range(Dw)]:
                f.result()
            nw = 3 * Dw
            tw = time.perf_counter()
            for f in [pool[i % Dw].submit(whole_step, i % Dw) for i in range(nw)]:
                f.result()
            msw = (time.perf_counter() - tw) / nw * 1e3
            extras["whole_batch_on_one_gpu"] = {"ms_per_step": round(msw, 3), "frames_per_s": round(frames_total / (msw / 1e3)),
                                                "steps_in_flight": Dw, "strong_scaling_vs_it": round((msw / ms_per_step) / world, 3),
                                                "note": "rank 0 alone on 512 clips (its shard tiled), four steps in flight with the "
                                                        "bench's detector settings, after the timed region (the other ranks idle)"}
            for e in pool:
                e.shutdown()
            del pws, xw
        if world > 1:
            dist.barrier()

    if rank == 0:
        out, gathered = res
        ranks_seen = int(gathered.shape[0])   # blocks the all-gather delivered: one per rank
        gathered = unpack_gathered(gathered)  # decode (and check) the last step's exchange
        recs = records_to_numpy(gathered)
        stage_ms = {k: v / args.steps for k, v in stage_acc.items()}
        cand_ms = stage_ms.pop("hp_candidates", 0.0)  # a part of the hp stage, reported separately
        passes = out["info"]
        stage_bytes = dict(hp=BYTES_DETECT, db=BYTES_DETECT, ar=BYTES_DETECT, rel=BYTES_DETECT, mm=BYTES_DETECT // 2,
                           logic=BYTES_DETECT // 2, stft_mel=4 * HOP + BYTES_SPECTRUM + BYTES_FINGERPRINT, mlp=0)
        kernels = {"hp": "k_hp_candidates", "ar": "k_ar_sym_combine+k_ar_warm2|k_ar_warm_both+k_ar_chunk",
                   "mm": "k_mm_warm2+k_mm_chunk | k_mm_warm_both+k_mm_chunk_both", "db": "k_rect_db_sym (dB + the sums of the follower guess)",
                   "rel": "k_rel_out", "logic": "k_block_scan+k_last_clear+k_visit_count/scan/scatter+k_state_machine",
                   "stft_mel": "k_stft_power<1024, mlp> (mel + FCNN in the epilogue)", "mlp": "-"}
        # The dominant KERNEL = the single launch with the longest duration among those timed one by one (HIP events on
        # their streams): k_stft_power, k_rect_db, k_rel_out and the IIR candidate launch when it is ONE kernel
        # (k_hp_candidates, the latency layout).  In the batch layout the candidates are a family of short launches
        # (k_hp_seg0 / k_hp_dedupe / k_hp_seg / k_hp_seg_chunk): reported as a group under roofline_groups.
        staged = bool(passes.get("hp_chunk_runs"))
        singles = {k: stage_ms[k] for k in ("stft_mel", "db", "rel") if k in stage_ms}
        if cand_ms > 0 and not staged:
            singles["hp"] = cand_ms
        dom = max(singles, key=lambda k: singles[k])
        dom_ms, dom_bytes = singles[dom], (4 * HOP if dom == "hp" else stage_bytes[dom])
        achieved = dom_bytes * frames_local / (dom_ms / 1e3) / 1e9
        stage_ms["hp_candidates(part of hp)"] = cand_ms
        traffic, traffic_src = None, None
        try:
            pmc_clips = json.load(open(TRAFFIC_FILE.with_name("pmc_meta.json")))["clips"]
        except (OSError, KeyError, ValueError):
            pmc_clips = -1
        if TRAFFIC_FILE.exists() and workload == "c2" and n_local == pmc_clips:  # (counters per launch: only for the batch size they were taken at)
            tj = json.load(open(TRAFFIC_FILE))
            key = {"hp": "k_hp_candidates", "stft_mel": "k_stft_power", "db": "k_rect_db", "rel": "k_rel_out"}.get(dom)
            if key in tj:
                traffic = tj[key]["hbm_mb_per_launch"] * 1e6
                traffic_src = f"{TRAFFIC_FILE.relative_to(REPO)} (separate rocprofv3 --pmc passes of this command, not this run)"
        e2e_gbs = BYTES_E2E * frames_total / (ms_per_step / 1e3) / 1e9
        if workload == "c2":
            wl = (f"C2 x {n_local} per GPU and step: {n_local} distinct 8 ch x 60 s @ 48 kHz drum-hit clips side by side, "
                  "1024/256, detect + rFFT |X|^2 + 40-mel + FCNN(40-10-10-10-8); exchange of the onset records")
        else:
            wl = (f"C4: 512 clips x 4 ch x 10 s @ 48 kHz (Poisson hits) sharded {n_local} clips per GPU, 1024/256, "
                  "detect + rFFT |X|^2 + 40-mel + FCNN(40-10-10-10-8); RCCL all-gather of the onset records")
            if world == 1 and args.shard_of > 1:
                wl = (f"ONE RANK'S SHARE of C4 over {args.shard_of} GPUs ({n_local} of the 512 clips x 4 ch x 10 s), measured "
                      "alone on one GPU: value counts this shard only")
        result = {
            "metric": "frames/sec (1024-pt, hop 256, 48 kHz) detect+FFT+classify at 1/2/4/8 MI355X",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": dict({"workload": wl, "clips_per_gpu_per_step": n_local, "clips_total_per_step": total_clips,
                            "frames_per_step": frames_total, "frames_per_gpu_per_step": frames_local,
                            "onsets_gathered": int(len(recs)), "ranks_in_exchange": ranks_seen, "backend": backend if world > 1 else "-",
                            "parallelism": f"clips x{world}", "steps_in_flight_per_gpu": min(D * G, args.steps) if G > 1 else D, "steps_per_call": G, "calls_in_flight_per_gpu": D, "detector_tuning": tuning or "library defaults",
                            "latency_ms_per_step": round(1e3 * float(np.mean(lat_acc)), 3),
                            "exchange_call_ms_per_step": round(1e3 * float(np.mean(gather_acc)), 3)}, **extras, **result_extra),
            "roofline": {"bound": "hbm", "kernel": kernels[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "launches_per_step": 1, "avg_launch_ms": dom_ms, "algorithmic_bytes_per_frame": dom_bytes,
                         "peak_measured": ({"device_copy_GBps": round(copy_gbs, 1), "frac_of_it": achieved / copy_gbs,
                                            "note": "read + write rate of a 2 GiB device-to-device copy on this GPU in this run"}
                                           if copy_gbs else None),
                         "note": "per launch: algorithmic bytes of the launch / its duration (HIP events on its stream) "
                                 "while the other steps in flight share the GPU with it"},
            "roofline_e2e": {"bound": "hbm", "achieved": e2e_gbs, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                             "frac": e2e_gbs / (HBM_PEAK_GBS * world), "algorithmic_bytes_per_frame": BYTES_E2E,
                             "frac_of_measured_copy": (e2e_gbs / (copy_gbs * world) if copy_gbs else None),
                             "note": "whole step: 4 292 B per frame (samples in, rel + |X|^2 + mel + logits out) x frames / ms_per_step"},
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "detector_passes": {k: passes[k] for k in ("hp_passes", "ar_passes", "mm_passes", "repaired")},
        }
        if alone_ms.get(dom if dom != "hp" else "hp_candidates"):
            a_ms = alone_ms[dom if dom != "hp" else "hp_candidates"]
            a_gbs = dom_bytes * frames_local / (a_ms / 1e3) / 1e9
            result["roofline"]["alone"] = {"avg_launch_ms": a_ms, "achieved": a_gbs, "unit": "GB/s", "frac": a_gbs / HBM_PEAK_GBS,
                                           "note": "the same launch in the same batch step with no OTHER step on the GPU (one step "
                                                   "after the timed region; the launch still runs beside its own step's detector head)"}
            result["stage_ms_alone"] = {k: round(v, 4) for k, v in alone_ms.items()}
        if SQ_FILE.exists() and workload == "c2" and n_local == pmc_clips and "stft_mel" in stage_ms:
            # north_star: "MFMA used only for the dense classifier GEMM, evidenced by ... MFMA-busy against chip peak".  The
            # classifier runs in the epilogue of k_stft_power (19 v_mfma_f32_16x16x4_f32 per tile of 16 frames); instruction and
            # busy-cycle counts per launch from the PMC pass, the launch duration from this run
            sq = json.load(open(SQ_FILE)).get("k_stft_power", {})
            n_mfma, busy = sq.get("SQ_INSTS_MFMA"), sq.get("SQ_VALU_MFMA_BUSY_CYCLES")
            if n_mfma:
                flop = n_mfma * 2.0 * 16 * 16 * 4
                ms_l = stage_ms["stft_mel"]
                a_ms = alone_ms.get("stft_mel")
                result["roofline_mfma"] = {
                    "bound": "mfma", "kernel": kernels["stft_mel"], "dtype": "f32 (v_mfma_f32_16x16x4_f32)",
                    "mfma_instructions_per_launch": n_mfma, "flop_per_launch": flop,
                    "achieved": flop / (ms_l / 1e3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": flop / (ms_l / 1e3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                    "alone": ({"achieved": flop / (a_ms / 1e3) / 1e12, "frac": flop / (a_ms / 1e3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                               "mfma_busy_frac_of_simd_cycles": (busy / (a_ms / 1e3 * 2.4e9 * 1024) if busy else None)}
                              if a_ms else None),
                    "source": f"{SQ_FILE.relative_to(REPO)} (rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES, a separate pass of this "
                              "command) / the launch durations of this run",
                    "note": "the classifier is 1.4 kflop of the ~32 kflop a frame costs: the matrix cores are busy under 1 % of "
                            "the launch by construction (SURVEY.md 8d expects 'a small share of time')"}
        if cand_ms > 0 and passes.get("hp_candidate_steps"):
            flop = 17.0 * passes["hp_candidate_steps"]
            tf = flop / (cand_ms / 1e3) / 1e12
            issue = {"bound": "valu fp32 without fma", "flop_per_launch": flop, "achieved": tf,
                     "peak": VALU_NOFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / VALU_NOFMA_PEAK_TFLOPS}
            if dom == "hp":
                result["roofline"]["issue"] = issue
            else:
                gbs = 4 * HOP * frames_local / (cand_ms / 1e3) / 1e9
                result["roofline_groups"] = [{
                    "kernels": "k_hp_seg0 + k_hp_dedupe + k_hp_seg + k_hp_seg_chunk (the IIR stage's speculative candidates in stages, "
                               "duplicate runs removed between them)" if staged else "k_hp_candidates",
                    "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                    "ms": cand_ms, "algorithmic_bytes_per_frame": 4 * HOP,
                    "runs_that_walk_a_chunk": passes.get("hp_chunk_runs"), "issue": issue}]
        if world == 1 and not args.no_cpu:
            secs = min(args.cpu_seconds, seconds)
            n = int(secs * SR)
            sd = fcnn_state()
            last = (args.steps - 1) % D
            if host_clip0 is None:  # c4 with steps in flight: only slot 0 has a host copy
                last = 0
                out, _ = finish(*run_step(0, False), False)
                recs = records_to_numpy(unpack_gathered(_))
                parity_out = {k: out[k][0].cpu() for k in ("rel", "mel", "logits")}
            x0 = np.ascontiguousarray((host_clip0 if host_clip0 is not None else slots[last][1])[:n])   # clip 0 of the batch the checked step ran on
            cb = oracle_pass(x0, sd)
            r0 = recs[recs["clip"] == 0]
            k = r0["sample"] < (n // HOP) * HOP
            ok_idx = np.array_equal(r0["channel"][k], cb["ch"]) and np.array_equal(r0["sample"][k], cb["on"])
            nbs = (n // HOP) * HOP
            ok_rel = np.array_equal(parity_out["rel"][:nbs].numpy().view(np.uint32), cb["rel"].view(np.uint32))
            Hs = cb["mel"].shape[1]
            gm = parity_out["mel"][:, :Hs].numpy()
            big = cb["mel"] >= 1e-5 * cb["mel"].max()  # (the fp32 transform's error floor, tests/test_gpu_spectral.py)
            mel_err = float((np.abs(gm - cb["mel"])[big] / cb["mel"][big]).max())
            mel_err_norm = float(np.abs(gm - cb["mel"]).max() / cb["mel"].max())
            gl = parity_out["logits"][:, :Hs].numpy()
            log_err = float(np.abs(gl - cb["logits"]).max() / np.abs(cb["logits"]).max())
            fan = cpu_fanout(workload, min(secs, 20.0), sd)  # (bounded: every worker holds its clip's fp64 spectra)
            result["cpu_baseline"] = {"value": cb["frames"] / cb["seconds"], "unit": "frames/s", "cores": 1, "kind": "port",
                                      "cpu_model": cpu_model(),
                                      "sample": f"the first {secs:.0f} s of clip 0 of the timed batch ({cb['frames']} frames) "
                                                f"through oracle/ (C detector + numpy rFFT/mel/FCNN), {cb['seconds']:.2f} s wall",
                                      "fanout": {"value": fan["value"], "unit": "frames/s", "cores": fan["cores"],
                                                 "sample": fan["sample"]}}
            result["parity"] = {"onset_indices_exact": bool(ok_idx), "rel_bit_exact": bool(ok_rel),
                                "mel_max_rel_err_elementwise": mel_err, "mel_elementwise_floor": "bands >= 1e-5 of the largest",
                                "mel_max_err_over_max": mel_err_norm, "logits_max_rel_err": log_err,
                                "checked": "clip 0 of the last timed step against the oracle"}
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
