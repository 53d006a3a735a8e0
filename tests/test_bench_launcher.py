"""CPU-only: `bench.py --gpus N` is its own launcher.  The rehearsal
mode runs the launcher, the rendezvous, the clip sharding, the one-collective exchange, the max-over-ranks
timing and the JSON line over gloo with fabricated onset records -- no GPU work, and the line says so."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
BENCH = str(REPO / "bench.py")


def _run(cmd, env=None, timeout=240):
    e = dict(os.environ)
    e.pop("RANK", None), e.pop("WORLD_SIZE", None), e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=timeout, cwd=str(REPO))


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # ONE line on stdout
    return json.loads(lines[0])


def test_gpus_2_launches_two_ranks_and_prints_one_line():
    r = _run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearsal"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1
    assert j["scaling"] == "strong" and j["config"]["workload"] == "c4"
    c = j["config"]
    assert c["ranks_in_exchange"] == 2 and c["clips_per_rank"] == 256 and c["clips_total"] == 512
    assert c["clips_seen"] == 512 and c["onsets_gathered"] == 3 * 512  # every clip of both shards arrived
    assert "rehearsal" in j["data"] and j["value"] == 0.0            # not presented as a measurement


def test_three_ranks_shard_512_clips_unevenly():
    r = _run([sys.executable, BENCH, "--gpus", "3", "--steps", "2", "--warmup", "1", "--rehearsal"])
    assert r.returncode == 0, r.stderr[-2000:]
    c = _json_line(r.stdout)["config"]
    assert c["ranks_in_exchange"] == 3 and c["clips_seen"] == 512 and c["clips_per_rank"] == 171


def test_under_torch_distributed_run_the_ranks_are_used_as_they_are():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
              "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1",
              "--rehearsal", "--workload", "c2", "--clips", "3"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["clips_seen"] == 6


def test_gpus_must_equal_world_size():
    r = _run([sys.executable, BENCH, "--gpus", "1", "--rehearsal"], env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE is 2" in r.stderr and r.stdout.strip() == ""


def test_a_failed_rank_fails_the_launcher():
    r = _run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--rehearsal"],
             env={"OFP_BENCH_TEST_FAIL_RANK": "1"})
    assert r.returncode != 0 and r.stdout.strip() == "" and "ranks failed" in r.stderr
