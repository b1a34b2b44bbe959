"""`bench.py --gpus N` must start its own ranks when it is not already under torchrun (the
driver calls it both ways).  CPU test of that plumbing with the stub workload
(QARIG_BENCH_STUB=1: gloo, no model, no GPU): N ranks rendezvous on 127.0.0.1, the time is the
max over ranks, rank 0 prints exactly one JSON line, a failing rank fails the whole run."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(extra_env, *args):
    env = dict(os.environ, QARIG_BENCH_STUB="1", OMP_NUM_THREADS="1", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args],
                          capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)


def test_bench_gpus2_self_launches_two_ranks_and_prints_one_line():
    r = _run({}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["stub"] is True and out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]


def test_bench_gpus1_runs_in_process():
    r = _run({}, "--gpus", "1", "--steps", "1")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 1


def test_bench_world_mismatch_fails_loudly():
    # under a torchrun-style environment that disagrees with --gpus the rank must abort
    env = {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}
    e = dict(os.environ, QARIG_BENCH_STUB="1", **env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                       capture_output=True, text=True, timeout=120, env=e, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
