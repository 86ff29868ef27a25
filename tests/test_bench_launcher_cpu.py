"""bench.py's own N-rank launcher (no torchrun around it), on CPU: the rank plumbing -- fresh child per rank,
rendezvous on 127.0.0.1, barrier, max-over-ranks timing, one JSON line from rank 0 -- runs over gloo with the GPU step
stubbed out; and without the stub it must REFUSE to report an N-GPU number from fewer than N devices."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    return env


def test_self_launch_two_ranks_gloo_stub():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--cpu-stub", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one JSON line, got {lines}"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["rank_devices"] == [0, 1]
    assert out["stub"] is True and out["steps"] == 3 and out["warmup"] == 1


def test_refuses_more_ranks_than_devices():
    import torch
    n = torch.cuda.device_count() + 1 if torch.cuda.device_count() else 2
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode != 0
    assert r.stdout.strip() == "", "no result line may be printed"
    assert "refusing" in r.stderr


def test_world_size_must_match_gpus():
    env = dict(_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--cpu-stub"], capture_output=True, text=True, timeout=120,
                       env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
