"""bench.py's launch path without a GPU (missing 1): `python bench.py --gpus N` started by nobody's launcher must start its own N ranks and must never
come back with a number for fewer GPUs than it was asked for."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env.update(extra)
    return env


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out            # rank 0 prints ONE line
    return json.loads(lines[0])


def test_gpus_2_without_a_launcher_starts_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["local_ranks"] == [0, 1]
    assert d["launched_by"] == "torch.distributed.run"


def test_the_drivers_own_launcher_is_left_alone():
    # the way the driver starts N > 1: torch.distributed.run around bench.py -- no second launch inside
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29731",
                        BENCH, "--gpus", "2", "--dry-launch"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2


def test_a_launcher_that_disagrees_with_gpus_is_refused():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-launch"], env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "refusing" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_more_gpus_than_devices_is_refused_not_measured_on_fewer():
    # this container has no GPU at all: `--gpus 2` must say so instead of stepping whatever it finds
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two devices here")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "0"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
