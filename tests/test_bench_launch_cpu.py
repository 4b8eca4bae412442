"""bench.py --gpus N starts and verifies its own ranks (no GPU needed for the rendezvous: --launch-check, gloo).
Reference wiring this replaces: src/train.py:19-31,210-213 (torchrun-style environment, one process per GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _run(args, env_extra=None, timeout=240):
    env = dict(os.environ, SPLAT_BENCH_BACKEND="gloo", **(env_extra or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def _line(out):
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.timeout(300)
def test_plain_invocation_starts_its_own_ranks():
    r = _run(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["self_launched"] is True


@pytest.mark.timeout(300)
def test_single_rank_needs_no_launcher():
    r = _run(["--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r)
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and line["self_launched"] is False


@pytest.mark.timeout(300)
def test_world_size_must_equal_gpus():
    """under torchrun with a different rank count than --gpus the bench refuses instead of reporting a wrong n_gpus"""
    env = dict(os.environ, SPLAT_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--launch-check"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode != 0
    assert "runs inside a world of 2 ranks" in (r.stderr + r.stdout)
