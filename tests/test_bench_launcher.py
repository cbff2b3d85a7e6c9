"""`python bench.py --gpus N` launched BARE (no torchrun, no WORLD_SIZE) must either produce N ranks or fail
loudly -- never a one-GPU line labelled N (sbdart_amd/launch.py).  CPU: the launcher path
over gloo (--rendezvous-only gloo stops after the ranks have counted each other)."""
import json
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e["OMP_NUM_THREADS"] = "1"
    return e


def test_bare_gpus_2_brings_up_two_ranks_over_gloo():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-only", "gloo"], env=_env(),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                       # rank 0 only
    assert lines[0] == {"rendezvous_only": True, "n_gpus": 2, "backend": "gloo", "ranks_counted": 2}


def test_bare_gpus_2_without_two_gpus_fails_loudly():
    """This container has no GPU at all (and the GPU box has one): --gpus 2 must not print a bench line."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("a node with two GPUs runs the real thing")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_env(),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "refusing" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith('{"metric')]


def test_world_that_is_not_the_asked_one_is_refused():
    """A launcher that started 2 ranks for `--gpus 4` (or 1 for 2): every rank refuses."""
    from sbdart_amd.launch import launcher_command
    cmd = launcher_command(2, BENCH, ["--gpus", "4", "--rendezvous-only", "gloo"])
    out = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "refusing to mislabel" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
