"""bench.py's launcher logic on a machine without GPUs: `--gpus N` must start N ranks itself or fail loudly, never
run one rank and report n_gpus = 1."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    if env:
        e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=e,
                          timeout=300)


def test_gpus_flag_refuses_when_too_few_devices():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this machine has the GPUs: the refusal path does not apply")
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "visible" in r.stderr
    assert r.stdout.strip() == ""  # no JSON line pretending to be a result


def test_rank_refuses_world_size_mismatch():
    r = _run("--gpus", "1", "--steps", "1", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "--gpus 1" in (r.stderr + r.stdout)


def test_help_lists_the_contract_flags():
    r = _run("--help")
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--batch", "--legs"):
        assert flag in r.stdout
