"""The C++ host layer on a GPU: the programs under tests/cpp that touch pixels (built in-tree by `make`) run here as
children of the test process -- the reference-style lifecycle programs again (this time their frames reach the kernels),
the hot switch between frameworks on a live engine, and the bounded soak loop."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(prog, timeout=180, env=None):
    exe = os.path.join(ROOT, "tests", "cpp", "bin", prog)
    assert os.path.exists(exe), "run `make` first"
    e = dict(os.environ)
    if env:
        e.update(env)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=timeout, env=e)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("prog", ["test_adapters_lifecycle", "test_slam_engine", "test_frontend_stats"])
def test_lifecycle_programs_with_a_gpu(cuda, prog):
    out = _run(prog)
    assert "passed" in out or "ok" in out


def test_pc_client_callback_example_with_a_gpu(cuda):
    out = _run("pc_client_video_callback")
    assert "and 60 frames (0 frames dropped before the engine)" in out
    assert "front-end: 60 frames" in out


def test_hot_switch_between_frameworks_on_a_live_engine(cuda):
    """selectFramework + initialize + calibration + first frames under the reference's 5 s target for each of the four
    frameworks, early return when the type is unchanged, stage timing per engine, gpu_device / enable_gpu keys."""
    out = _run("test_hot_switch")
    assert "hot switch: all checks passed" in out
    assert out.count("switch to ") == 4


def test_soak_loop_bounded(cuda):
    out = _run("test_soak", env={"VSA_SOAK_SECONDS": "6"})
    assert "soak: all checks passed" in out
