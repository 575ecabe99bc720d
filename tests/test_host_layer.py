"""CPU: the C++ host layer (SLAMEngine + adapters): the reference-style lifecycle programs,
the configuration parsers on the reference's own YAML keys, and the streaming harness up to the
point where pixels would be touched."""
import os
import subprocess

import numpy as np
import pytest

from vi_slam_amd import engine as eng

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("prog", ["test_adapters_lifecycle", "test_slam_engine", "test_ingest", "test_reference_shaped_adapter"])
def test_cpp_lifecycle_programs(prog):
    exe = os.path.join(ROOT, "tests", "cpp", "bin", prog)
    assert os.path.exists(exe), "run `make` first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "passed" in r.stdout


def test_frontend_stats_program():
    """FrontEndStats arithmetic (the reference's average / P99 window) and the engine's observer hook."""
    exe = os.path.join(ROOT, "tests", "cpp", "bin", "test_frontend_stats")
    assert os.path.exists(exe), "run `make` first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "frontend stats ok" in r.stdout


def test_pc_client_callback_example_runs():
    """The receiver callbacks of the reference's PC client written out: 60 fabricated frames and their IMU packets through
    StreamIngest into the engine; every frame is accounted for, with or without a GPU."""
    exe = os.path.join(ROOT, "tests", "cpp", "bin", "pc_client_video_callback")
    assert os.path.exists(exe), "run `make` first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "and 60 frames (0 frames dropped before the engine)" in r.stdout
    assert "front-end: 60 frames" in r.stdout and '"frontend":{"frame":60' in r.stdout


def write(tmp_path, name, text):
    p = tmp_path / name
    p.write_text(text)
    return p


VINS_YAML = """%YAML:1.0
image_width: 752
image_height: 480
max_cnt: 150
min_dist: 30
freq: 10
F_threshold: 1.0
equalize: 1
"""
OPENVINS_YAML = """num_pts: 200
use_klt: true
fast_threshold: 20
grid_x: 5
grid_y: 5
min_px_dist: 10
num_klt_levels: 3
"""
BASALT_YAML = """optical_flow:
  max_points: 200
  quality_threshold: 0.7
  pyramid_levels: 3
  patch_size: 21
  min_distance: 10
"""
ORB_YAML = """ORBextractor.nFeatures: 1200
ORBextractor.scaleFactor: 1.2
ORBextractor.nLevels: 8
ORBextractor.iniThFAST: 20
ORBextractor.minThFAST: 7
"""


def test_front_end_keys_of_the_reference_yaml_files(tmp_path):
    """Keys and values as in config/vins_mono.yaml:38-39,91-104, config/openvins.yaml:22-39,
    config/basalt.yaml:85-108, config/orbslam3.yaml:97-110."""
    e = eng.SLAMEngine()
    cases = [(eng.VINS_MONO, VINS_YAML, {"maxCnt": 150, "minDist": 30, "equalize": 1, "imageWidth": 752, "imageHeight": 480}),
             (eng.OPENVINS, OPENVINS_YAML, {"numFeatures": 200, "minPxDist": 10, "fastThreshold": 20, "gridX": 5, "gridY": 5,
                                            "useKLT": 1, "numKltLevels": 3, "imuInitWindow": 1.0}),
             (eng.BASALT, BASALT_YAML, {"maxFlowPoints": 200, "flowQualityThreshold": 0.7, "pyramidLevels": 3, "patchSize": 21,
                                        "minDistance": 10}),
             (eng.ORB_SLAM3, ORB_YAML, {"nFeatures": 1200, "scaleFactor": 1.2, "nLevels": 8, "iniThFAST": 20, "minThFAST": 7})]
    for t, text, expect in cases:
        assert e.select_framework(t) and e.current_framework() == t
        assert e.initialize(write(tmp_path, f"cfg{t}.yaml", text))
        for k, v in expect.items():
            assert e.config_value(k) == v, (t, k)
    e.close()


def test_openvins_tracking_front_end_is_opt_in(tmp_path):
    """`gpu_frontend: track` is this build's switch for OpenVINS's own scheme (KLT + FAST grid); without it the adapter
    keeps the reference's behaviour whatever use_klt says."""
    e = eng.SLAMEngine()
    assert e.select_framework(eng.OPENVINS) and e.initialize(write(tmp_path, "a.yaml", OPENVINS_YAML))
    assert e.config_value("trackFrontEnd") == 0 and e.config_value("useKLT") == 1
    assert e.select_framework(eng.OPENVINS) and e.initialize(write(tmp_path, "b.yaml", OPENVINS_YAML + "gpu_frontend: track\n"))
    assert e.config_value("trackFrontEnd") == 1
    e.close()


def test_engine_gates_like_the_reference(tmp_path):
    cfg = write(tmp_path, "ov.yaml", OPENVINS_YAML)
    cal = write(tmp_path, "calib.yaml", "cam0: {}\n")
    e = eng.SLAMEngine()
    frame = np.full((480, 640), 128, np.uint8)
    e.process_image(frame, 1)                        # no framework: ignored
    assert e.status() == eng.UNINITIALIZED
    assert e.select_framework(eng.OPENVINS)
    e.process_image(frame, 1)                        # not initialised: ignored, no callback
    assert e.status_callbacks() == 0
    assert not e.initialize(tmp_path / "missing.yaml")
    assert e.initialize(cfg) and e.status() == eng.INITIALIZING
    e.process_image(frame, 10**9)                    # calibration missing: adapter returns early
    assert e.tracked_count() == 0 and e.status_callbacks() == 1
    assert e.load_calibration(cal)
    e.process_imu(10**9, (0, 0, 9.81), (0, 0, 0))
    e.process_imu(10**9 + 1, (0, 0, float("nan")), (0, 0, 0))   # rejected
    e.process_imu(10**9 + 2, (0, 0, 9.81), (0, 0, 50.0))        # gyro out of range: rejected
    e.process_image(frame, 10**9 + 500_000_000)      # 0.5 s < imu_init_window: gate holds, no GPU touched
    assert e.tracked_count() == 0 and e.frontend_error() == ""
    ok, _ = e.pose()
    assert not ok and e.pose_callbacks() == 0
    e.reset()
    assert e.last_status() == eng.INITIALIZING
    e.shutdown()
    assert e.last_status() == eng.UNINITIALIZED and e.status() == eng.UNINITIALIZED
    e.close()


def test_first_real_frame_without_gpu_fails_loudly(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered by the GPU streaming tests")
    e = eng.SLAMEngine()
    assert e.select_framework(eng.VINS_MONO)
    assert e.initialize(write(tmp_path, "vm.yaml", VINS_YAML))
    e.process_image(np.full((480, 752), 128, np.uint8), 10**9)
    assert e.tracked_count() == 0
    assert "device" in e.frontend_error().lower() or "hip" in e.frontend_error().lower()
    e.close()


def test_batch_frontend_refuses_without_a_gpu():
    """vi_slam::BatchFrontEnd has no CPU fallback: on a machine without a GPU open() fails and says why."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("this machine has a GPU")
    with pytest.raises(RuntimeError, match="no usable GPU|no HIP device"):
        eng.BatchFrontEnd(640, 480, 4)
