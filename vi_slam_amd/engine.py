"""Python view of the C++ host layer (libvi_slam_amd.so: SLAMEngine + framework adapters).

Mirrors the reference's facade (include/slam/slam_engine.hpp:71-242) for harnesses and tests;
images are numpy arrays standing in for cv::Mat.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvi_slam_amd.so")
_LIB = None

VINS_MONO, OPENVINS, ORB_SLAM3, BASALT = range(4)
UNINITIALIZED, INITIALIZING, TRACKING, LOST, RELOCALIZATION = range(5)


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: run `make`")
        L = C.CDLL(LIB_PATH)
        L.vis_engine_create.restype = C.c_void_p
        for name, args, res in [
            ("vis_engine_destroy", [C.c_void_p], None),
            ("vis_engine_select", [C.c_void_p, C.c_int], C.c_int),
            ("vis_engine_current", [C.c_void_p], C.c_int),
            ("vis_engine_initialize", [C.c_void_p, C.c_char_p], C.c_int),
            ("vis_engine_load_calibration", [C.c_void_p, C.c_char_p], C.c_int),
            ("vis_engine_process_image", [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int64], None),
            ("vis_engine_process_imu", [C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)], None),
            ("vis_engine_status", [C.c_void_p], C.c_int),
            ("vis_engine_pose", [C.c_void_p, C.POINTER(C.c_double)], C.c_int),
            ("vis_engine_tracked_count", [C.c_void_p], C.c_int),
            ("vis_engine_pose_callbacks", [C.c_void_p], C.c_long),
            ("vis_engine_status_callbacks", [C.c_void_p], C.c_long),
            ("vis_engine_last_status", [C.c_void_p], C.c_int),
            ("vis_engine_reset", [C.c_void_p], None),
            ("vis_engine_shutdown", [C.c_void_p], None),
            ("vis_engine_frontend_error", [C.c_void_p], C.c_char_p),
            ("vis_engine_points", [C.c_void_p, C.c_void_p, C.c_int], C.c_int),
            ("vis_engine_config_value", [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)], C.c_int),
            ("vis_engine_stats_callbacks", [C.c_void_p], C.c_long),
            ("vis_engine_enable_stage_timing", [C.c_void_p, C.c_int], None),
            ("vis_engine_stats", [C.c_void_p, C.POINTER(C.c_double)], C.c_int),
            ("vis_engine_stage_us", [C.c_void_p, C.POINTER(C.c_double)], C.c_int),
            ("vis_engine_stats_json", [C.c_void_p], C.c_char_p),
            ("vis_engine_reset_latency_stats", [C.c_void_p], None),
            ("vis_batch_open", [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_int),
                                C.c_int, C.c_int], C.c_void_p),
            ("vis_batch_close", [C.c_void_p], None),
            ("vis_batch_is_open", [C.c_void_p], C.c_int),
            ("vis_batch_shards", [C.c_void_p], C.c_int),
            ("vis_batch_error", [C.c_void_p], C.c_char_p),
            ("vis_batch_submit", [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_size_t, C.c_int64], C.c_int),
            ("vis_batch_staging", [C.c_void_p, C.c_int], C.c_void_p),
            ("vis_batch_staging_pitch", [C.c_void_p], C.c_size_t),
            ("vis_batch_submit_staged", [C.c_void_p, C.c_int64], C.c_int),
            ("vis_batch_wait", [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
            ("vis_batch_points", [C.c_void_p, C.c_int, C.c_void_p, C.c_int], C.c_int),
            ("vis_batch_reset", [C.c_void_p], C.c_int),
            ("vis_matcher_open", [C.POINTER(C.c_int), C.c_int], C.c_void_p),
            ("vis_matcher_close", [C.c_void_p], None),
            ("vis_matcher_is_open", [C.c_void_p], C.c_int),
            ("vis_matcher_world", [C.c_void_p], C.c_int),
            ("vis_matcher_error", [C.c_void_p], C.c_char_p),
            ("vis_matcher_collective", [C.c_void_p], C.c_char_p),
            ("vis_matcher_set_map", [C.c_void_p, C.c_void_p, C.c_longlong], C.c_int),
            ("vis_matcher_match", [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p], C.c_int),
            ("vis_matcher_last_ms", [C.c_void_p], C.c_double),
        ]:
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = res
        _LIB = L
    return _LIB


class SLAMEngine:
    def __init__(self):
        self._h = C.c_void_p(lib().vis_engine_create())

    def close(self):
        if self._h:
            lib().vis_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def select_framework(self, t):
        return bool(lib().vis_engine_select(self._h, t))

    def current_framework(self):
        return lib().vis_engine_current(self._h)

    def initialize(self, config_path):
        return bool(lib().vis_engine_initialize(self._h, str(config_path).encode()))

    def load_calibration(self, calib_path):
        return bool(lib().vis_engine_load_calibration(self._h, str(calib_path).encode()))

    def process_image(self, image, timestamp_ns):
        if image is None:
            lib().vis_engine_process_image(self._h, None, 0, 0, 0, 1, timestamp_ns)
            return
        a = np.ascontiguousarray(image, np.uint8)
        ch = 1 if a.ndim == 2 else a.shape[2]
        lib().vis_engine_process_image(self._h, a.ctypes.data, a.shape[0], a.shape[1], a.strides[0], ch, timestamp_ns)

    def process_imu(self, timestamp_ns, acc, gyro):
        a = (C.c_double * 3)(*acc)
        g = (C.c_double * 3)(*gyro)
        lib().vis_engine_process_imu(self._h, timestamp_ns, a, g)

    def status(self):
        return lib().vis_engine_status(self._h)

    def pose(self):
        out = (C.c_double * 8)()
        ok = lib().vis_engine_pose(self._h, out)
        return bool(ok), list(out)

    def tracked_count(self):
        return lib().vis_engine_tracked_count(self._h)

    def pose_callbacks(self):
        return lib().vis_engine_pose_callbacks(self._h)

    def status_callbacks(self):
        return lib().vis_engine_status_callbacks(self._h)

    def last_status(self):
        return lib().vis_engine_last_status(self._h)

    def reset(self):
        lib().vis_engine_reset(self._h)

    def shutdown(self):
        lib().vis_engine_shutdown(self._h)

    def frontend_error(self):
        return lib().vis_engine_frontend_error(self._h).decode()

    def points(self, cap=8192):
        out = np.zeros((cap, 2), np.float32)
        n = lib().vis_engine_points(self._h, out.ctypes.data, cap)
        return out[:n].copy()

    def stats(self):
        """Front-end statistics kept by the engine: dict with frames, failed, avg_latency_us, p99_latency_us and the last
        frame's ok / tracked / detected / total / latency_us (None before the first frame reached the front-end)."""
        out = (C.c_double * 9)()
        if not lib().vis_engine_stats(self._h, out):
            return None
        keys = ("frames", "failed", "avg_latency_us", "p99_latency_us", "ok", "tracked", "detected", "total", "latency_us")
        d = dict(zip(keys, list(out)))
        for k in ("frames", "failed", "tracked", "detected", "total"):
            d[k] = int(d[k])
        d["ok"] = bool(d["ok"])
        return d

    def stats_callbacks(self):
        return lib().vis_engine_stats_callbacks(self._h)

    def stats_json(self):
        return lib().vis_engine_stats_json(self._h).decode()

    def enable_stage_timing(self, on=True):
        lib().vis_engine_enable_stage_timing(self._h, 1 if on else 0)

    def stage_us(self):
        """GPU microseconds per stage of the last frame, {stage name: us}; empty unless stage timing is on."""
        from . import capi
        out = (C.c_double * 64)()
        if not lib().vis_engine_stage_us(self._h, out):
            return {}
        L = capi.lib()
        return {L.vfe_prof_kernel_name(i).decode(): out[i] for i in range(capi.prof_slots()) if out[i] > 0}

    def reset_latency_stats(self):
        lib().vis_engine_reset_latency_stats(self._h)

    def config_value(self, key):
        v = C.c_double(0)
        if not lib().vis_engine_config_value(self._h, key.encode(), C.byref(v)):
            raise KeyError(key)
        return v.value


def run_streaming(engine, frames, imu_per_frame=10, frame_dt_ns=100_000_000, t0_ns=2_000_000_000, warm_imu_s=1.5):
    """Counterpart of the reference's E2ETestFixture::runStreamingTest
    (tests/e2e/e2e_test_fixture.cpp:115-171): per frame, feed `imu_per_frame` IMU samples then the
    image, and record the wall-clock latency of that iteration in milliseconds.  IMU samples for
    `warm_imu_s` seconds are fed first so that the adapters' 1-second initialisation gate is open.
    Returns (latencies_ms, tracked_counts)."""
    import time
    t = t0_ns - int(warm_imu_s * 1e9)
    while t < t0_ns:
        engine.process_imu(t, (0.0, 0.0, 9.81), (0.0, 0.0, 0.0))
        t += 5_000_000
    lat, counts = [], []
    for i, f in enumerate(frames):
        ts = t0_ns + i * frame_dt_ns
        s = time.perf_counter()
        for k in range(imu_per_frame):
            engine.process_imu(ts - frame_dt_ns + (k + 1) * frame_dt_ns // imu_per_frame, (0.0, 0.0, 9.81), (0.0, 0.0, 0.01))
        engine.process_image(f, ts)
        lat.append((time.perf_counter() - s) * 1e3)
        counts.append(engine.tracked_count())
    return lat, counts


class BatchFrontEnd:
    """vi_slam::BatchFrontEnd (C++): `streams` independent live streams per call, split over `devices` in contiguous
    blocks, one host thread per block; frames are numpy uint8 arrays [H, W] (or one [streams, H, W] array)."""

    def __init__(self, width, height, streams, max_features=1000, levels=4, win=21, quality=0.01, min_distance=20.0,
                 devices=(), copy_threads=4):
        dev = (C.c_int * max(1, len(devices)))(*devices)
        self.width, self.height, self.streams, self.cap = width, height, streams, max_features
        self._h = C.c_void_p(lib().vis_batch_open(width, height, streams, max_features, levels, win, quality, min_distance,
                                                  dev, len(devices), copy_threads))
        if not lib().vis_batch_is_open(self._h):
            msg = lib().vis_batch_error(self._h).decode()
            self.close()
            raise RuntimeError(f"BatchFrontEnd: {msg}")

    def close(self):
        if self._h:
            lib().vis_batch_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def shards(self):
        return lib().vis_batch_shards(self._h)

    def submit(self, frames, ts=0):
        """frames: sequence of `streams` C-contiguous-row uint8 arrays of one row step; -> ticket."""
        step = frames[0].strides[0]
        ptrs = (C.c_void_p * self.streams)(*[f.ctypes.data for f in frames])
        t = lib().vis_batch_submit(self._h, ptrs, len(frames), step, ts)
        if t < 0:
            raise RuntimeError(lib().vis_batch_error(self._h).decode())
        return t

    def staging(self, stream):
        """numpy view [H, pitch] of the pinned staging memory of `stream` for the next submit_staged()."""
        pitch = lib().vis_batch_staging_pitch(self._h)
        p = lib().vis_batch_staging(self._h, stream)
        return np.ctypeslib.as_array((C.c_uint8 * (pitch * self.height)).from_address(p)).reshape(self.height, pitch)

    def submit_staged(self, ts=0):
        t = lib().vis_batch_submit_staged(self._h, ts)
        if t < 0:
            raise RuntimeError(lib().vis_batch_error(self._h).decode())
        return t

    def wait(self, ticket, points=True):
        """-> (totals [streams], tracked [streams], list of [n, 2] float32 arrays or None)."""
        tot = (C.c_int * self.streams)()
        trk = (C.c_int * self.streams)()
        if not lib().vis_batch_wait(self._h, ticket, tot, trk):
            raise RuntimeError(lib().vis_batch_error(self._h).decode())
        pts = None
        if points:
            pts = []
            for s in range(self.streams):
                a = np.empty((max(tot[s], 1), 2), np.float32)
                n = lib().vis_batch_points(self._h, s, a.ctypes.data, tot[s])
                pts.append(a[:n])
        return np.array(tot[:]), np.array(trk[:]), pts

    def reset(self):
        return bool(lib().vis_batch_reset(self._h))


class ShardedMatcher:
    """vi_slam::ShardedMatcher (C++): the map split over `devices` in contiguous row blocks, per-shard top-2 exchanged by
    RCCL all-gathers issued from C++, merged on every device."""

    def __init__(self, devices=()):
        dev = (C.c_int * max(1, len(devices)))(*devices)
        self._h = C.c_void_p(lib().vis_matcher_open(dev, len(devices)))
        if not lib().vis_matcher_is_open(self._h):
            msg = lib().vis_matcher_error(self._h).decode()
            self.close()
            raise RuntimeError(f"ShardedMatcher: {msg}")

    def close(self):
        if self._h:
            lib().vis_matcher_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def world(self):
        return lib().vis_matcher_world(self._h)

    @property
    def collective(self):
        return lib().vis_matcher_collective(self._h).decode()

    @property
    def last_ms(self):
        return lib().vis_matcher_last_ms(self._h)

    def set_map(self, desc):
        desc = np.ascontiguousarray(desc, np.uint8)
        if not lib().vis_matcher_set_map(self._h, desc.ctypes.data, len(desc)):
            raise RuntimeError(lib().vis_matcher_error(self._h).decode())

    def match(self, queries, ratio=0.7):
        q = np.ascontiguousarray(queries, np.uint8)
        nq = len(q)
        idx = np.empty((nq, 2), np.int32)
        dist = np.empty((nq, 2), np.int32)
        good = np.empty(nq, np.uint8)
        if not lib().vis_matcher_match(self._h, q.ctypes.data, nq, idx.ctypes.data, dist.ctypes.data, ratio, good.ctypes.data):
            raise RuntimeError(lib().vis_matcher_error(self._h).decode())
        return idx, dist, good.astype(bool)
