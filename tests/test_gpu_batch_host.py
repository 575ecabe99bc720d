"""The C++ batched front-end host (vi_slam::BatchFrontEnd) on a GPU: its own C++ program (batch against separate engines),
and from Python against the CPU oracle's composition of the same step."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as ora
from vi_slam_amd import engine as eng
from vi_slam_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_batch_frontend_cpp_program(cuda):
    exe = os.path.join(ROOT, "tests", "cpp", "bin", "test_batch_frontend")
    assert os.path.exists(exe), "run `make` first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "all checks passed" in r.stdout


@pytest.mark.parametrize("devices", [(), (0, 0, 0)])
def test_batch_frontend_equals_the_oracle_per_stream(cuda, devices):
    """Five streams (one shard, or three shards of 2 + 2 + 1 streams on device 0), four frames each, results collected one
    submission late: every stream's points equal the oracle's pyramid + LK + refill for that stream, bit for bit."""
    w, h, n, d, levels, S, T = 640, 400, 120, 16.0, 3, 5, 4
    seqs = [[synth.scene(synth.BASE_SEED + 300 + s, w, h, t) for t in range(T)] for s in range(S)]
    fe = eng.BatchFrontEnd(w, h, S, max_features=n, levels=levels, min_distance=d, devices=devices, copy_threads=2)
    assert fe.shards == max(1, len(devices))
    tickets = []
    got = {}
    for t in range(T):
        if t == 2:  # in place in the pinned staging memory
            for s in range(S):
                fe.staging(s)[:, :w] = seqs[s][t]
            tickets.append(fe.submit_staged(t))
        else:
            tickets.append(fe.submit([seqs[s][t] for s in range(S)], t))
        if t >= 1:
            got[t - 1] = fe.wait(tickets[t - 1])
    got[T - 1] = fe.wait(tickets[T - 1])
    fe.close()
    for s in range(S):
        ref, _ = ora.good_features(seqs[s][0], n, 0.01, d)
        tot, trk, pts = got[0]
        assert tot[s] == len(ref) and trk[s] == 0
        assert np.array_equal(pts[s].view(np.uint32), ref.view(np.uint32))
        for t in range(1, T):
            nxt, st, _ = ora.klt_track(seqs[s][t - 1], seqs[s][t], ref, 21, levels - 1)
            r = np.rint(nxt)
            keep = st.astype(bool) & (r[:, 0] >= 1) & (r[:, 0] < w - 1) & (r[:, 1] >= 1) & (r[:, 1] < h - 1)
            tracked = nxt[keep]
            new = np.zeros((0, 2), np.float32)
            if n - len(tracked) > 0:
                new, _ = ora.good_features(seqs[s][t], n - len(tracked), 0.01, d, seeds=tracked)
            ref = np.concatenate([tracked, new]) if len(new) else tracked
            tot, trk, pts = got[t]
            assert tot[s] == len(ref) and trk[s] == len(tracked), (s, t)
            assert np.array_equal(pts[s].view(np.uint32), ref.view(np.uint32)), (s, t)
