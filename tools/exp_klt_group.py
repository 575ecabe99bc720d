#!/usr/bin/env python3
"""Experiment: what grouping the tracker's points by their iteration counts could be worth.
  mode collect : track the benchmark's 64 x 1000 points once with VFE_KLT_DEBUG_ITERS=1 and save every point's total
                 iteration count (gpurun_out/klt_iters.npy)
  mode time    : time vfe_dev_klt_track_ws (no err output) with the library's own order, and -- run with
                 VFE_KLT_ORDER_GIVEN=1 -- with orders built here from the saved counts
Usage: python tools/exp_klt_group.py collect|time [streams]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from vi_slam_amd import capi

mode = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
host = bench.make_frames(0, B, min(16, B), 3)
frames = []
for t in range(3):
    d = capi.alloc_image(B, bench.H, bench.W, dev)
    d.copy_(torch.from_numpy(host[t]).to(dev))
    frames.append(d)
# a steady state: two steps of the front-end, then the tracker alone on frames 1 -> 2 with the points the step holds
L = bench.FrontendLeg(frames, B, 0.05, dev)
L.step()
L.step()
torch.cuda.synchronize()
prev, nxt = L.pyr[L.cur ^ 1], L.pyr[L.cur]
pts, n = L.fe.pts.clone(), L.fe.n_pts.clone()
out = torch.zeros_like(pts)
st = torch.zeros(pts.shape[:2], dtype=torch.uint8, device=dev)
err = torch.zeros(pts.shape[:2], dtype=torch.float32, device=dev)
ws = torch.zeros(capi.lib().vfe_klt_track_workspace(pts.shape[1], B), dtype=torch.uint8, device=dev)
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "klt_iters.npy")


def timeit(label, reps=20):
    capi.klt_track_ws(prev, nxt, pts, n, out, st, None, win=bench.WIN, workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        capi.klt_track_ws(prev, nxt, pts, n, out, st, None, win=bench.WIN, workspace=ws)
    e1.record()
    torch.cuda.synchronize()
    print(f"{label:46s} {e0.elapsed_time(e1) / reps:.4f} ms", flush=True)


if mode == "collect":
    capi.klt_track_ws(prev, nxt, pts, n, out, st, err, win=bench.WIN, workspace=ws)
    torch.cuda.synchronize()
    it = err.cpu().numpy()
    np.save(path, it)
    nn = n.cpu().numpy()
    v = np.concatenate([it[b, :nn[b]] for b in range(B)])
    print("iterations per point: mean %.2f  median %.0f  p90 %.0f  p99 %.0f  max %.0f" % (v.mean(), np.median(v), np.percentile(v, 90), np.percentile(v, 99), v.max()))
    w4 = np.concatenate([it[b, :nn[b] // 4 * 4].reshape(-1, 4) for b in range(B)])
    print("caller order, waves of four: mean of max %.2f (mean %.2f)" % (w4.max(axis=1).mean(), w4.mean()))
else:
    it = np.load(path)
    nn = n.cpu().numpy()
    cap = pts.shape[1]
    given = os.environ.get("VFE_KLT_ORDER_GIVEN") == "1"
    if not given:
        timeit("library order (rim class, tile)")
    else:
        p = pts.cpu().numpy()
        W, H = bench.W, bench.H

        def rim_class(x, y):
            rim = np.minimum(np.minimum(x, W - 1 - x), np.minimum(y, H - 1 - y))
            cls = np.zeros(len(x), np.int64)
            for c in range(4):
                cls += (rim >= (11 << c)).astype(np.int64) * (cls == c)
            return cls

        def set_order(keyfn, label):
            o = np.zeros((B, cap), np.uint16)
            mx = []
            for b in range(B):
                k = nn[b]
                x = np.clip(p[b, :k, 0].astype(np.int64), 0, W - 1); y = np.clip(p[b, :k, 1].astype(np.int64), 0, H - 1)
                idx = keyfn(x, y, it[b, :k])
                o[b, :k] = idx
                g = it[b, idx][: k // 4 * 4].reshape(-1, 4)
                mx.append(g.max(axis=1).mean())
            ws.view(torch.int16)[: B * cap].copy_(torch.from_numpy(o.view(np.int16).reshape(-1)).to(dev))
            timeit(f"{label} (mean of max {np.mean(mx):.2f})")

        tile = lambda x, y: (y >> 6) * 31 + (x >> 6)
        set_order(lambda x, y, i: np.lexsort((tile(x, y), rim_class(x, y))), "rim class, tile (numpy)")
        set_order(lambda x, y, i: np.lexsort((tile(x, y), -i, rim_class(x, y))), "rim class, iterations desc, tile")
        set_order(lambda x, y, i: np.lexsort((tile(x, y), -np.minimum(i // 4, 6), rim_class(x, y))), "rim class, iteration bucket /4 desc, tile")
        set_order(lambda x, y, i: np.lexsort((-i, tile(x, y), rim_class(x, y))), "rim class, tile, iterations desc within tile")
        set_order(lambda x, y, i: np.lexsort((tile(x, y), -i)), "iterations desc, tile")
