#!/usr/bin/env python3
"""Experiment: what would grouping the four points of a wave by iteration count be worth?  The largest iteration count
a point needs at any level is found by raising the cap until its result stops changing; the tracker is then timed
with the points in tile order, sorted by that count, and sorted by count within coarse tiles."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vi_slam_amd import capi, synth
from oracle import oracle as ora

dev = torch.device("cuda:0")
B, W, H = 64, 1920, 1080
seq = [synth.scene(synth.BASE_SEED, W, H, t) for t in range(2)]
f0 = capi.alloc_image(B, H, W, dev); f1 = capi.alloc_image(B, H, W, dev)
f0.copy_(torch.from_numpy(np.broadcast_to(seq[0], (B, H, W)).copy()).to(dev))
f1.copy_(torch.from_numpy(np.broadcast_to(seq[1], (B, H, W)).copy()).to(dev))
p0 = capi.build_pyramid(capi.alloc_pyramid(f0, 4)); p1 = capi.build_pyramid(capi.alloc_pyramid(f1, 4))
c, _ = ora.good_features(seq[0], 1000, 0.01, 20)
c = c[np.lexsort((c[:, 0], c[:, 1] // 64))].astype(np.float32)

def track(pts, it=30):
    d = torch.from_numpy(np.broadcast_to(pts, (B,) + pts.shape).copy()).to(dev)
    out, st, _ = capi.klt_track(p0, p1, d, max_iter=it)
    return out[0].cpu().numpy(), st[0].cpu().numpy()

ref, _ = track(c)
need = np.full(len(c), 30)
for it in range(12, 0, -1):
    o, _ = track(c, it)
    same = (o.view(np.uint32) == ref.view(np.uint32)).all(axis=1)
    need[same] = it
print("iteration cap needed per point: histogram", np.bincount(need)[:14], "mean %.2f" % need.mean())
g = need[: len(c) // 4 * 4].reshape(-1, 4)
print("tile order: mean of wave maxima %.2f" % g.max(axis=1).mean())

def run(pts, tag):
    d = torch.from_numpy(np.broadcast_to(pts, (B,) + pts.shape).copy()).to(dev)
    capi.klt_track(p0, p1, d); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): capi.klt_track(p0, p1, d)
    torch.cuda.synchronize()
    print(f"{tag:36s} {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms")
run(c, "tile order")
run(c[np.argsort(need, kind="stable")], "sorted by needed cap")
tile = (c[:, 1] // 270).astype(int) * 8 + (c[:, 0] // 480).astype(int)
run(c[np.lexsort((need, tile))], "by cap within 480x270 tiles")
