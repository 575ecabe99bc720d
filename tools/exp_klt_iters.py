#!/usr/bin/env python3
"""Experiment: tracker time against the iteration cap, on the benchmark's scene.  The increments show how many waves
still iterate at each count (a wave runs as long as the slowest of its four points)."""
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
o = np.lexsort((c[:, 0], c[:, 1] // 64))
d = torch.from_numpy(np.broadcast_to(c[o].astype(np.float32), (B,) + c.shape).copy()).to(dev)
prev = None
for it in (1, 2, 3, 4, 5, 6, 8, 12, 30):
    capi.klt_track(p0, p1, d, max_iter=it); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): capi.klt_track(p0, p1, d, max_iter=it)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 5 * 1e3
    print(f"max_iter {it:2d}: {t:.3f} ms" + ("" if prev is None else f"  (+{t - prev:.3f})"))
    prev = t
