#!/usr/bin/env python3
"""Experiment: how much tracker time is lost because the four points of a wave advance in lockstep?  Tracks 1000 distinct
points per frame, then 250 distinct points each repeated four times in a row (the four points of a wave are then the
same point, so no lane group waits for another), and the distinct points sorted by position."""
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
def run(pts, tag):
    d = torch.from_numpy(np.broadcast_to(pts.astype(np.float32), (B,) + pts.shape).copy()).to(dev)
    capi.klt_track(p0, p1, d); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): capi.klt_track(p0, p1, d)
    torch.cuda.synchronize()
    print(f"{tag:28s} {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms")
run(c, "1000 distinct (score order)")
run(np.repeat(c[:250], 4, axis=0), "250 x 4 repeated")
run(np.repeat(c[250:500], 4, axis=0), "next 250 x 4 repeated")
run(np.repeat(c[500:750], 4, axis=0), "3rd 250 x 4 repeated")
run(np.repeat(c[750:], 4, axis=0), "4th 250 x 4 repeated")
o = np.lexsort((c[:, 0], c[:, 1] // 64))
run(c[o], "1000 distinct (raster order)")
