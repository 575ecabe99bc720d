#!/usr/bin/env python3
"""Experiment: the front-end step at small stream counts (the streams64 share of a GPU at N = 8 is 8 streams), per-kernel
times from the library's HIP events.  Usage: python tools/exp_small_step.py [streams ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from vi_slam_amd import capi

dev = torch.device("cuda:0")
counts = [int(a) for a in sys.argv[1:]] or [1, 8]
host = bench.make_frames(0, max(counts), min(8, max(counts)), 4)
frames = []
for t in range(4):
    d = capi.alloc_image(max(counts), bench.H, bench.W, dev)
    d.copy_(torch.from_numpy(host[t]).to(dev))
    frames.append(d)
rk = bench.Ranks(1, 0, dev, False)
for nb in counts:
    L = bench.FrontendLeg(frames, nb, 0.05, dev)
    for overlap in (True, False):
        capi.set_overlap(overlap)
        el, prof = bench.timed(rk, L.step, 300, 20)
        k = {n: round(ms / c * 1e3, 1) for n, (ms, c) in prof.items()}
        print(f"streams {nb:3d} overlap {int(overlap)}: {el / 300 * 1e3:.4f} ms per step, {nb * 300 / el:.0f} frames/s; per-launch us {k}", flush=True)
