#!/usr/bin/env python3
"""Experiment: one tracker dispatch per configuration (iteration cap x pyramid levels), to be run under
`rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES`: the per-dispatch instruction counts split the kernel's work into its
per-level constant part (template, solve, patch traffic) and its per-iteration part without touching the kernel."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vi_slam_amd import capi, synth
from oracle import oracle as ora

dev = torch.device("cuda:0")
B, W, H = 16, 1920, 1080
seq = [synth.scene(synth.BASE_SEED, W, H, t) for t in range(2)]
f0 = capi.alloc_image(B, H, W, dev); f1 = capi.alloc_image(B, H, W, dev)
f0.copy_(torch.from_numpy(np.broadcast_to(seq[0], (B, H, W)).copy()).to(dev))
f1.copy_(torch.from_numpy(np.broadcast_to(seq[1], (B, H, W)).copy()).to(dev))
p0 = capi.build_pyramid(capi.alloc_pyramid(f0, 4)); p1 = capi.build_pyramid(capi.alloc_pyramid(f1, 4))
c, _ = ora.good_features(seq[0], 1000, 0.01, 20)
o = np.lexsort((c[:, 0], c[:, 1] // 64))
d = torch.from_numpy(np.broadcast_to(c[o].astype(np.float32), (B,) + c.shape).copy()).to(dev)
for levels in (4, 1):
    for it in (1, 2, 3, 30):
        capi.klt_track(p0[:levels], p1[:levels], d, max_iter=it)
        torch.cuda.synchronize()
        print("dispatched levels", levels, "max_iter", it, flush=True)
