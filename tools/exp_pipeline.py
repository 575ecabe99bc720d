#!/usr/bin/env python3
"""Experiment: the front-end step in two halves, the first half of frame t + 1 (pyramid, eigenvalue pass) on a side stream
beside the second half of frame t (tracker, selection), against the one-call step.  Checks that both give the same points.
Usage: python tools/exp_pipeline.py [streams ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vi_slam_amd import capi

dev = torch.device("cuda:0")
counts = [int(a) for a in sys.argv[1:]] or [8, 64, 256]
host = bench.make_frames(0, max(counts), min(16, max(counts)), 4)
frames = []
for t in range(4):
    d = capi.alloc_image(max(counts), bench.H, bench.W, dev)
    d.copy_(torch.from_numpy(host[t]).to(dev))
    frames.append(d)


class PipelinedLeg(bench.FrontendLeg):
    def __init__(self, frames, batch, loss, dev):
        super().__init__(frames, batch, loss, dev)
        pipe = capi.FrontendPipeline(batch, bench.H, bench.W, bench.NFEAT, bench.LEVELS, bench.WIN, quality=bench.QUALITY,
                                     min_distance=bench.MIN_DIST, device=dev)
        pipe.pts.copy_(self.fe.pts)
        pipe.n_pts.copy_(self.fe.n_pts)
        self.fe = pipe
        self.pyr.append(capi.alloc_pyramid(self.frames[0], bench.LEVELS))  # a third pyramid

    def frame_at(self, pos):
        return self.frames[self.order[pos % len(self.order)]]

    def step(self):
        nxt, coming = (self.cur + 1) % 3, (self.cur + 2) % 3
        self.pyr[nxt][0] = self.frame_at(self.pos)
        self.pyr[coming][0] = self.frame_at(self.pos + 1)
        self.fe.advance(self.pyr[self.cur], self.pyr[nxt], self.pyr[coming])
        if self.keep_num < self.keep_den:
            self.drop_tracks()
        self.cur = nxt
        self.pos += 1


for nb in counts:
    res = {}
    for name, cls in (("one call", bench.FrontendLeg), ("two halves", PipelinedLeg)):
        L = cls(frames, nb, 0.05, dev)
        period = len(L.order)
        for _ in range(2 * period):
            L.step()
        torch.cuda.synchronize()
        reps = max(10, 600 // period)
        t0 = time.perf_counter()
        for _ in range(reps * period):
            L.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (reps * period)
        res[name] = (dt, L.fe.pts.clone(), L.fe.n_pts.clone())
        del L
    same = torch.equal(res["one call"][2], res["two halves"][2]) and torch.equal(res["one call"][1], res["two halves"][1])
    a, b = res["one call"][0], res["two halves"][0]
    print(f"streams {nb:3d}: one call {a * 1e3:.4f} ms per step ({nb / a:.0f} frames/s), two halves {b * 1e3:.4f} ms "
          f"({nb / b:.0f} frames/s, {100 * (a / b - 1):+.1f} %), same points {same}", flush=True)
