#!/usr/bin/env python3
"""Experiment: the streams of a GPU stepped as G independent groups, each on a HIP stream of its own, against one batch
(a group's chain of dependent launches leaves tails that another group's kernels can fill).
Usage: python tools/exp_groups.py [streams] [groups ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vi_slam_amd import capi

dev = torch.device("cuda:0")
total = int(sys.argv[1]) if len(sys.argv) > 1 else 64
groups = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
host = bench.make_frames(0, total, min(16, total), 4)
frames = []
for t in range(4):
    d = capi.alloc_image(total, bench.H, bench.W, dev)
    d.copy_(torch.from_numpy(host[t]).to(dev))
    frames.append(d)
for G in groups:
    nb = total // G
    streams = [torch.cuda.Stream(device=dev) for _ in range(G)]
    legs = []
    for g in range(G):
        with torch.cuda.stream(streams[g]):
            legs.append(bench.FrontendLeg([f[g * nb:(g + 1) * nb] for f in frames], nb, 0.05, dev))
    torch.cuda.synchronize()

    def run(steps):
        for _ in range(steps):
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    legs[g].step()
    run(12)
    torch.cuda.synchronize()
    steps = 300
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    pts = sum(float(L.fe.n_pts.float().mean()) for L in legs) / G
    print(f"{total} streams as {G} group(s) of {nb}: {dt * 1e3:.4f} ms per step of all streams, {total / dt:.0f} frames/s "
          f"(mean points {pts:.0f})", flush=True)
    del legs
