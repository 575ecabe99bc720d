#!/usr/bin/env python3
"""Experiment: the front-end step replayed from a HIP graph (six steps = one period of the resident-frame cycle captured
once) against the same steps launched one by one.  Usage: python tools/exp_graph_step.py [streams ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vi_slam_amd import capi

dev = torch.device("cuda:0")
counts = [int(a) for a in sys.argv[1:]] or [1, 8, 64]
host = bench.make_frames(0, max(counts), min(8, max(counts)), 4)
frames = []
for t in range(4):
    d = capi.alloc_image(max(counts), bench.H, bench.W, dev)
    d.copy_(torch.from_numpy(host[t]).to(dev))
    frames.append(d)
for nb in counts:
    L = bench.FrontendLeg(frames, nb, 0.05, dev)
    period = len(L.order)
    for _ in range(3 * period):
        L.step()
    torch.cuda.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps * period):
        L.step()
    torch.cuda.synchronize()
    plain = (time.perf_counter() - t0) / (reps * period)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(period):
            L.step()  # the capture stream's first steps outside a capture (the library's helper stream is created here)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(period):
            L.step()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / (reps * period)
    print(f"streams {nb:3d}: launched {plain * 1e3:.4f} ms per step ({nb / plain:.0f} frames/s), from a graph {graph * 1e3:.4f} ms per step ({nb / graph:.0f} frames/s)", flush=True)
