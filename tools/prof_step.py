#!/usr/bin/env python3
"""Workload for rocprofv3: a few steady-state front-end steps (and optional ORB/match passes)
on a small batch, so that per-kernel traces and PMC passes finish in seconds.
Usage: rocprofv3 ... -- python3 tools/prof_step.py [--batch 16] [--steps 3]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--n", type=int, default=1000)
    args = ap.parse_args()
    import torch
    from vi_slam_amd import capi, synth
    dev = torch.device("cuda:0")
    B, W, H, N, L = args.batch, args.w, args.h, args.n, 4
    seqs = [[synth.scene(synth.BASE_SEED + s, W, H, t) for t in range(2)] for s in range(2)]
    frames = []
    for t in range(2):
        f = capi.alloc_image(B, H, W, dev)
        f.copy_(torch.from_numpy(np.stack([seqs[b % 2][t] for b in range(B)])).to(dev))
        frames.append(f)
    pyr = [capi.alloc_pyramid(frames[0], L), capi.alloc_pyramid(frames[1], L)]
    capi.build_pyramid(pyr[0])
    fe = capi.FrontendStep(B, H, W, N, L, 21, quality=0.01, min_distance=20.0, device=dev)
    gf = capi.GoodFeatures(B, H, W, N, device=dev)
    c, _, cnt = gf(frames[0], 0.01, 20.0)
    fe.pts.copy_(c)
    fe.n_pts.copy_(cnt)
    cur = 0
    for _ in range(args.steps):
        nxt = cur ^ 1
        fe.step(pyr[cur], pyr[nxt])
        cur = nxt
    torch.cuda.synchronize()
    print("done", fe.n_pts.cpu().numpy()[:4])


if __name__ == "__main__":
    main()
