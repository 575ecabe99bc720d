#!/usr/bin/env python3
"""Per-entry-point timings (HIP events inside the library) on a resident batch.
Usage: python3 tools/microbench.py [--batch 16] [--reps 5] [--only eig,klt,...]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--noise", action="store_true")
    args = ap.parse_args()
    only = set(args.only.split(",")) if args.only else None
    import torch
    from vi_slam_amd import capi, synth
    dev = torch.device("cuda:0")
    B, W, H, N, L = args.batch, 1920, 1080, 1000, 4
    if args.noise:
        seqs = [[synth.noise(100 + 2 * s + t, W, H) for t in range(2)] for s in range(2)]
    else:
        seqs = [[synth.scene(synth.BASE_SEED + s, W, H, t) for t in range(2)] for s in range(2)]
    frames = []
    for t in range(2):
        f = capi.alloc_image(B, H, W, dev)
        f.copy_(torch.from_numpy(np.stack([seqs[b % 2][t] for b in range(B)])).to(dev))
        frames.append(f)
    pyr = [capi.alloc_pyramid(frames[0], L), capi.alloc_pyramid(frames[1], L)]
    capi.build_pyramid(pyr[0]); capi.build_pyramid(pyr[1])
    gf = capi.GoodFeatures(B, H, W, N, device=dev)
    c, _, cnt = gf(frames[0], 0.01, 20.0)
    pts = c.clone()
    torch.cuda.synchronize()

    def run(name, fn):
        if only and name not in only:
            return
        fn()
        torch.cuda.synchronize()
        capi.prof_enable(True); capi.prof_read(True)
        for _ in range(args.reps):
            fn()
        torch.cuda.synchronize()
        capi.prof_enable(False)
        for k, (ms, n) in capi.prof_read(True).items():
            print(f"{name:10s} {k:14s} launches {n:3d} avg {ms / n * 1e3:9.1f} us  per-frame {ms / n * 1e3 / B:8.2f} us")

    l1 = pyr[0][1]
    run("pyr_l1", lambda: capi.pyr_down(frames[0], l1))
    run("pyramid", lambda: capi.build_pyramid(pyr[0]))
    run("eigmap", lambda: capi.min_eigen_map(frames[0]))
    run("gftt", lambda: gf(frames[0], 0.01, 20.0))
    run("klt", lambda: capi.klt_track(pyr[0], pyr[1], pts, cnt))
    print("counts", cnt.cpu().numpy()[:4])


if __name__ == "__main__":
    main()
