#!/usr/bin/env python3
"""Timings of the steps either side of the tracker on one GPU (wall clock around synchronised loops):
histogram equalisation / CLAHE in front of it; sub-pixel refinement, point undistortion and epipolar outlier
rejection behind it.

Usage: python3 tools/bench_post.py [--batch 64] [--reps 5]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--pts", type=int, default=1000)
    args = ap.parse_args()
    import torch
    from vi_slam_amd import capi, synth
    from two_view import two_view
    dev = torch.device("cuda:0")
    B, W, H, N = args.batch, args.w, args.h, args.pts
    out = {}

    def timed(name, fn, units, unit_name):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        out[name] = {"ms": dt * 1e3, unit_name + "_per_s": units / dt}
        print(f"{name:22s} {dt * 1e3:9.3f} ms  {units / dt:14.1f} {unit_name}/s", flush=True)

    base = [synth.scene(synth.BASE_SEED + s, W, H, 0) for s in range(2)]
    frames = capi.alloc_image(B, H, W, dev)
    frames.copy_(torch.from_numpy(np.stack([base[b % 2] for b in range(B)])).to(dev))
    dst = capi.alloc_image(B, H, W, dev)
    timed("equalize_hist", lambda: capi.equalize_hist(frames, dst), B, "frames")
    timed("clahe_8x8_clip10", lambda: capi.clahe(frames, 10.0, 8, 8, out=dst), B, "frames")

    rng = np.random.default_rng(1)
    pts = np.stack([rng.random((N, 2)) * [W - 40, H - 40] + 20 for _ in range(B)]).astype(np.float32)
    d_pts0 = torch.from_numpy(pts).to(dev)
    d_pts = d_pts0.clone()

    def subpix():
        d_pts.copy_(d_pts0)
        capi.corner_subpix(frames, d_pts, None, 5, -1, 30, 0.01)
    timed("corner_subpix_win5", subpix, B * N, "points")
    cam = (458.654, 457.296, W / 2, H / 2)
    dist = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0)
    timed("undistort_points", lambda: capi.undistort_points(d_pts0, cam, dist), B * N, "points")

    x1, x2, _, _ = two_view(N, 0.2, 0.3, 3, W, H)
    n = len(x1)
    p1 = torch.from_numpy(np.broadcast_to(x1, (B, n, 2)).copy()).to(dev)
    p2 = torch.from_numpy(np.broadcast_to(x2, (B, n, 2)).copy()).to(dev)
    st = torch.ones((B, n), dtype=torch.uint8, device=dev)
    res = {}

    def ransac():
        st.fill_(1)
        res["r"] = capi.fundamental_ransac(p1, p2, st, None, 1.0, 0.99, 1000, seed=99)
    timed("fundamental_ransac", ransac, B, "frames")
    print("inliers/frame", res["r"][1][:4].cpu().numpy(), "of", n)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/bench_post.json", "w"), indent=1)


if __name__ == "__main__":
    main()
