#!/usr/bin/env python3
"""Regenerates tests/golden/frontend_golden.npz from the CPU oracle on seeded synthetic inputs.

The reference holds no golden vectors for this path (its tests never reach the OpenCV calls),
so these vectors pin the oracle's own outputs: a later change to the oracle or the generators
that alters any of them fails tests/test_oracle_golden.py.  Inputs are regenerated from seeds;
only outputs are stored.  Run from the repository root after `make`.
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as ora  # noqa: E402
from vi_slam_amd import synth  # noqa: E402


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def build():
    g = {}
    w, h = 752, 480
    s0, s1 = synth.scene(synth.BASE_SEED, w, h, 0), synth.scene(synth.BASE_SEED, w, h, 1)
    nz = synth.noise(7, 320, 240)
    g["scene_crc"] = np.array([crc(s0), crc(s1), crc(nz), crc(synth.scene(synth.BASE_SEED + 1, 1920, 1080, 2))], np.uint32)
    pyr = ora.build_pyramid(s0, 21, 3)
    g["pyr_crc"] = np.array([crc(p) for p in pyr], np.uint32)
    g["pyr_l3"] = pyr[3]
    g["eig_crc"] = np.array([crc(ora.min_eigen_map(s0)), crc(ora.min_eigen_map(nz))], np.uint32)
    # cfg1 shape: vins_mono.yaml max_cnt 150 / min_dist 30; reference defaults 200 / 10 and Basalt's 0.007
    for tag, (n, q, d) in {"vins": (150, 0.01, 30), "openvins": (200, 0.01, 10), "basalt": (200, 0.007, 10)}.items():
        xy, sc = ora.good_features(s0, n, q, d)
        g[f"gftt_{tag}_xy"] = xy.astype(np.int16)
        g[f"gftt_{tag}_sc"] = sc.view(np.uint32)
    xy, _ = ora.good_features(s0, 150, 0.01, 30)
    nxt, st, err = ora.klt_track(s0, s1, xy, 21, 2)
    g["klt_next"] = nxt.view(np.uint32)
    g["klt_status"] = st
    g["klt_err"] = err.view(np.uint32)
    f = ora.fast9(s0, 20)
    g["fast_count_crc"] = np.array([len(f), crc(f)], np.uint32)
    g["fast_head"] = f[:64].astype(np.int16)
    fx, fs = ora.fast_grid(s0, 20, 5, 5, 200, 10)
    g["fast_grid_xy"] = fx.astype(np.int16)
    g["fast_grid_sc"] = fs.astype(np.int16)
    k, d = ora.orb_extract(s0, 500)
    g["orb_lxy_oct"] = np.stack([k["lx"], k["ly"], k["octave"]], axis=1).astype(np.int16)
    g["orb_angle"] = k["angle"].view(np.uint32)
    g["orb_resp"] = k["response"].view(np.uint32)
    g["orb_desc"] = d
    k1, d1 = ora.orb_extract(s1, 500)
    idx, dist = ora.match_knn2(d, d1)
    g["match_idx"] = idx.astype(np.int16)
    g["match_dist"] = dist.astype(np.int16)
    g["match_good"] = ora.ratio_test(idx, dist, 0.7)
    g["gauss7_kernel"] = ora.gauss7_kernel()
    g["orb_quotas_2000"] = ora.orb_quotas(2000)
    wh, sc = ora.orb_level_sizes(752, 480)
    g["orb_sizes_752"] = wh
    g["resize_crc"] = np.array([crc(ora.resize_linear_exact(s0, 627, 400)), crc(ora.gauss7(s0))], np.uint32)
    rng = np.random.default_rng(1)
    bgr = rng.integers(0, 256, (32, 48, 3), dtype=np.uint8)
    g["bgr2gray"] = ora.bgr2gray(bgr)
    g["equalize_crc"] = np.array([crc(ora.equalize_hist(s0))], np.uint32)
    # stages either side of the tracker
    g["clahe_crc"] = np.array([crc(ora.clahe(s0, 10.0, 8, 8)), crc(ora.clahe(s0[:477, :750], 2.0, 4, 6)), crc(ora.clahe(nz, 0.0, 8, 8))],
                              np.uint32)
    g["subpix"] = ora.corner_subpix(s0, xy, 5, -1, 30, 0.01).view(np.uint32)
    fwd, fst, _, back = ora.klt_track_bidir(s0, s1, xy, 0.04, 21, 2)
    g["bidir_status"] = fst
    g["bidir_back"] = back.view(np.uint32)
    cam = (458.654, 457.296, 367.215, 248.375)
    dist5 = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0)
    g["undistort"] = ora.undistort_points(xy, cam, dist5).view(np.uint32)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from two_view import two_view
    x1, x2, _, _ = two_view(300, 0.25, 0.3, 21)
    F, fmask, fcnt = ora.fundamental_ransac(x1, x2, None, 1.0, 0.99, 1000, seed=1234)
    g["ransac_F"] = F.view(np.uint64)
    g["ransac_mask"] = fmask
    g["ransac_count"] = np.array([fcnt], np.int32)
    return g


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "frontend_golden.npz")
    g = build()
    np.savez_compressed(out, **g)
    print(out, os.path.getsize(out), "bytes;", len(g), "arrays")
