#!/usr/bin/env python3
"""Single-stream latency of the host-pointer API (what the framework adapters call): one 1080p frame per
call, upload included.  Usage: python3 tools/bench_ctx_latency.py [--frames 60]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--n", type=int, default=1000)
    args = ap.parse_args()
    from vi_slam_amd import capi, synth
    frames = [synth.scene(synth.BASE_SEED, args.w, args.h, t) for t in range(8)]
    ctx = capi.Context(args.w, args.h, max_features=max(args.n, 2000), pyramid_levels=4, win_size=21, orb_levels=8)
    order = list(range(8)) + list(range(6, 0, -1))
    for name, fn in [("process_frame (KLT + refill)", lambda f, t: ctx.process_frame(f, t, args.n, 0.01, 20.0)),
                     ("detect_shi_tomasi", lambda f, t: ctx.detect_shi_tomasi(f, args.n, 0.01, 20.0)),
                     ("orb_extract 2000", lambda f, t: ctx.orb_extract(f, 2000))]:
        lat = []
        for i in range(args.frames):
            f = frames[order[i % len(order)]]
            t0 = time.perf_counter()
            fn(f, 10**9 + i * 33_000_000)
            lat.append((time.perf_counter() - t0) * 1e3)
        lat = np.array(lat[5:])
        print(f"{name:32s} median {np.median(lat):7.3f} ms  p95 {np.percentile(lat, 95):7.3f} ms  -> {1e3 / np.median(lat):7.0f} frames/s")
    ctx.close()


if __name__ == "__main__":
    main()
