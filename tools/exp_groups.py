#!/usr/bin/env python3
"""Experiment: the 256 streams of the headline step as G independent groups, each on a HIP stream of its own, so that the
tail of one group's kernel is filled by the other groups' work.  Usage: python3 tools/exp_groups.py [--batch 256] [--steps 100]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--groups", default="1,2,4")
    args = ap.parse_args()
    import torch
    from vi_slam_amd import capi
    dev = torch.device("cuda:0")
    host = bench.make_frames(0, args.batch, 16, 4)
    frames = []
    for t in range(4):
        d = capi.alloc_image(args.batch, bench.H, bench.W, dev)
        d.copy_(torch.from_numpy(host[t]).to(dev))
        frames.append(d)
    for overlap in (True, False):
        capi.set_overlap(overlap)
        for G in [int(g) for g in args.groups.split(",")]:
            per = args.batch // G
            legs, streams = [], []
            for g in range(G):
                s = torch.cuda.Stream(device=dev)
                with torch.cuda.stream(s):
                    legs.append(bench.FrontendLeg([f[g * per:(g + 1) * per] for f in frames], per, 0.05, dev))
                streams.append(s)
            torch.cuda.synchronize()

            def step():
                for L, s in zip(legs, streams):
                    with torch.cuda.stream(s):
                        L.step()
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"overlap {int(overlap)} groups {G} x {per} streams: {dt / args.steps * 1e3:.3f} ms per step, {args.batch * args.steps / dt:,.0f} frames/s", flush=True)
            del legs


if __name__ == "__main__":
    main()
