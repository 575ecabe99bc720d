#!/usr/bin/env python3
"""Timings for the ORB and matching configurations on one GPU (wall clock around synchronised loops,
plus the per-kernel HIP-event times the library keeps).

  orb     8-level ORB, 2000 keypoints per frame, batch of 1080p frames
  f2f     frame-to-frame kNN-2 + ratio test on the extracted descriptors (2000 x 2000 per frame pair)
  reloc   2000 queries against one map shard (default 1.25M descriptors = 10M / 8 GPUs) + merge of 8
          shard lists

Usage: python3 tools/bench_orb_match.py [--batch 16] [--reps 5] [--shard 1250000] [--only orb,f2f,reloc]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--shard", type=int, default=1250000)
    ap.add_argument("--nq", type=int, default=2000)
    ap.add_argument("--only", default="")
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    args = ap.parse_args()
    only = set(args.only.split(",")) if args.only else None
    import torch
    from vi_slam_amd import capi, synth
    dev = torch.device("cuda:0")
    B, W, H = args.batch, args.w, args.h
    out = {}

    def timed(name, fn, reps, units, unit_name):
        fn()
        torch.cuda.synchronize()
        capi.prof_enable(True); capi.prof_read(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        capi.prof_enable(False)
        ker = {k: ms / reps for k, (ms, n) in capi.prof_read(True).items()}  # kernel ms per call of fn
        out[name] = {"ms": dt * 1e3, unit_name + "_per_s": units / dt, "kernel_ms": ker}
        print(f"{name:8s} {dt * 1e3:9.3f} ms  {units / dt:14.1f} {unit_name}/s  " +
              " ".join(f"{k}={v:.3f}" for k, v in ker.items()), flush=True)

    orb = None
    if not only or only & {"orb", "f2f"}:
        seqs = [[synth.scene(synth.BASE_SEED + s, W, H, t) for t in range(2)] for s in range(2)]
        frames = []
        for t in range(2):
            f = capi.alloc_image(B, H, W, dev)
            f.copy_(torch.from_numpy(np.stack([seqs[b % 2][t] for b in range(B)])).to(dev))
            frames.append(f)
        orb = [capi.Orb(B, H, W, 2000, device=dev), capi.Orb(B, H, W, 2000, device=dev)]
        timed("orb", lambda: orb[0](frames[0]), args.reps, B, "frames")
        orb[1](frames[1])
        torch.cuda.synchronize()
        print("keypoints/frame", orb[0].counts.cpu().numpy()[:4], flush=True)
    if orb is not None and (not only or "f2f" in only):
        n0 = orb[0].counts.cpu().numpy(); n1 = orb[1].counts.cpu().numpy()

        def f2f():
            capi.match_knn2_batch(orb[1].desc, orb[1].counts, orb[0].desc, orb[0].counts, 0.7)
        timed("f2f", f2f, args.reps, B, "frame-pairs")
    if not only or "reloc" in only:
        nt = args.shard
        g = torch.Generator(device="cpu"); g.manual_seed(1234)
        train = torch.randint(0, 256, (nt, 32), dtype=torch.uint8, generator=g).to(dev)
        q = torch.randint(0, 256, (args.nq, 32), dtype=torch.uint8, generator=g).to(dev)
        res = {}

        def reloc():
            res["r"] = capi.match_knn2(q, train)
        timed("reloc", reloc, args.reps, args.nq * nt, "pairs")
        idx, dist = res["r"]
        si = idx[None].repeat(8, 1, 1).contiguous(); sd = dist[None].repeat(8, 1, 1).contiguous()
        timed("merge8", lambda: capi.merge_knn2(si, sd), args.reps, args.nq, "queries")
        out["reloc"]["train_GBs"] = nt * 32 / (out["reloc"]["ms"] * 1e-3) / 1e9
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/bench_orb_match.json", "w"), indent=1)


if __name__ == "__main__":
    main()
