#!/usr/bin/env python3
"""Front-end throughput benchmark: frames/s at 1920x1080 with 1000 features.

One "step" = one steady-state front-end pass over one batch of independent synthetic
1080p streams per GPU (new-frame pyramid + 4-level KLT of <=1000 points + Shi-Tomasi refill),
all inputs resident in HBM before the timed region.  Streams are sharded across ranks with
no data-path collective (weak scaling: `--batch` streams per GPU).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, NFEAT, LEVELS, WIN = 1920, 1080, 1000, 4, 21
QUALITY, MIN_DIST = 0.01, 20.0
HBM_PEAK_GBS = 8000.0

# algorithmic bytes per 1080p frame (each distinct input byte once + each output byte once)
BYTES_PYR = 2073600 + 518400 + 129600 + 32400              # read L0, write L1..L3
BYTES_KLT = 2 * BYTES_PYR + NFEAT * 8 + NFEAT * (8 + 1 + 4)  # two pyramids + point I/O
BYTES_DETECT = 2073600 + NFEAT * 8
ALGO_BYTES = {"k_pyr_down": BYTES_PYR, "k_klt": BYTES_KLT, "k_gftt_eig": 2073600, "k_gftt_select": NFEAT * 8,
              "k_compact": NFEAT * (8 + 1 + 8), "k_append": NFEAT * 16}


def make_frames(rank, batch, n_distinct, n_frames):
    """[n_frames][batch, H, W] uint8 host arrays; `n_distinct` seeded sequences tiled over the batch."""
    from vi_slam_amd import synth
    seqs = []
    for s in range(n_distinct):
        seed = synth.BASE_SEED + rank * 1000 + s
        seqs.append([synth.scene(seed, W, H, t) for t in range(n_frames)])
    return [np.stack([seqs[b % n_distinct][t] for b in range(batch)]) for t in range(n_frames)]


def cpu_baseline(frames0, loss, budget_s=12.0):
    """Reference-shaped CPU path (one frame at a time, one thread): LK between consecutive
    frames + refill detection, timed on the CPU oracle built -O3 -march=native."""
    from oracle import oracle as ora
    ora.build(fast=True)
    seq = frames0
    pts, _ = ora.good_features(seq[0], NFEAT, QUALITY, MIN_DIST, fast=True)
    done, t_used, i = 0, 0.0, 0
    order = list(range(1, len(seq))) + list(range(len(seq) - 2, -1, -1))
    prev = seq[0]
    while t_used < budget_s and done < 200:
        cur = seq[order[i % len(order)]]
        t0 = time.perf_counter()
        nxt, st, _ = ora.klt_track(prev, cur, pts, WIN, LEVELS - 1, fast=True, want_err=False)  # the step reads no residuals
        r = np.rint(nxt)
        keep = st.astype(bool) & (r[:, 0] >= 1) & (r[:, 0] < W - 1) & (r[:, 1] >= 1) & (r[:, 1] < H - 1)
        tracked = nxt[keep]
        if NFEAT - len(tracked) > 0:
            new, _ = ora.good_features(cur, NFEAT - len(tracked), QUALITY, MIN_DIST, seeds=tracked, fast=True)
            pts = np.concatenate([tracked, new]) if len(new) else tracked
        else:
            pts = tracked
        t_used += time.perf_counter() - t0
        pts = pts[:int(len(pts) * (1.0 - loss))]  # same track-loss model as the GPU leg
        prev = cur
        done += 1
        i += 1
    return {"value": done / t_used, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{done} consecutive 1920x1080 frames of one synthetic stream, {NFEAT} features, "
                      f"{LEVELS}-level LK (pyramids rebuilt per call as calcOpticalFlowPyrLK does) + refill detect"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="independent streams per GPU")
    ap.add_argument("--distinct", type=int, default=4, help="distinct synthetic sequences tiled over the batch")
    ap.add_argument("--frames", type=int, default=4, help="resident frames per stream (visited ping-pong)")
    ap.add_argument("--loss", type=float, default=0.05,
                    help="fraction of each stream's tracks dropped after every step (models feature loss, so that "
                         "every frame has corners to refill; the synthetic scene alone loses almost none)")
    ap.add_argument("--overlap", choices=["auto", "on", "off"], default="auto",
                    help="run the detector's eigenvalue pass beside pyramid + tracker on a helper stream: auto = the library's "
                         "rule (small batches only, so that per-kernel timings of the default run are not blurred by sharing "
                         "the chip), on / off = force")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--rehearse-one-device", action="store_true",
                    help="multi-rank rehearsal on a box with one GPU: every rank uses cuda:0 and the barrier / max-reduce "
                         "go over gloo (RCCL refuses two ranks on one device); not a measurement")
    args = ap.parse_args()
    if args.overlap != "auto":
        os.environ["VFE_OVERLAP"] = "1" if args.overlap == "on" else "0"  # read once by the library at its first step

    import torch
    import torch.distributed as dist
    from vi_slam_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the front-end has no CPU fallback")
    if args.rehearse_one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.rehearse_one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    capi.lib()

    B = args.batch
    host = make_frames(rank, B, min(args.distinct, B), args.frames)
    frames = []
    for t in range(args.frames):
        d = capi.alloc_image(B, H, W, dev)
        d.copy_(torch.from_numpy(host[t]).to(dev))
        frames.append(d)
    pyr = [capi.alloc_pyramid(frames[0], LEVELS), capi.alloc_pyramid(frames[0], LEVELS)]
    fe = capi.FrontendStep(B, H, W, NFEAT, LEVELS, WIN, quality=QUALITY, min_distance=MIN_DIST, device=dev)

    # first frame of every stream: pyramid + initial detection (untimed)
    cur = 0
    pyr[cur][0] = frames[0]
    capi.build_pyramid(pyr[cur])
    gf = capi.GoodFeatures(B, H, W, NFEAT, device=dev)
    c, _, cnt = gf(frames[0], QUALITY, MIN_DIST)
    fe.pts.copy_(c)
    fe.n_pts.copy_(cnt)
    del gf
    order = list(range(1, args.frames)) + list(range(args.frames - 2, -1, -1))
    pos = 0
    keep_den = 1000
    keep_num = int(round((1.0 - min(max(args.loss, 0.0), 1.0)) * keep_den))

    def one_step():
        nonlocal cur, pos
        nxt = cur ^ 1
        pyr[nxt][0] = frames[order[pos % len(order)]]
        fe.step(pyr[cur], pyr[nxt])
        if keep_num < keep_den:  # drop the newest tracks: n <- floor(n * keep)
            torch.mul(fe.n_pts, keep_num, out=fe.n_pts)
            torch.div(fe.n_pts, keep_den, rounding_mode="floor", out=fe.n_pts)
        cur = nxt
        pos += 1

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    capi.prof_enable(True)
    capi.prof_read(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    capi.prof_enable(False)
    prof = capi.prof_read(reset=True)
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_one_device else dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    n_tracked = fe.n_tracked.cpu().numpy()
    n_pts = fe.n_pts.cpu().numpy()

    if rank == 0:
        kernels = {}
        for name, (ms, n) in prof.items():
            per_launch_bytes = ALGO_BYTES.get(name, 0) * B * args.steps / max(n, 1)
            avg_ms = ms / n
            kernels[name] = {"launches": int(n), "avg_ms": avg_ms, "share": ms / (elapsed * 1e3),
                             "algo_GBs": per_launch_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else None}
        dom = max(kernels, key=lambda k: kernels[k]["share"])
        roof = {"kernel": dom, "bound": "hbm", "achieved": kernels[dom]["algo_GBs"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": kernels[dom]["algo_GBs"] / HBM_PEAK_GBS, "traffic": None,
                "algo_bytes_per_launch": ALGO_BYTES.get(dom, 0) * B}
        # HBM bytes per launch from the committed counter passes of this same command (tools/profile_round.sh);
        # only valid for the batch size they were taken at
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if tr.get("streams_per_gpu") == B and dom in tr["kernels"]:
                roof["traffic"] = tr["kernels"][dom]["total_bytes"]
                roof["traffic_source"] = "profiles/traffic.json (" + tr.get("tag", "") + ")"
            if dom in tr.get("limiter", {}):
                roof["limiter"] = tr["limiter"][dom]
        except (OSError, ValueError, KeyError):
            pass
        value = B * world * args.steps / elapsed
        out = {
            "metric": "front-end frames/s at 1920x1080, 1000 feats; % of HBM roofline",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/int32/f32", "data": "synthetic",
            "config": {"workload": "configs[1]: 1920x1080 streams, 1000 Shi-Tomasi corners + 4-level KLT (win 21) + refill, "
                                   f"{B} independent streams per GPU per step, {args.loss:.0%} of the tracks dropped per step", "streams_per_gpu": B,
                       "track_loss_per_step": args.loss, "overlap": args.overlap,
                       "frame_bytes": W * H, "algo_bytes_per_frame": BYTES_PYR + BYTES_KLT + BYTES_DETECT,
                       "parallelism": f"streams sharded over {world} GPU(s), no collective"},
            "frontend_frac_of_hbm_roofline": value / world * (BYTES_PYR + BYTES_KLT + BYTES_DETECT) / (HBM_PEAK_GBS * 1e9),
            "roofline": roof,
            "kernels": kernels,
            "tracks": {"mean_tracked": float(n_tracked.mean()), "mean_points": float(n_pts.mean())},
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline([host[t][0] for t in range(args.frames)], args.loss)
            out["speedup_vs_cpu_1core"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
