#!/usr/bin/env python3
"""Front-end throughput benchmark: frames/s at 1920x1080 with 1000 features.

    python bench.py --gpus N --steps K --warmup W

One "step" of the headline leg = one steady-state front-end pass over one batch of independent synthetic 1080p
streams per GPU (new-frame pyramid + 4-level KLT of <= 1000 points + Shi-Tomasi refill), all inputs resident in HBM
before the timed region.  Streams are sharded across ranks with no data-path collective (weak scaling: `--batch`
streams per GPU).  W untimed steps, then exactly K timed steps between barrier + synchronize pairs, the slowest rank
counts.  The other configurations of the same front-end are measured in the same run and reported as sub-objects of
the one JSON line rank 0 prints:

    streams64      64 independent 1080p streams in total, 64 / N per GPU (strong scaling of the headline step)
    orb_match      8-level ORB, 2000 keypoints per frame, + frame-to-frame kNN-2 / ratio test, `--orb-batch` streams per GPU
    reloc          2000 query descriptors against a 10 M-descriptor map sharded over the ranks, per-shard top-2
                   all-gathered (RCCL) and merged; planted near-duplicates are verified
    single_stream  one live stream through the host-pointer API the framework adapters call (upload included),
                   at 752x480 / 150 corners / 3 levels and at 1920x1080 / 1000 corners / 4 levels
    cpu_baseline   the CPU oracle on the host cores of the same box: one thread (the reference's one-frame-at-a-time
                   harness shape) and all cores (one independent stream per thread); N = 1 only

With --gpus N > 1 and no WORLD_SIZE in the environment the script starts its N ranks itself (a child
`python -m torch.distributed.run`, started before anything in this process touches a GPU); under a launcher it is a rank.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, NFEAT, LEVELS, WIN = 1920, 1080, 1000, 4, 21
QUALITY, MIN_DIST = 0.01, 20.0
HBM_PEAK_GBS = 8000.0
ORB_NFEAT, ORB_LEVELS, ORB_SCALE, ORB_FAST_THR, KNN_RATIO = 2000, 8, 1.2, 20, 0.7
MAP_ROWS, N_QUERIES = 10_000_000, 2000
# VALU issue ceiling measured on this chip (tools/valu_ceiling.hip, profiles/r02_valu_ceiling.txt): packed 16-bit, dot2/dot4,
# DPP, perm, med3, popcount, min/max and shifts occupy a SIMD's issue port for ~4.2 cycles per wave64 instruction at any
# occupancy; only v_add/v_fma f32, v_add_u32, v_xor, v_mov reach ~2.2 with two or more waves per SIMD.
VALU_CYCLES_PER_INST, SIMDS, CLOCK_GHZ = 4.2, 1024, 2.4
# average issue cost of a kernel's instruction mix where it is known to differ (none at present): 4.2 is used for every
# instruction, which overstates the time at the ceiling (and so the fraction of it reached) by whatever share of a kernel's
# instructions belongs to the 2.2-cycle classes.
VALU_CYCLES_BY_KERNEL = {}
MFMA_FP4_PEAK_TFLOPS = 10000.0   # dense block-scaled fp4 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4), the matcher's instruction
MATCH_FLOP_PER_PAIR = 512        # one descriptor pair = a 256-element product of +-1 vectors

# algorithmic bytes per 1080p frame (each distinct input byte once + each API-visible output byte once)
BYTES_PYR = 2073600 + 518400 + 129600 + 32400               # read L0, write L1..L3
BYTES_KLT = 2 * BYTES_PYR + NFEAT * 8 + NFEAT * (8 + 1 + 4)  # two pyramids + point I/O
BYTES_DETECT = 2073600 + NFEAT * 8
BYTES_FRAME = BYTES_PYR + BYTES_KLT + BYTES_DETECT
ALGO_BYTES = {"k_pyr_down": BYTES_PYR, "k_klt": BYTES_KLT, "k_gftt_eig": 2073600, "k_gftt_select": NFEAT * 8,
              "k_compact": NFEAT * (8 + 1 + 8), "k_append": NFEAT * 16}


def log(msg):
    """Progress on stderr (stdout carries the one JSON line)."""
    print(f"[bench r{os.environ.get('RANK', '0')} {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------------ launching ranks

def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n):
    """Start n ranks of this script under torch.distributed.run and wait for them.  Runs before this process has
    touched a GPU (device_count() only asks the driver how many there are)."""
    import torch
    have = torch.cuda.device_count()
    if have < n:
        print(f"bench.py: --gpus {n} asked for, {have} GPU(s) visible", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------ synthetic inputs

def make_frames(rank, batch, n_distinct, n_frames):
    """[n_frames][batch, H, W] uint8 host arrays; `n_distinct` seeded sequences tiled over the batch."""
    from concurrent.futures import ThreadPoolExecutor
    from vi_slam_amd import synth
    def one_sequence(s):  # a thread renders all frames of a sequence: they share the generator's cached world
        return [synth.scene(synth.BASE_SEED + rank * 1000 + s, W, H, t) for t in range(n_frames)]
    with ThreadPoolExecutor(max_workers=host_cores()) as ex:  # the generator is native code and releases the GIL
        seqs = list(ex.map(one_sequence, range(n_distinct)))
    return [np.stack([seqs[b % n_distinct][t] for b in range(batch)]) for t in range(n_frames)]


def host_cores():
    """Cores this process may use: its affinity mask, capped by the box's thread budget when one is exported."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cap = os.environ.get("OMP_NUM_THREADS")
    if cap and cap.isdigit() and int(cap) > 0:
        n = min(n, int(cap))
    return max(1, n)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# ------------------------------------------------------------------------------------------------ CPU baseline

def cpu_stream(seq, loss, budget_s, max_frames, fast=True, opt=False):
    """The reference-shaped CPU path for one stream (one frame at a time): LK between consecutive frames + refill
    detection on the CPU oracle.  Returns (frames done, seconds inside the front-end calls)."""
    from oracle import oracle as ora
    pts, _ = ora.good_features(seq[0], NFEAT, QUALITY, MIN_DIST, fast=fast, opt=opt)
    done, t_used, i = 0, 0.0, 0
    order = list(range(1, len(seq))) + list(range(len(seq) - 2, -1, -1))
    prev = seq[0]
    while t_used < budget_s and done < max_frames:
        cur = seq[order[i % len(order)]]
        t0 = time.perf_counter()
        nxt, st, _ = ora.klt_track(prev, cur, pts, WIN, LEVELS - 1, fast=fast, want_err=False, opt=opt)  # the step reads no residuals
        r = np.rint(nxt)
        keep = st.astype(bool) & (r[:, 0] >= 1) & (r[:, 0] < W - 1) & (r[:, 1] >= 1) & (r[:, 1] < H - 1)
        tracked = nxt[keep]
        if NFEAT - len(tracked) > 0:
            new, _ = ora.good_features(cur, NFEAT - len(tracked), QUALITY, MIN_DIST, seeds=tracked, fast=fast, opt=opt)
            pts = np.concatenate([tracked, new]) if len(new) else tracked
        else:
            pts = tracked
        t_used += time.perf_counter() - t0
        pts = pts[:int(len(pts) * (1.0 - loss))]  # same track-loss model as the GPU leg
        prev = cur
        done += 1
        i += 1
    return done, t_used


def cpu_baseline(host, loss, n_distinct, budget_s=7.0):
    """The CPU oracle (-O3 -march=native, built on this machine) on the same workload, in two organisations of the same
    arithmetic with bit-identical results: `port_fast` (oracle/ora_opt.cpp: one Scharr derivative image per pyramid level,
    border-extended images, row-streaming eigenvalue map -- how cv::calcOpticalFlowPyrLK / cv::cornerMinEigenVal organise
    the work) and `port` (the plain restatement the parity tests check against).  Each (i) on one thread over consecutive
    frames of one stream -- the shape of the reference's E2E harness -- and (ii) with one independent stream per host
    core.  `value` is the faster organisation on all cores."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as ora
    ora.build(fast=True)
    ora.lib(fast=True)
    n_frames = len(host)
    cores = host_cores()
    seqs = [[host[t][c % n_distinct] for t in range(n_frames)] for c in range(cores)]

    def run(opt, budget):
        one_done, one_t = cpu_stream([host[t][0] for t in range(n_frames)], loss, budget * 0.6, 200, opt=opt)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=cores) as ex:  # the oracle is native code behind ctypes: the GIL is released
            res = list(ex.map(lambda s: cpu_stream(s, loss, budget, 200, opt=opt), seqs))
        wall = time.perf_counter() - t0
        all_done = sum(d for d, _ in res)
        return {"value": all_done / wall, "unit": "frames/s", "cores": cores,
                "one_core": {"value": one_done / one_t, "unit": "frames/s", "cores": 1,
                             "sample": f"{one_done} consecutive 1920x1080 frames of one stream, one thread"},
                "sample": f"{all_done} 1920x1080 frames: {cores} independent synthetic streams, one per core, each frame by frame "
                          f"({NFEAT} features, {LEVELS}-level LK with pyramids rebuilt per call as calcOpticalFlowPyrLK does, refill "
                          f"detect), {wall:.1f} s wall"}

    fastp = run(True, budget_s)
    plain = run(False, budget_s * 0.6)
    out = {"value": fastp["value"], "unit": "frames/s", "cores": cores, "kind": "port",
           "what": "this build's CPU restatement of the OpenCV calls (oracle/, -O3 -march=native), NOT OpenCV itself: "
                   "OpenCV is not installed on this box.  value = port_fast, the organisation an optimised CPU library uses "
                   "(scalar code, no hand-written SIMD); port = the plain restatement",
           "cpu_model": cpu_model(), "one_core": fastp["one_core"], "sample": fastp["sample"],
           "port_fast": fastp, "port": plain}
    return out


# ------------------------------------------------------------------------------------------------ timing helpers

class Ranks:
    def __init__(self, world, rank, dev, gloo):
        self.world, self.rank, self.dev, self.gloo = world, rank, dev, gloo

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()

    def max_over_ranks(self, x):
        if self.world == 1:
            return x
        import torch
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if self.gloo else self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        if self.world == 1:
            return x
        import torch
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if self.gloo else self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())


def timed(rk, fn, steps, warmup, events=True):
    """`warmup` untimed + exactly `steps` timed calls of fn between barrier + synchronize pairs; -> (seconds, the
    slowest rank's; per-kernel HIP-event times of this rank inside the timed region).  `events=False`: without the library's
    HIP events around every kernel (two event records per launch cost a short step tens of microseconds) -- for regions
    whose kernel table comes from a pass of its own."""
    import torch
    from vi_slam_amd import capi
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    rk.barrier()
    capi.prof_enable(events)
    capi.prof_read(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    rk.barrier()
    elapsed = time.perf_counter() - t0
    capi.prof_enable(False)
    prof = capi.prof_read(reset=True)
    return rk.max_over_ranks(elapsed), prof


def kernel_table(prof, elapsed, bytes_per_step, steps):
    """Per kernel slot: launches, average launch duration (HIP events on the launch stream), share of the timed region,
    algorithmic GB/s = algorithmic bytes per launch / average duration."""
    out = {}
    for name, (ms, n) in prof.items():
        avg_ms = ms / n
        per_launch = bytes_per_step.get(name, 0) * steps / max(n, 1)
        out[name] = {"launches": int(n), "avg_ms": avg_ms, "share": ms / (elapsed * 1e3),
                     "algo_bytes_per_launch": per_launch,
                     "algo_GBs": per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 and per_launch else None}
    return out


def load_counters():
    """Counter-derived figures of the committed rocprofv3 passes (tools/profile_round.sh): HBM bytes and VALU instructions
    per launch, valid for the batch size they were taken at."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return {}


def roofline_for(kernels, counters, leg, batch, flop_per_step=None, steps=1):
    """`roofline` object for the kernel with the largest share of the leg's timed region.  `flop_per_step`: arithmetic per
    step of the kernel slots that run on the matrix cores; such a slot is priced against the MFMA peak."""
    scored = {k: v for k, v in kernels.items() if v["algo_GBs"]}
    if not scored:
        return None
    dom = max(scored, key=lambda k: scored[k]["share"])
    k = scored[dom]
    if flop_per_step and dom in flop_per_step:
        sec_per_step = k["avg_ms"] * 1e-3 * k["launches"] / steps
        ach = flop_per_step[dom] / sec_per_step / 1e12
        roof = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": MFMA_FP4_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / MFMA_FP4_PEAK_TFLOPS, "traffic": None, "algo_flop_per_step": flop_per_step[dom],
                "launches_per_step": k["launches"] / steps, "avg_launch_ms": k["avg_ms"],
                "algo_GBs": k["algo_GBs"],
                "note": "fp4 operands (+1 / -1 per descriptor bit), fp32 accumulation: 512 flop per descriptor pair"}
        c = counters.get("legs", {}).get(leg, {})
        if c.get("batch") == batch and dom in c.get("kernels", {}):
            roof["traffic"] = c["kernels"][dom].get("total_bytes")
            roof["traffic_source"] = f"profiles/traffic.json ({counters.get('tag', '')})"
            if c["kernels"][dom].get("mfma_busy_cycles"):
                # SQ_VALU_MFMA_BUSY_CYCLES summed over the SIMDs against the SIMD cycles of the launch at the nominal clock
                roof["mfma_busy_frac"] = c["kernels"][dom]["mfma_busy_cycles"] / (SIMDS * CLOCK_GHZ * 1e9 * k["avg_ms"] * 1e-3)
        return roof
    roof = {"kernel": dom, "bound": "hbm", "achieved": k["algo_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": k["algo_GBs"] / HBM_PEAK_GBS, "traffic": None, "algo_bytes_per_launch": k["algo_bytes_per_launch"],
            "avg_launch_ms": k["avg_ms"]}
    c = counters.get("legs", {}).get(leg, {})
    if c.get("batch") == batch and dom in c.get("kernels", {}):
        ck = c["kernels"][dom]
        roof["traffic"] = ck.get("total_bytes")
        roof["traffic_source"] = f"profiles/traffic.json ({counters.get('tag', '')})"
        if ck.get("valu_insts"):
            # instructions x measured issue cost against the SIMD cycles of the launch
            rate = ck["valu_insts"] / (k["avg_ms"] * 1e-3)
            cyc = VALU_CYCLES_BY_KERNEL.get(dom, VALU_CYCLES_PER_INST)
            peak = SIMDS * CLOCK_GHZ * 1e9 / cyc
            roof["valu"] = {"achieved": rate / 1e9, "peak": peak / 1e9, "unit": "G wave-instructions/s", "frac": rate / peak,
                            "insts_per_launch": ck["valu_insts"], "cycles_per_inst": cyc,
                            "ceiling_source": "profiles/r02_valu_ceiling.txt"}
            if rate / peak > 0.6 and roof["frac"] < 0.5:
                roof["bound"] = "valu"
                roof["note"] = ("achieved / peak / frac are against the HBM roof the metric names; what binds this kernel is "
                                "VALU instruction issue, priced under 'valu'")
    return roof


# ------------------------------------------------------------------------------------------------ legs

class FrontendLeg:
    """Steady-state detect + track step over `batch` resident streams (the headline step and streams64)."""

    def __init__(self, frames, batch, loss, dev):
        import torch
        from vi_slam_amd import capi
        self.torch, self.capi = torch, capi
        self.frames = [f[:batch] if f is not None else None for f in frames]
        self.B = batch
        self.pyr = [capi.alloc_pyramid(self.frames[0], LEVELS), capi.alloc_pyramid(self.frames[0], LEVELS)]
        self.fe = capi.FrontendStep(batch, H, W, NFEAT, LEVELS, WIN, quality=QUALITY, min_distance=MIN_DIST, device=dev)
        # first frame of every stream: pyramid + initial detection (untimed)
        self.cur = 0
        self.pyr[0][0] = self.frames[0]
        capi.build_pyramid(self.pyr[0])
        gf = capi.GoodFeatures(batch, H, W, NFEAT, device=dev)
        c, _, cnt = gf(self.frames[0], QUALITY, MIN_DIST)
        self.fe.pts.copy_(c)
        self.fe.n_pts.copy_(cnt)
        del gf
        nf = len(frames)
        self.order = list(range(1, nf)) + list(range(nf - 2, -1, -1))
        self.pos = 0
        self.keep_den = 1000
        self.keep_num = int(round((1.0 - min(max(loss, 0.0), 1.0)) * self.keep_den))

    def drop_tracks(self):
        """n <- floor(n * keep): the newest tracks of every stream are dropped.  One torch kernel: a look-up in the table of
        floor(n * keep) for n = 0 .. NFEAT into a spare count array, which then changes places with the step's."""
        torch = self.torch
        fe = self.fe
        if getattr(fe, "keep_table", None) is None:
            fe.keep_table = (torch.arange(NFEAT + 1, dtype=torch.int64, device=fe.n_pts.device) * self.keep_num // self.keep_den).to(torch.int32)
            fe.n_spare = torch.zeros_like(fe.n_pts)
        torch.index_select(fe.keep_table, 0, fe.n_pts, out=fe.n_spare)
        fe.n_pts, fe.n_spare = fe.n_spare, fe.n_pts

    def step(self):
        nxt = self.cur ^ 1
        self.pyr[nxt][0] = self.frames[self.order[self.pos % len(self.order)]]
        self.fe.step(self.pyr[self.cur], self.pyr[nxt])
        if self.keep_num < self.keep_den:
            self.drop_tracks()
        self.cur = nxt
        self.pos += 1

    def torch_kernels_ms(self, reps=50):
        """Average duration of the track-loss arithmetic of one step (not a library kernel, so not in the HIP-event table)."""
        torch = self.torch
        saved = self.fe.n_pts.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            self.drop_tracks()
        e1.record()
        torch.cuda.synchronize()
        self.fe.n_pts.copy_(saved)
        return e0.elapsed_time(e1) / reps


class UploadLeg(FrontendLeg):
    """The same step with every frame arriving from pinned host memory (the H2D-inclusive figure of configs[3]): three
    device frame buffers per stream, the upload of frame i + 1 on a copy stream beside the step on frame i (the step reads
    frame i - 1 as the previous image, so a buffer is free again one step later)."""

    def __init__(self, host_frames, batch, loss, dev):
        import torch
        from vi_slam_amd import capi
        # rows packed (pitch = width, 16-byte aligned at 1920): nothing but pixels crosses the link
        self.bufs = [torch.empty((batch, H, W), dtype=torch.uint8, device=dev) for _ in range(3)]
        self.pinned = [torch.from_numpy(np.ascontiguousarray(hf[:batch])).pin_memory() for hf in host_frames]
        self.bufs[0].copy_(self.pinned[0])
        super().__init__([self.bufs[0]] + [None] * (len(host_frames) - 1), batch, loss, dev)
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.up_done = [torch.cuda.Event() for _ in range(3)]
        self.step_done = torch.cuda.Event()
        self.step_done.record()
        self.upload(1)
        self.bytes_per_step = self.pinned[0].numel()

    def upload(self, i):
        """Frame i of the visiting order into buffer i % 3, once the step that last read that buffer is done."""
        torch = self.torch
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.step_done)
            self.bufs[i % 3].copy_(self.pinned[self.order[(i - 1) % len(self.order)]], non_blocking=True)
            self.up_done[i % 3].record(self.copy_stream)

    def step(self):
        torch = self.torch
        i = self.pos + 1
        torch.cuda.current_stream().wait_event(self.up_done[i % 3])
        self.upload(i + 1)  # waits for step i - 1 only: runs beside step i
        nxt = self.cur ^ 1
        self.pyr[nxt][0] = self.bufs[i % 3]
        self.fe.step(self.pyr[self.cur], self.pyr[nxt])
        if self.keep_num < self.keep_den:
            self.drop_tracks()
        self.step_done.record()
        self.cur = nxt
        self.pos += 1


class PipelinedLeg(FrontendLeg):
    """The same steps with the library's stage-by-stage entry points: pyramid and detector candidates of frame t + 1 on a side
    stream beside tracker and refill of frame t (capi.FrontendPipeline; three pyramids in rotation, two workspaces)."""

    def __init__(self, frames, batch, loss, dev):
        super().__init__(frames, batch, loss, dev)
        capi = self.capi
        pipe = capi.FrontendPipeline(batch, H, W, NFEAT, LEVELS, WIN, quality=QUALITY, min_distance=MIN_DIST, device=dev)
        pipe.pts.copy_(self.fe.pts)
        pipe.n_pts.copy_(self.fe.n_pts)
        self.fe = pipe
        self.pyr.append(capi.alloc_pyramid(self.frames[0], LEVELS))

    def step(self):
        nxt, coming = (self.cur + 1) % 3, (self.cur + 2) % 3
        self.pyr[nxt][0] = self.frames[self.order[self.pos % len(self.order)]]
        self.pyr[coming][0] = self.frames[self.order[(self.pos + 1) % len(self.order)]]
        self.fe.advance(self.pyr[self.cur], self.pyr[nxt], self.pyr[coming])
        if self.keep_num < self.keep_den:
            self.drop_tracks()
        self.cur = nxt
        self.pos += 1


def pipelined_steps(frames, batch, loss, dev, reps=40):
    """The leg's steps with the stages of consecutive frames side by side; after the same number of steps from the same start
    the points equal those of the one-call leg (checked on a fresh pair of legs, 12 steps)."""
    import torch
    try:
        a, b = FrontendLeg(frames, batch, loss, dev), PipelinedLeg(frames, batch, loss, dev)
        for _ in range(12):
            a.step()
            b.step()
        torch.cuda.synchronize()
        same = bool(torch.equal(a.fe.n_pts, b.fe.n_pts) and torch.equal(a.fe.pts, b.fe.pts))
        del a
        period = len(b.order)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps * period):
            b.step()
        torch.cuda.synchronize()
        per_step = (time.perf_counter() - t0) / (reps * period)
        return {"ms_per_step": per_step * 1e3, "frames_per_s_per_gpu": batch / per_step, "steps": reps * period,
                "same_points_as_the_one_call_step": same,
                "what": "vfe_dev_build_pyramid + vfe_dev_frontend_candidates of frame t + 1 on a side stream beside "
                        "vfe_dev_frontend_track + vfe_dev_frontend_refill of frame t (host wall clock, this rank); never the headline value"}
    except Exception as e:  # a measurement beside the contract: its failure must not take the line with it
        return {"error": f"{type(e).__name__}: {e}"}


def run_frontend_leg(rk, frames, batch, loss, steps, warmup, counters, leg, overlap):
    """-> (seconds of the timed region, kernel table, roofline, track statistics, timing note).  With the detector running
    beside the tracker (the library's default) per-kernel durations inside the timed region are those of kernels sharing the
    chip, so the kernel table and the roofline come from a second, shorter pass with everything on one stream."""
    from vi_slam_amd import capi
    L = FrontendLeg(frames, batch, loss, rk.dev)
    capi.set_overlap(overlap)
    # the headline region keeps its HIP events; the 64-stream region, whose kernel table comes from the pass below, does not
    elapsed, prof = timed(rk, L.step, steps, warmup, events=(leg == "main" or not overlap))
    k_elapsed, k_steps = elapsed, steps
    note = "kernel durations from the timed region itself (one stream)"
    if overlap:
        capi.set_overlap(False)
        k_steps = max(10, min(steps, 100))
        k_elapsed, prof = timed(rk, L.step, k_steps, 2)
        capi.set_overlap(True)
        note = (f"kernel durations from a separate pass of {k_steps} steps with every kernel on one stream "
                f"({k_elapsed / k_steps * 1e3:.3f} ms per step); in the timed region the eigenvalue pass runs beside pyramid + tracker")
    kernels = kernel_table(prof, k_elapsed, {k: v * batch for k, v in ALGO_BYTES.items()}, k_steps)
    extra_ms = L.torch_kernels_ms() if L.keep_num < L.keep_den else 0.0
    kernels["torch_track_loss"] = {"launches": k_steps, "avg_ms": extra_ms, "share": extra_ms * k_steps / (k_elapsed * 1e3),
                                   "algo_bytes_per_launch": 8 * batch, "algo_GBs": None,
                                   "what": "n_pts <- floor(n_pts * keep): one torch look-up kernel per step, timed apart"}
    tracked = float(L.fe.n_tracked.float().mean().item())
    points = float(L.fe.n_pts.float().mean().item())
    tracks = {"mean_tracked": tracked, "mean_points": points}
    if leg == "main" and steps >= 50:  # not in the few-step counter passes: their per-kernel averages are the step's launches
        tracks["klt_with_err_output"] = klt_full_call_ms(L)
    if steps >= 50:
        tracks["graph_replay"] = graph_replay(L, batch)
        if overlap:
            tracks["pipelined"] = pipelined_steps(frames, batch, loss, rk.dev)
    return elapsed, kernels, roofline_for(kernels, counters, leg, batch), tracks, note


def graph_replay(L, batch, reps=40):
    """The same steps replayed from a HIP graph: one period of the resident-frame cycle (six steps) captured once on a side
    stream -- the library's own fork to its helper stream and back is captured with it -- and replayed `reps` times.  What a
    host that submits whole periods gets; never the headline value (the timed region launches every kernel)."""
    import torch
    try:
        period = len(L.order)
        side = torch.cuda.Stream(device=L.fe.pts.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # the stream's first steps outside a capture: the helper stream is created here
            for _ in range(period):
                L.step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
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
        per_step = (time.perf_counter() - t0) / (reps * period)
        del g
        return {"ms_per_step": per_step * 1e3, "frames_per_s_per_gpu": batch / per_step, "steps": reps * period,
                "what": "six consecutive steps captured into one HIP graph and replayed (host wall clock, this rank)"}
    except Exception as e:  # a measurement beside the contract: its failure must not take the line with it
        return {"error": f"{type(e).__name__}: {e}"}


def klt_full_call_ms(L, reps=10):
    """The tracker as the whole cv::calcOpticalFlowPyrLK call, i.e. with the per-point residual (`err`) that the front-end
    step never reads and therefore skips: the leg's current pyramids and points, caller order, results discarded."""
    torch, capi = L.torch, L.capi
    prev, nxt = L.pyr[L.cur ^ 1], L.pyr[L.cur]
    pts, n = L.fe.pts, L.fe.n_pts
    out = [torch.zeros_like(pts), torch.zeros(pts.shape[:2], dtype=torch.uint8, device=pts.device),
           torch.zeros(pts.shape[:2], dtype=torch.float32, device=pts.device)]
    res = {}
    ws = torch.empty(capi.lib().vfe_klt_track_workspace(pts.shape[1], pts.shape[0]), dtype=torch.uint8, device=pts.device)
    calls = {"without_err_ms": lambda: capi.klt_track_no_err(prev, nxt, pts, n, out[0], out[1], win=WIN),
             "with_err_ms": lambda: capi.klt_track(prev, nxt, pts, n, out[0], out[1], out[2], win=WIN),
             "with_err_own_order_ms": lambda: capi.klt_track_ws(prev, nxt, pts, n, out[0], out[1], out[2], win=WIN, workspace=ws)}
    for name, call in calls.items():
        call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / reps
    res["what"] = ("vfe_dev_klt_track on the leg's pyramids and points in caller order, with and without the err output of "
                   "calcOpticalFlowPyrLK (the front-end step runs the form without), and vfe_dev_klt_track_ws, which orders "
                   "the points itself (order kernel included in the time)")
    return res


def orb_level_pixels():
    from ctypes import c_float, c_int
    from vi_slam_amd import capi
    wh = (c_int * (2 * ORB_LEVELS))()
    sc = (c_float * ORB_LEVELS)()
    assert capi.lib().vfe_orb_level_sizes(W, H, ORB_SCALE, ORB_LEVELS, wh, sc) == 0
    return [wh[2 * i] * wh[2 * i + 1] for i in range(ORB_LEVELS)]


def run_orb_match_leg(rk, frames, batch, steps, warmup, counters):
    """configs[2]: per step, 8-level ORB (2000 keypoints) on a new frame of each of `batch` streams, then kNN-2 + ratio test
    of its descriptors against those of the stream's previous frame."""
    import torch
    from vi_slam_amd import capi
    fr = [f[:batch] for f in frames]
    orb = [capi.Orb(batch, H, W, ORB_NFEAT, ORB_SCALE, ORB_LEVELS, ORB_FAST_THR, device=rk.dev) for _ in range(2)]
    state = {"cur": 0, "pos": 0, "out": None}
    nf = len(fr)
    order = list(range(1, nf)) + list(range(nf - 2, -1, -1))
    orb[0](fr[0])

    def step():
        nxt = state["cur"] ^ 1
        orb[nxt](fr[order[state["pos"] % len(order)]])
        state["out"] = capi.match_knn2_batch(orb[nxt].desc, orb[nxt].counts, orb[state["cur"]].desc, orb[state["cur"]].counts,
                                             KNN_RATIO)
        state["cur"] = nxt
        state["pos"] += 1

    elapsed, prof = timed(rk, step, steps, warmup)
    px = orb_level_pixels()
    nk = float(orb[state["cur"]].counts.float().mean().item())
    good = float(state["out"][2].float().sum(dim=1).mean().item())
    per_frame = {"k_orb_resize": px[0] + sum(px[1:]),           # read level 0, write levels 1..7
                 "k_fast": sum(px) + 12 * nk,                    # read every level once, write (x, y, score) per keypoint
                 "k_orb_select": 12 * nk + 32 * nk,              # FAST keypoints in, keypoint records out
                 "k_orb_desc": (32 + 43 * 48 + 32) * nk,         # keypoint records and 43 x 48 level windows in, descriptors out
                 "k_match": 32 * (2 * nk) + 16 * nk + nk}        # both descriptor sets in, top-2 lists + flags out
    kernels = kernel_table(prof, elapsed, {k: v * batch for k, v in per_frame.items()}, steps)
    total = rk.sum_over_ranks(batch * steps)
    orb_bytes = px[0] + sum(px[1:]) + (43 * 48 + 64) * nk        # image in, pyramid out, keypoint windows in, keypoints + descriptors out
    return {"workload": f"configs[2]: 8-level ORB ({ORB_NFEAT} keypoints, 1.2/8/31/20) on 1920x1080 frames + frame-to-frame "
                        f"Hamming kNN-2 with ratio {KNN_RATIO}, {batch} streams per GPU per step",
            "value": total / elapsed, "unit": "frames/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps,
            "streams_per_gpu": batch, "mean_keypoints": nk, "mean_ratio_matches": good,
            "algo_bytes_per_frame": orb_bytes + per_frame["k_match"],
            "frac_of_hbm_roofline": total / elapsed / rk.world * (orb_bytes + per_frame["k_match"]) / (HBM_PEAK_GBS * 1e9),
            "roofline": roofline_for(kernels, counters, "orb_match", batch), "kernels": kernels}


def run_reloc_leg(rk, steps, warmup, counters, force_collective=False):
    """configs[4]: 2000 queries against a 10 M-row map, rows [shard_range(10 M, rank, world)) on this rank; per-shard
    top-2 all-gathered and merged by (distance, global index).  Every query has a planted near-duplicate
    (<= 20 flipped bits) at a known global row: top-1 must be exactly that row at exactly that distance."""
    import torch
    from vi_slam_amd import capi, dist as vdist, synth
    from vi_slam_amd.planted import map_shard, verify_planted
    b, e = vdist.shard_range(MAP_ROWS, rk.rank, rk.world)
    # queries: ORB descriptors of the first frame of the base scene (the same on every rank), padded to 2000 if short
    img = capi.alloc_image(1, H, W, rk.dev)
    img.copy_(torch.from_numpy(synth.scene(synth.BASE_SEED, W, H, 0)[None]).to(rk.dev))
    orb = capi.Orb(1, H, W, ORB_NFEAT, ORB_SCALE, ORB_LEVELS, ORB_FAST_THR, device=rk.dev)
    orb(img)
    _, desc = orb.host_keypoints(0)
    q = np.concatenate([desc[:N_QUERIES], synth.descriptors(synth.MAP_SEED + 1, max(0, N_QUERIES - len(desc)))])[:N_QUERIES]
    q = np.ascontiguousarray(q)
    del orb, img
    shard = map_shard(q, b, e)
    qd, td = torch.from_numpy(q).to(rk.dev), torch.from_numpy(shard).to(rk.dev)
    del shard
    res = {}

    def step():
        res["r"] = vdist.sharded_match_knn2(qd, td, b, force_collective=force_collective)

    elapsed, prof = timed(rk, step, steps, warmup)
    idx, dist_ = res["r"]
    ok, exact_top1 = verify_planted(q, idx.cpu().numpy(), dist_.cpu().numpy())
    if not ok:
        raise SystemExit("bench.py: relocalisation leg returned wrong neighbours for the planted duplicates")
    rows = e - b
    per_step = {"k_match": 32 * (N_QUERIES + rows) + 16 * N_QUERIES}
    kernels = kernel_table(prof, elapsed, per_step, steps)
    pairs = N_QUERIES * MAP_ROWS * steps
    roof = roofline_for(kernels, counters, "reloc", rows, {"k_match": MATCH_FLOP_PER_PAIR * N_QUERIES * rows}, steps)
    out = {"workload": f"configs[4]: {N_QUERIES} query descriptors against a {MAP_ROWS}-row 256-bit map, rows sharded over "
                       f"{rk.world} GPU(s), per-shard top-2 all-gathered (32 000 B per rank) and merged",
           "value": N_QUERIES * steps / elapsed, "unit": "queries/s", "ms_per_query_set": elapsed / steps * 1e3, "steps": steps,
           "pairs_per_s": pairs / elapsed, "rows_per_gpu": rows, "planted_neighbours_verified": ok, "queries_whose_top1_is_their_own_planted_row": exact_top1,
           "collective": ("all_gather (RCCL)" if rk.world > 1 and not rk.gloo else "all_gather (gloo rehearsal)" if rk.world > 1
                          else "all_gather (RCCL, one-rank group, forced)" if force_collective else "none (one shard)"),
           "roofline": roof, "kernels": kernels}
    if rk.world == 1:
        # the same query set through the C++ host (vi_slam::ShardedMatcher: scan, RCCL all-gathers issued from C++ over a
        # one-rank communicator, merge; queries in and results out over the link, one synchronisation per call)
        try:
            from vi_slam_amd import engine as eng
            sm = eng.ShardedMatcher((rk.dev.index or 0,))
            sm.set_map(td.cpu().numpy())
            ms = []
            for i in range(max(5, steps // 3)):
                ci, cd, _ = sm.match(q, KNN_RATIO)
                ms.append(sm.last_ms)
            ok_cpp, _ = verify_planted(q, ci, cd)
            out["cpp_sharded_matcher"] = {"ms_per_query_set": float(np.median(ms[1:])), "collective": sm.collective,
                                          "planted_neighbours_verified": bool(ok_cpp),
                                          "includes": "query upload, scan, all-gather, merge, ratio test, result download"}
            sm.close()
        except (RuntimeError, OSError) as e:  # RCCL not loadable from C++ on this box: report, do not fail the leg
            out["cpp_sharded_matcher"] = {"error": str(e)}
    if "k_match" in kernels:
        # the search kernel alone (sample + main launch of a step), without all-gather and merges
        km = kernels["k_match"]
        out["k_match_pairs_per_s"] = N_QUERIES * rows / (km["avg_ms"] * 1e-3 * km["launches"] / steps)
    return out


def run_cpp_batch_host_leg(host, nb, device, steps, warmup):
    """configs[3]'s per-GPU share through the C++ host (vi_slam::BatchFrontEnd: one host thread per GPU, pinned staging,
    upload of set i + 1 beside the step on set i, results collected one submission late): (i) frames written in place into
    the pinned staging memory, as a decoder would deliver them, (ii) frames in pageable caller memory, repacked by the
    host's copy threads.  Every frame crosses the link; compare with streams64.h2d_inclusive (the Python loop)."""
    from vi_slam_amd import engine as eng
    out = {}
    nf = len(host)
    order = list(range(1, nf)) + list(range(nf - 2, -1, -1))
    threads = max(1, min(8, host_cores() - 2))
    for name in ("staged", "caller_memory"):
        fe = eng.BatchFrontEnd(W, H, nb, max_features=NFEAT, levels=LEVELS, win=WIN, quality=QUALITY, min_distance=MIN_DIST,
                               devices=(device,), copy_threads=threads)
        views = [[host[t][b] for b in range(nb)] for t in range(nf)]

        def put(i):
            fr = views[order[i % len(order)]] if i else views[0]
            if name == "staged":
                for b in range(nb):  # the memcpy a producer would not need: it would write here directly
                    fe.staging(b)[:, :W] = fr[b]
                return fe.submit_staged(i)
            return fe.submit(fr, i)

        # at least three sets before the clock starts: from then on every staging slot holds a frame, and the staged path
        # re-submits what its slot holds (frames 1, 2, 3, 1, 2, 3 ... of each stream) without touching host memory -- the
        # shape of a producer that decodes straight into the pinned slot
        warm = max(warmup, 3)
        prev = put(0)
        for i in range(1, warm + 1):
            t = put(i)
            fe.wait(prev, points=False)
            prev = t
        t0 = time.perf_counter()
        for i in range(warm + 1, warm + 1 + steps):
            t = fe.submit_staged(i) if name == "staged" else put(i)
            tot, trk, _ = fe.wait(prev, points=False)
            prev = t
        tot, trk, _ = fe.wait(prev, points=False)
        el = time.perf_counter() - t0
        fe.close()
        out[name] = {"value": nb * steps / el, "unit": "frames/s", "steps": steps, "ms_per_step": el / steps * 1e3,
                     "streams": nb, "copy_threads": threads, "mean_points": float(np.mean(tot)), "mean_tracked": float(np.mean(trk)),
                     "upload_GBs": nb * W * H * steps / el / 1e9}
    out["what"] = ("vi_slam::BatchFrontEnd (C++ host, libvi_slam_amd.so) driven through its C view: per step one set of "
                   f"{nb} 1920x1080 frames uploaded from host memory and stepped; no track-loss model (the scene loses few tracks)")
    return out


def run_single_stream_leg(frames_host):
    """configs[0] / [1] through the host-pointer API the adapters call (vfe_process_frame: upload + pyramid + KLT + refill,
    one synchronisation per call), one live stream on this rank's GPU."""
    from vi_slam_amd import capi, synth
    out = {}
    for name, (w, h, n, d, lv, nfr) in {"vga_150": (752, 480, 150, 30.0, 3, 200), "1080p_1000": (W, H, NFEAT, MIN_DIST, LEVELS, 120)}.items():
        seq = [synth.scene(synth.BASE_SEED, w, h, t) for t in range(8)] if (w, h) != (W, H) else frames_host
        ctx = capi.Context(w, h, max_features=n, pyramid_levels=lv, win_size=WIN, orb_levels=0)
        order = list(range(len(seq))) + list(range(len(seq) - 2, 0, -1))
        lat = []
        for i in range(nfr):
            f = seq[order[i % len(order)]]
            t0 = time.perf_counter()
            ctx.process_frame(f, 10**9 + i * 33_333_333, n, QUALITY, d)
            lat.append((time.perf_counter() - t0) * 1e3)
        ctx.close()
        lat = np.array(lat[10:])
        out[name] = {"frame": f"{w}x{h}", "features": n, "levels": lv, "frames": len(lat), "median_ms": float(np.median(lat)),
                     "p95_ms": float(np.percentile(lat, 95)), "frames_per_s": float(1e3 / np.median(lat)),
                     "includes": "host-to-device upload of the frame, one synchronisation per call"}
    return out


# ------------------------------------------------------------------------------------------------ main

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500, help="timed steps of the headline leg (default: about one second)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="independent streams per GPU (headline leg)")
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic sequences tiled over the batch")
    ap.add_argument("--frames", type=int, default=4, help="resident frames per stream (visited ping-pong)")
    ap.add_argument("--loss", type=float, default=0.05,
                    help="fraction of each stream's tracks dropped after every step (models feature loss, so that "
                         "every frame has corners to refill; the synthetic scene alone loses almost none)")
    ap.add_argument("--overlap", choices=["on", "off"], default="on",
                    help="on (the library's default): the detector's eigenvalue pass runs beside pyramid + tracker on a helper "
                         "stream, and per-kernel durations come from a separate one-stream pass; off: everything on one stream")
    ap.add_argument("--legs", default="main,streams64,orb_match,reloc,single_stream,cpp_host",
                    help="comma-separated legs to run (main always runs)")
    ap.add_argument("--orb-batch", type=int, default=64, help="streams per GPU in the orb_match leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--force-collective", action="store_true",
                    help="with --gpus 1: initialise a one-rank RCCL group and run the reloc leg's all-gather + merge through it, "
                         "so that the exchange executes on a one-GPU machine (same results, the collective's cost included)")
    ap.add_argument("--rehearse-one-device", action="store_true",
                    help="multi-rank rehearsal on a box with one GPU: every rank uses cuda:0 and the collectives go over gloo "
                         "(RCCL refuses two ranks on one device); not a measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus) if not args.rehearse_one_device else launch_rehearsal(args.gpus))
    overlap = args.overlap == "on"
    legs = set(args.legs.split(","))
    # stdout carries the one JSON line and nothing else: libraries that print to descriptor 1 (RCCL writes a version banner
    # there when a group is initialised) are sent to stderr, the line goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from vi_slam_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the front-end has no CPU fallback")
    if args.rehearse_one_device:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} has no GPU (local rank {local_rank}, {torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.rehearse_one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    elif args.force_collective:  # a one-rank RCCL group, set up before any other GPU work of this process
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    capi.lib()
    if os.environ.get("VFE_BENCH_WATCHDOG"):  # stack traces of every thread if a leg stalls (debugging aid)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["VFE_BENCH_WATCHDOG"]), exit=True)
    rk = Ranks(world, rank, dev, args.rehearse_one_device)
    counters = load_counters()

    B = args.batch
    n_distinct = min(args.distinct, B)
    log(f"rendering {n_distinct} x {args.frames} synthetic 1080p frames")
    host = make_frames(rank, B, n_distinct, args.frames)
    log("headline leg")
    frames = []
    for t in range(args.frames):
        d = capi.alloc_image(B, H, W, dev)
        d.copy_(torch.from_numpy(host[t]).to(dev))
        frames.append(d)

    # ---- headline leg: configs[1] shape, `B` streams per GPU
    elapsed, kernels, roof, tracks, knote = run_frontend_leg(rk, frames, B, args.loss, args.steps, args.warmup, counters, "main", overlap)
    value = B * world * args.steps / elapsed
    out = {
        "metric": "front-end frames/s at 1920x1080, 1000 feats; % of HBM roofline",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8/int32/f32", "data": "synthetic",
        "config": {"workload": "the per-frame work of configs[1] (1920x1080, 1000 Shi-Tomasi corners + 4-level KLT, win 21, + "
                               f"refill) batched as configs[3] batches it, over {B} independent streams per GPU per step "
                               f"({args.loss:.0%} of the tracks dropped per step); configs[3]'s own 64 streams: see streams64, "
                               "one live stream (configs[1] as stated): see single_stream",
                   "streams_per_gpu": B, "distinct_sequences": n_distinct, "track_loss_per_step": args.loss, "overlap": args.overlap,
                   "frame_bytes": W * H, "algo_bytes_per_frame": BYTES_FRAME,
                   "parallelism": f"streams sharded over {world} GPU(s), no collective"},
        "frontend_frac_of_hbm_roofline": value / world * BYTES_FRAME / (HBM_PEAK_GBS * 1e9),
        "roofline": roof, "kernels": kernels, "kernel_timing": knote, "tracks": tracks,
    }

    # ---- configs[3]: 64 streams in total, split over the ranks (strong scaling)
    if "streams64" in legs:
        from vi_slam_amd import dist as vdist
        b0, b1 = vdist.shard_range(64, rank, world)
        nb = b1 - b0
        if nb > 0 and nb <= B:
            log(f"streams64 leg ({nb} streams on this GPU)")
            steps64 = args.steps
            el, ker, rf, trk, kn = run_frontend_leg(rk, frames, nb, args.loss, steps64, args.warmup, counters, "streams64", overlap)
            out["streams64"] = {"workload": f"configs[3]: 64 independent 1080p streams (detect + track) over {world} GPU(s), "
                                            f"{nb} per GPU", "value": 64 * steps64 / el, "unit": "frames/s", "scaling": "strong",
                                "streams_per_gpu": nb, "steps": steps64, "ms_per_step": el / steps64 * 1e3,
                                "frac_of_hbm_roofline": 64 * steps64 / el / world * BYTES_FRAME / (HBM_PEAK_GBS * 1e9),
                                "roofline": rf, "kernels": ker, "kernel_timing": kn, "tracks": trk}
            log("streams64 leg, frames uploaded from pinned host memory every step")
            U = UploadLeg(host, nb, args.loss, dev)
            capi.set_overlap(overlap)
            steps_up = max(20, args.steps // 4)
            el_up, _ = timed(rk, U.step, steps_up, max(3, args.warmup // 2), events=False)
            out["streams64"]["h2d_inclusive"] = {
                "value": 64 * steps_up / el_up, "unit": "frames/s", "steps": steps_up, "ms_per_step": el_up / steps_up * 1e3,
                "upload_bytes_per_step_per_gpu": U.bytes_per_step,
                "upload_GBs_per_gpu": U.bytes_per_step * steps_up / el_up / 1e9,
                "what": "the same step with every frame copied from pinned host memory first (packed rows, one "
                        "contiguous copy per batch on a copy stream, three frame buffers per stream so that the upload of the next "
                        "frame runs beside the step); never the headline value"}
            del U
    if "orb_match" in legs:
        ob = min(args.orb_batch, B)
        log("orb_match leg")
        out["orb_match"] = run_orb_match_leg(rk, frames, ob, max(20, args.steps // 5), max(2, args.warmup // 2), counters)
    frames_host_1080 = [host[t][0] for t in range(args.frames)]
    del frames
    torch.cuda.empty_cache()
    if "reloc" in legs:
        log("reloc leg")
        out["reloc"] = run_reloc_leg(rk, max(10, args.steps // 10), 3, counters, args.force_collective)
    rk.barrier()
    if rank == 0:
        log("single_stream / cpu_baseline legs")
        if "single_stream" in legs:
            out["single_stream"] = run_single_stream_leg(frames_host_1080)
        if "cpp_host" in legs and world == 1:
            log("cpp_host leg (vi_slam::BatchFrontEnd)")
            out["cpp_batch_host"] = run_cpp_batch_host_leg(host, min(64, B), local_rank, max(20, args.steps // 10), 3)
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(host, args.loss, n_distinct)
            out["speedup_vs_cpu_all_cores"] = value / out["cpu_baseline"]["value"]
            out["speedup_vs_cpu_1core"] = value / out["cpu_baseline"]["one_core"]["value"]
            out["speedup_vs_plain_port_all_cores"] = value / out["cpu_baseline"]["port"]["value"]
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    rk.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


def launch_rehearsal(n):
    """--rehearse-one-device with --gpus N: N ranks on cuda:0 over gloo (no device-count check)."""
    env = dict(os.environ)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


if __name__ == "__main__":
    main()
