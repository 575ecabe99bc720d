#!/usr/bin/env python3
"""Times vfe_dev_match_knn2 alone on random descriptors: 2000 queries against N rows (default 10 M).
Usage: python tools/match_only.py [rows] (VFE_LIB_PATH selects a variant build)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vi_slam_amd import capi

NT = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
q = torch.randint(0, 256, (2000, 32), dtype=torch.uint8, device=dev, generator=g)
t = torch.randint(0, 256, (NT, 32), dtype=torch.uint8, device=dev, generator=g)
for _ in range(2):
    capi.match_knn2(q, t)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
reps = 10
e[0].record()
for _ in range(reps):
    capi.match_knn2(q, t)
e[1].record()
torch.cuda.synchronize()
ms = e[0].elapsed_time(e[1]) / reps
flop = 2000.0 * NT * 512
print(f"2000 x {NT}: {ms:.3f} ms per search, {flop / ms / 1e12:.2f} PFLOP/s = {flop / ms / 1e12 / 10.066 * 100:.1f} % of the fp4 peak, lib {capi.LIB_PATH}")
