#!/usr/bin/env python3
"""Times vfe_dev_build_pyramid alone on resident 1080p frames: per-launch HIP-event durations of the three levels.
Usage: python tools/pyr_only.py [batch] (VFE_LIB_PATH selects a variant build)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vi_slam_amd import capi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
img = capi.alloc_image(B, 1080, 1920, dev)
img.random_(0, 256)
lv = capi.alloc_pyramid(img, 4)
for _ in range(3):
    capi.build_pyramid(lv)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
reps = 50
e[0].record()
for _ in range(reps):
    capi.build_pyramid(lv)
e[1].record()
torch.cuda.synchronize()
ms = e[0].elapsed_time(e[1]) / reps
algo = B * (2073600 + 518400 + 129600 + 32400)
print(f"batch {B}: pyramid {ms*1e3:.1f} us per call, {algo/ms/1e9:.2f} TB/s algorithmic ({algo/1e6:.0f} MB), lib {capi.LIB_PATH}")
