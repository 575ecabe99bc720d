#!/usr/bin/env python3
"""Times vfe_dev_klt_track alone on resident 1080p frame pairs (smooth random texture, second frame shifted by a
sub-pixel offset), 1000 points per frame, for several batch sizes, with the one-wave-per-point kernel off and on
(VFE_KLT_WIDE_MAX is read at every call).  Usage: python tools/klt_only.py [points_per_frame]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from vi_slam_amd import capi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(7)


def texture(b):
    t = torch.rand((b, 1, 1080 // 8 + 2, 1920 // 8 + 2), generator=g)
    t = F.interpolate(t, scale_factor=8, mode="bicubic", align_corners=False)
    return (t.clamp(0, 1) * 255).to(torch.uint8)


for B in (1, 2, 4, 8, 16, 32, 64):
    big = texture(B)
    a = capi.alloc_image(B, 1080, 1920, dev)
    c = capi.alloc_image(B, 1080, 1920, dev)
    a.copy_(big[:, 0, 4:1084, 4:1924].to(dev))
    c.copy_(big[:, 0, 6:1086, 7:1927].to(dev))
    la, lc = capi.alloc_pyramid(a, 4), capi.alloc_pyramid(c, 4)
    capi.build_pyramid(la)
    capi.build_pyramid(lc)
    pts = torch.rand((B, N, 2), generator=g) * torch.tensor([1900.0, 1060.0]) + 10.0
    pts = pts.to(dev)
    n = torch.full((B,), N, dtype=torch.int32, device=dev)
    out = torch.zeros_like(pts)
    st = torch.zeros((B, N), dtype=torch.uint8, device=dev)
    er = torch.zeros((B, N), dtype=torch.float32, device=dev)
    res = {}
    for wide in (0, 1 << 30):
        os.environ["VFE_KLT_WIDE_MAX"] = str(wide)
        for _ in range(3):
            capi.klt_track(la, lc, pts, n, out, st, er)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        reps = 30
        e[0].record()
        for _ in range(reps):
            capi.klt_track(la, lc, pts, n, out, st, er)
        e[1].record()
        torch.cuda.synchronize()
        res[wide] = (e[0].elapsed_time(e[1]) / reps * 1e3, out.clone(), st.clone(), er.clone())
    same = all(torch.equal(x, y) for x, y in zip(res[0][1:], res[1 << 30][1:]))
    print(f"batch {B:3d} x {N}: four per wave {res[0][0]:7.1f} us, one per wave {res[1 << 30][0]:7.1f} us, "
          f"tracked {int(st.sum())}/{B * N}, identical {same}", flush=True)
