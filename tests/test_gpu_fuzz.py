"""GPU parity on seeded random shapes and parameters: every case is checked bit for bit against the CPU oracle through
the C ABI.  The cases are drawn from a fixed seed, so a failure names a reproducible (stage, case) pair."""
import numpy as np
import pytest

from oracle import oracle as ora
from vi_slam_amd import capi, synth

pytestmark = pytest.mark.gpu

import os
# VFE_FUZZ_SCALE=10 runs ten times as many cases (a sweep to run by hand; the default keeps the suite quick)
SCALE = int(os.environ.get("VFE_FUZZ_SCALE", "1"))


def _dev(imgs, cuda):
    import torch
    t = capi.alloc_image(len(imgs), imgs[0].shape[0], imgs[0].shape[1], cuda)
    t.copy_(torch.from_numpy(np.stack(imgs)).to(cuda))
    return t


def _images(rng, w, h, k):
    out = []
    for i in range(k):
        kind = rng.integers(0, 4)
        seed = int(rng.integers(1, 1 << 30))
        if kind == 0:
            out.append(synth.noise(seed, w, h))
        elif kind == 1:
            out.append(synth.scene(seed, w, h, int(rng.integers(0, 5))))
        elif kind == 2:
            out.append((synth.scene(seed, w, h, 0) // 8 * 8).astype(np.uint8))  # plateaus: ties everywhere
        else:
            g = np.add.outer(np.arange(h) * int(rng.integers(1, 4)), np.arange(w) * int(rng.integers(1, 4)))
            out.append((g % 256).astype(np.uint8))                              # ramps with wrap-around edges
    return out


@pytest.mark.parametrize("case", range(12 * SCALE))
def test_fuzz_pyramid_fast_gftt(cuda, case):
    rng = np.random.default_rng(1000 + case)
    w, h = int(rng.integers(16, 520)), int(rng.integers(16, 300))
    imgs = _images(rng, w, h, 3)
    t = _dev(imgs, cuda)
    # pyrDown
    down = capi.pyr_down(t).cpu().numpy()
    for i, im in enumerate(imgs):
        assert np.array_equal(down[i], ora.pyr_down(im)), f"pyr_down {w}x{h} frame {i}"
    # FAST-9
    thr = int(rng.integers(1, 80))
    kps, cnt = capi.fast9(t, thr)
    kps, cnt = kps.cpu().numpy(), cnt.cpu().numpy()
    for i, im in enumerate(imgs):
        ref = ora.fast9(im, thr)
        assert cnt[i] == len(ref) and np.array_equal(kps[i, :cnt[i]], ref), f"fast9 {w}x{h} thr {thr} frame {i}"
    # Shi-Tomasi
    n = int(rng.integers(1, 600)); q = float(rng.choice([0.001, 0.01, 0.05, 0.3])); d = float(rng.choice([0, 1, 2.5, 7, 15, 40]))
    gf = capi.GoodFeatures(len(imgs), h, w, n, device=cuda)
    c, s, k = gf(t, q, d)
    c, s, k = c.cpu().numpy(), s.cpu().numpy(), k.cpu().numpy()
    for i, im in enumerate(imgs):
        rxy, rsc = ora.good_features(im, n, q, d)
        assert k[i] == len(rxy), f"gftt {w}x{h} n {n} q {q} d {d} frame {i}: {k[i]} vs {len(rxy)}"
        assert np.array_equal(c[i, :k[i]], rxy) and np.array_equal(s[i, :k[i]].view(np.uint32), rsc.view(np.uint32))


@pytest.mark.parametrize("case", range(8 * SCALE))
def test_fuzz_klt(cuda, case):
    import torch
    rng = np.random.default_rng(2000 + case)
    w, h = int(rng.integers(64, 500)), int(rng.integers(64, 300))
    levels = int(rng.integers(1, 5))
    levels = min(levels, capi.lk_max_level(w, h, 21, levels - 1) + 1)  # as cv::buildOpticalFlowPyramid limits them
    seed = int(rng.integers(1, 1 << 30))
    a, b = synth.scene(seed, w, h, 0), synth.scene(seed, w, h, int(rng.integers(1, 3)))
    npts = int(rng.integers(1, 200))
    pts = np.column_stack([rng.uniform(-3, w + 3, npts), rng.uniform(-3, h + 3, npts)]).astype(np.float32)
    pa = capi.build_pyramid(capi.alloc_pyramid(_dev([a], cuda), levels))
    pb = capi.build_pyramid(capi.alloc_pyramid(_dev([b], cuda), levels))
    nxt, st, err = capi.klt_track(pa, pb, torch.from_numpy(pts[None]).to(cuda))
    rn, rs, re = ora.klt_track(a, b, pts, 21, levels - 1)
    st = st[0].cpu().numpy().astype(bool)
    assert np.array_equal(st, rs.astype(bool)), f"klt {w}x{h} L{levels}"
    assert np.array_equal(nxt[0].cpu().numpy()[st].view(np.uint32), rn[st].view(np.uint32))
    assert np.array_equal(err[0].cpu().numpy()[st].view(np.uint32), re[st].view(np.uint32))


@pytest.mark.parametrize("case", range(6 * SCALE))
def test_fuzz_orb(cuda, case):
    rng = np.random.default_rng(3000 + case)
    w, h = int(rng.integers(80, 500)), int(rng.integers(80, 320))
    nfeat = int(rng.integers(20, 900))
    nlevels = int(rng.integers(1, 9))
    sf = float(rng.choice([1.2, 1.5, 2.0]))
    thr = int(rng.choice([7, 20, 35]))
    imgs = _images(rng, w, h, 2)
    orb = capi.Orb(len(imgs), h, w, nfeat, sf, nlevels, thr, device=cuda)
    orb(_dev(imgs, cuda))
    for i, im in enumerate(imgs):
        rk, rd = ora.orb_extract(im, nfeat, sf, nlevels, thr)
        gk, gd = orb.host_keypoints(i)
        tag = f"orb {w}x{h} n {nfeat} L{nlevels} sf {sf} thr {thr} frame {i}"
        assert len(gk) == len(rk), f"{tag}: {len(gk)} vs {len(rk)}"
        for f in ("octave", "lx", "ly"):
            assert np.array_equal(gk[f], rk[f]), f"{tag}: {f}"
        for f in ("x", "y", "size", "response", "angle"):
            assert np.array_equal(gk[f].view(np.uint32), rk[f].view(np.uint32)), f"{tag}: {f}"
        assert np.array_equal(gd, rd), f"{tag}: descriptors"
