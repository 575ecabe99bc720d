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
def test_fuzz_klt(cuda, case, klt_kernel):
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


def _oracle_step(prev, cur, pts, n, q, d, levels, w, h, f_thr=0.0):
    nxt, st, _ = ora.klt_track(prev, cur, pts, 21, levels - 1)
    r = np.rint(nxt)
    keep = st.astype(bool) & (r[:, 0] >= 1) & (r[:, 0] < w - 1) & (r[:, 1] >= 1) & (r[:, 1] < h - 1)
    if f_thr > 0:
        _, mask, _ = ora.fundamental_ransac(pts, nxt, keep.astype(np.uint8), f_thr, 0.99, 1000, seed=0x5EED)
        keep = mask.astype(bool)
    tracked = nxt[keep]
    if n - len(tracked) > 0:
        new, _ = ora.good_features(cur, n - len(tracked), q, d, seeds=tracked)
        return (np.concatenate([tracked, new]) if len(new) else tracked), len(tracked)
    return tracked, len(tracked)


@pytest.mark.parametrize("case", range(6 * SCALE))
def test_fuzz_frontend_steps(cuda, case):
    """Two steady-state steps (pyramid, tracker, compaction, refill against the tracked points) on random shapes."""
    import torch
    rng = np.random.default_rng(4000 + case)
    w, h = int(rng.integers(96, 520)), int(rng.integers(96, 320))
    n = int(rng.integers(5, 400)); d = float(rng.choice([3, 5, 8, 12, 20, 33])); q = float(rng.choice([0.005, 0.01, 0.05]))
    levels = capi.lk_max_level(w, h, 21, int(rng.integers(0, 4))) + 1
    B = int(rng.integers(1, 4))
    seeds = [int(rng.integers(1, 1 << 30)) for _ in range(B)]
    seq = [[synth.scene(s, w, h, t) for t in range(3)] for s in seeds]
    fe = capi.FrontendStep(B, h, w, n, levels, 21, quality=q, min_distance=d, device=cuda)
    f_thr = float(rng.choice([0.0, 0.0, 0.03, 0.2, 1.0]))   # epipolar filter between tracking and refill
    fe.set_f_rejection(f_thr)
    frames = [_dev([seq[b][t] for b in range(B)], cuda) for t in range(3)]
    pyr = [capi.alloc_pyramid(frames[0], levels), capi.alloc_pyramid(frames[0], levels)]
    capi.build_pyramid(pyr[0])
    ref = []
    for b in range(B):
        k0 = int(rng.integers(0, n + 1))   # start with fewer points than the capacity, down to none
        xy, _ = ora.good_features(seq[b][0], max(k0, 1), q, d)
        xy = xy[:k0]
        ref.append(xy)
        fe.pts[b, :len(xy)] = torch.from_numpy(xy).to(cuda)
        fe.n_pts[b] = len(xy)
    cur = 0
    for t in range(1, 3):
        nxt = cur ^ 1
        pyr[nxt][0] = frames[t]
        pts, n_pts, n_tr = fe.step(pyr[cur], pyr[nxt])
        cur = nxt
        pts, n_pts, n_tr = pts.cpu().numpy(), n_pts.cpu().numpy(), n_tr.cpu().numpy()
        for b in range(B):
            ref[b], ntr = _oracle_step(seq[b][t - 1], seq[b][t], ref[b], n, q, d, levels, w, h, f_thr)
            tag = f"{w}x{h} n {n} d {d} q {q} L{levels} F {f_thr} t={t} b={b}"
            assert n_tr[b] == ntr and n_pts[b] == len(ref[b]), f"{tag}: {n_tr[b]}/{n_pts[b]} vs {ntr}/{len(ref[b])}"
            assert np.array_equal(pts[b, :n_pts[b]].view(np.uint32), ref[b].view(np.uint32)), tag


@pytest.mark.parametrize("case", range(6 * SCALE))
def test_fuzz_match(cuda, case):
    import torch
    rng = np.random.default_rng(5000 + case)
    nq, nt = int(rng.integers(1, 700)), int(rng.integers(0, 9000))
    q = synth.descriptors(int(rng.integers(1, 1 << 20)), nq)
    t = synth.descriptors(int(rng.integers(1, 1 << 20)), nt) if nt else np.zeros((0, 32), np.uint8)
    for i in range(min(nq, nt, 40)):     # near-duplicates and exact ties
        t[int(rng.integers(0, nt))] = synth.flip_bits(100 + i, q[i], int(rng.integers(0, 30)))
    if nt > 10:
        t[int(rng.integers(0, nt))] = t[int(rng.integers(0, nt))]
    base = int(rng.integers(0, 1 << 20))
    dq = torch.from_numpy(q).to(cuda)
    dt = torch.from_numpy(t).to(cuda) if nt else torch.zeros((0, 32), dtype=torch.uint8, device=cuda)
    idx, dist = capi.match_knn2(dq, dt, base_index=base)
    ridx, rdist = ora.match_knn2(q, t, base)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(dist.cpu().numpy(), rdist), f"nq {nq} nt {nt}"
    ratio = float(rng.choice([0.6, 0.7, 0.8]))
    assert np.array_equal(capi.ratio_test(idx, dist, ratio).cpu().numpy(), ora.ratio_test(ridx, rdist, ratio))


@pytest.mark.parametrize("case", range(6 * SCALE))
def test_fuzz_preprocess_and_refine(cuda, case):
    """equalizeHist, CLAHE and cornerSubPix on random shapes, grids and windows."""
    import torch
    rng = np.random.default_rng(6000 + case)
    w, h = int(rng.integers(40, 520)), int(rng.integers(40, 300))
    imgs = _images(rng, w, h, 2)
    t = _dev(imgs, cuda)
    eq = capi.equalize_hist(t).cpu().numpy()
    tx, ty = int(rng.integers(1, 9)), int(rng.integers(1, 9))
    clip = float(rng.choice([0.0, 1.5, 4.0, 10.0, 40.0]))
    cl = capi.clahe(t, clip, tx, ty).cpu().numpy()
    for i, im in enumerate(imgs):
        assert np.array_equal(eq[i], ora.equalize_hist(im)), f"equalize {w}x{h} frame {i}"
        assert np.array_equal(cl[i], ora.clahe(im, clip, tx, ty)), f"clahe {w}x{h} clip {clip} grid {tx}x{ty} frame {i}"
    win = int(rng.integers(1, 11)); zz = int(rng.choice([-1, -1, 0, 1])); it = int(rng.integers(1, 40)); eps = float(rng.choice([0.0, 0.001, 0.03]))
    npts = int(rng.integers(1, 150))
    pts = np.stack([np.column_stack([rng.uniform(0, w - 1, npts), rng.uniform(0, h - 1, npts)]) for _ in imgs]).astype(np.float32)
    d_pts = torch.from_numpy(pts).to(cuda)
    capi.corner_subpix(t, d_pts, None, win, zz, it, eps)
    got = d_pts.cpu().numpy()
    for i, im in enumerate(imgs):
        ref = ora.corner_subpix(im, pts[i], win, zz, it, eps)
        assert np.array_equal(got[i].view(np.uint32), ref.view(np.uint32)), f"subpix {w}x{h} win {win} zz {zz} it {it} eps {eps} frame {i}"


@pytest.mark.parametrize("case", range(6 * SCALE))
def test_fuzz_fast_grid(cuda, case):
    """Grid-bucketed FAST (OpenVINS-style) with random grids, budgets, distances and tracked points."""
    import torch
    rng = np.random.default_rng(7000 + case)
    gx, gy = int(rng.integers(1, 9)), int(rng.integers(1, 7))
    w, h = int(rng.integers(max(64, 12 * gx), 520)), int(rng.integers(max(48, 12 * gy), 320))
    thr = int(rng.choice([7, 15, 20, 40])); n = int(rng.integers(1, 400)); d = int(rng.choice([1, 3, 5, 10, 20]))
    imgs = _images(rng, w, h, 2)
    t = _dev(imgs, cuda)
    nseed = int(rng.integers(0, 60))
    seeds0 = np.column_stack([rng.uniform(0, w - 1, nseed), rng.uniform(0, h - 1, nseed)]).astype(np.float32)
    sd = torch.zeros((2, 64, 2), dtype=torch.float32, device=cuda)
    sd[0, :nseed] = torch.from_numpy(seeds0).to(cuda)
    ns = torch.tensor([nseed, 0], dtype=torch.int32, device=cuda)
    pts, sc, cnt = capi.fast_grid(t, thr, gx, gy, n, d, sd, ns)
    pts, sc, cnt = pts.cpu().numpy(), sc.cpu().numpy(), cnt.cpu().numpy()
    for i, im in enumerate(imgs):
        rxy, rsc = ora.fast_grid(im, thr, gx, gy, n, d, seeds0 if i == 0 else None)
        tag = f"fast_grid {w}x{h} grid {gx}x{gy} thr {thr} n {n} d {d} seeds {nseed} frame {i}"
        assert cnt[i] == len(rxy), f"{tag}: {cnt[i]} vs {len(rxy)}"
        assert np.array_equal(pts[i, :cnt[i]], rxy) and np.array_equal(sc[i, :cnt[i]], rsc), tag


@pytest.mark.parametrize("case", range(4 * SCALE))
def test_fuzz_fundamental_ransac(cuda, case):
    import torch
    from two_view import two_view
    rng = np.random.default_rng(8000 + case)
    B = int(rng.integers(1, 5))
    cap = int(rng.integers(8, 700))
    p1 = np.zeros((B, cap, 2), np.float32); p2 = np.zeros((B, cap, 2), np.float32)
    st = np.zeros((B, cap), np.uint8); n = np.zeros(B, np.int32)
    for b in range(B):
        x1, x2, _, _ = two_view(int(rng.integers(1, cap + 1)), float(rng.choice([0, 0.1, 0.3, 0.6])), float(rng.choice([0, 0.2, 0.5])),
                                int(rng.integers(1, 1 << 20)))
        n[b] = len(x1); p1[b, :n[b]] = x1; p2[b, :n[b]] = x2
        st[b, :n[b]] = rng.random(n[b]) < float(rng.choice([1.0, 0.9, 0.5]))
    thr = float(rng.choice([0.5, 1.0, 3.0])); conf = float(rng.choice([0.9, 0.99, 0.999])); iters = int(rng.choice([50, 300, 1000]))
    seed = int(rng.integers(0, 1 << 31))
    d_st = torch.from_numpy(st).to(cuda)
    F, inl = capi.fundamental_ransac(torch.from_numpy(p1).to(cuda), torch.from_numpy(p2).to(cuda), d_st, torch.from_numpy(n).to(cuda),
                                     thr, conf, iters, seed=seed)
    F = F.cpu().numpy(); inl = inl.cpu().numpy(); got = d_st.cpu().numpy()
    for b in range(B):
        rF, rst, rc = ora.fundamental_ransac(p1[b, :n[b]], p2[b, :n[b]], st[b, :n[b]], thr, conf, iters, seed=seed)
        tag = f"ransac b {b} n {n[b]} thr {thr} conf {conf} iters {iters} seed {seed}"
        assert inl[b] == rc, f"{tag}: {inl[b]} vs {rc}"
        assert np.array_equal(got[b, :n[b]], rst), tag
        assert np.array_equal(F[b].view(np.uint64), rF.view(np.uint64)), tag


@pytest.mark.parametrize("case", range(8 * SCALE))
def test_fuzz_pyramid_three_levels_per_launch(cuda, case, monkeypatch):
    """Random geometries the one-launch pyramid takes (width a multiple of 16, height of 8, level 3 at least 48 x 34), a few
    frames each: levels 1 to 3 and their reflect-101 borders against the oracle's pyrDown chain -- every band split, strip
    count and last-strip width the sizes produce."""
    monkeypatch.setenv("VFE_PYR3_MIN_BLOCKS", "1")
    rng = np.random.default_rng(7000 + case)
    w, h = 16 * int(rng.integers(24, 150)), 8 * int(rng.integers(34, 160))
    imgs = _images(rng, w, h, int(rng.integers(1, 4)))
    lv = capi.build_pyramid(capi.alloc_pyramid(_dev(imgs, cuda), 4))
    B = capi.LK_BORDER
    for i, im in enumerate(imgs):
        ref = im
        for l in range(1, 4):
            ref = ora.pyr_down(ref)
            t = lv[l]
            b, lh, lw = t.shape
            full = t.as_strided((b, lh + 2 * B, lw + 2 * B), t.stride(), t.storage_offset() - B * t.stride(1) - B)
            assert np.array_equal(full[i].cpu().numpy(), np.pad(ref, B, mode="reflect")), f"{w}x{h} level {l} frame {i}"
