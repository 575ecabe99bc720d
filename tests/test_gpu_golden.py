"""GPU results against the committed golden vectors (tests/golden/frontend_golden.npz) -- without running the CPU
oracle: the inputs are regenerated from their seeds, the HIP path runs through the C ABI and every stored output is
compared bit for bit."""
import os
import zlib

import numpy as np
import pytest

from vi_slam_amd import capi, synth

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frontend_golden.npz"), allow_pickle=False)


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def dev(imgs, cuda):
    import torch
    t = capi.alloc_image(len(imgs), imgs[0].shape[0], imgs[0].shape[1], cuda)
    t.copy_(torch.from_numpy(np.stack(imgs)).to(cuda))
    return t


@pytest.fixture(scope="module")
def frames():
    w, h = 752, 480
    return synth.scene(synth.BASE_SEED, w, h, 0), synth.scene(synth.BASE_SEED, w, h, 1), synth.noise(7, 320, 240)


def test_golden_pyramid_eig_detect(cuda, frames):
    s0, s1, nz = frames
    pyr = capi.build_pyramid(capi.alloc_pyramid(dev([s0], cuda), 4))
    assert np.array_equal(np.array([crc(p[0].cpu().numpy()) for p in pyr], np.uint32), G["pyr_crc"])
    assert np.array_equal(pyr[3][0].cpu().numpy(), G["pyr_l3"])
    e = [capi.min_eigen_map(dev([im], cuda))[0].cpu().numpy() for im in (s0, nz)]
    assert np.array_equal(np.array([crc(x) for x in e], np.uint32), G["eig_crc"])
    for tag, (n, q, d) in {"vins": (150, 0.01, 30), "openvins": (200, 0.01, 10), "basalt": (200, 0.007, 10)}.items():
        gf = capi.GoodFeatures(1, 480, 752, n, device=cuda)
        c, s, k = gf(dev([s0], cuda), q, d)
        k = int(k[0])
        assert np.array_equal(c[0, :k].cpu().numpy().astype(np.int16), G[f"gftt_{tag}_xy"]), tag
        assert np.array_equal(s[0, :k].cpu().numpy().view(np.uint32), G[f"gftt_{tag}_sc"]), tag


def test_golden_tracker_and_after(cuda, frames):
    import torch
    s0, s1, _ = frames
    xy = G["gftt_vins_xy"].astype(np.float32)
    p0 = capi.build_pyramid(capi.alloc_pyramid(dev([s0], cuda), 3))
    p1 = capi.build_pyramid(capi.alloc_pyramid(dev([s1], cuda), 3))
    d_xy = torch.from_numpy(xy[None]).to(cuda)
    nxt, st, err = capi.klt_track(p0, p1, d_xy)
    assert np.array_equal(st[0].cpu().numpy(), G["klt_status"])
    assert np.array_equal(nxt[0].cpu().numpy().view(np.uint32), G["klt_next"])
    assert np.array_equal(err[0].cpu().numpy().view(np.uint32), G["klt_err"])
    _, bst, _, back = capi.klt_track_bidir(p0, p1, d_xy, None, 0.04)
    assert np.array_equal(bst[0].cpu().numpy(), G["bidir_status"])
    assert np.array_equal(back[0].cpu().numpy().view(np.uint32), G["bidir_back"])
    sp = d_xy.clone()
    capi.corner_subpix(dev([s0], cuda), sp, None, 5, -1, 30, 0.01)
    assert np.array_equal(sp[0].cpu().numpy().view(np.uint32), G["subpix"])
    cam = (458.654, 457.296, 367.215, 248.375)
    dist5 = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0)
    assert np.array_equal(capi.undistort_points(d_xy[0], cam, dist5).cpu().numpy().view(np.uint32), G["undistort"])
    from two_view import two_view
    x1, x2, _, _ = two_view(300, 0.25, 0.3, 21)
    stt = torch.ones((1, len(x1)), dtype=torch.uint8, device=cuda)
    F, inl = capi.fundamental_ransac(torch.from_numpy(x1[None]).to(cuda), torch.from_numpy(x2[None]).to(cuda), stt, None, 1.0, 0.99,
                                     1000, seed=1234)
    assert int(inl[0]) == int(G["ransac_count"][0])
    assert np.array_equal(stt[0].cpu().numpy(), G["ransac_mask"])
    assert np.array_equal(F[0].cpu().numpy().view(np.uint64), G["ransac_F"])


def test_golden_fast_orb_match(cuda, frames):
    s0, s1, _ = frames
    kps, cnt = capi.fast9(dev([s0], cuda), 20)
    f = kps[0, :int(cnt[0])].cpu().numpy()
    assert np.array_equal(np.array([len(f), crc(f)], np.uint32), G["fast_count_crc"])
    assert np.array_equal(f[:64].astype(np.int16), G["fast_head"])
    pts, sc, k = capi.fast_grid(dev([s0], cuda), 20, 5, 5, 200, 10)
    k = int(k[0])
    assert np.array_equal(pts[0, :k].cpu().numpy().astype(np.int16), G["fast_grid_xy"])
    assert np.array_equal(sc[0, :k].cpu().numpy().astype(np.int16), G["fast_grid_sc"])
    orb = capi.Orb(2, 480, 752, 500, device=cuda)
    orb(dev([s0, s1], cuda))
    k0, d0 = orb.host_keypoints(0)
    _, d1 = orb.host_keypoints(1)
    assert np.array_equal(np.stack([k0["lx"], k0["ly"], k0["octave"]], axis=1).astype(np.int16), G["orb_lxy_oct"])
    assert np.array_equal(k0["angle"].view(np.uint32), G["orb_angle"])
    assert np.array_equal(k0["response"].view(np.uint32), G["orb_resp"])
    assert np.array_equal(d0, G["orb_desc"])
    import torch
    idx, dist = capi.match_knn2(torch.from_numpy(d0).to(cuda), torch.from_numpy(d1).to(cuda))
    assert np.array_equal(idx.cpu().numpy().astype(np.int16), G["match_idx"])
    assert np.array_equal(dist.cpu().numpy().astype(np.int16), G["match_dist"])
    assert np.array_equal(capi.ratio_test(idx, dist, 0.7).cpu().numpy(), G["match_good"])


def test_golden_preprocessing(cuda, frames):
    import torch
    s0, _, nz = frames
    rng = np.random.default_rng(1)
    bgr = rng.integers(0, 256, (32, 48, 3), dtype=np.uint8)
    assert np.array_equal(capi.bgr2gray(torch.from_numpy(bgr.reshape(1, 32, 48 * 3)).to(cuda))[0].cpu().numpy(), G["bgr2gray"])
    assert crc(capi.equalize_hist(dev([s0], cuda))[0].cpu().numpy()) == G["equalize_crc"][0]
    got = [crc(capi.clahe(dev([s0], cuda), 10.0, 8, 8)[0].cpu().numpy()),
           crc(capi.clahe(dev([np.ascontiguousarray(s0[:477, :750])], cuda), 2.0, 4, 6)[0].cpu().numpy()),
           crc(capi.clahe(dev([nz], cuda), 0.0, 8, 8)[0].cpu().numpy())]
    assert np.array_equal(np.array(got, np.uint32), G["clahe_crc"])
