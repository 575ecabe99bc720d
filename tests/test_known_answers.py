"""Answers that can be derived by hand, checked against BOTH the CPU oracle (always) and the HIP path (GPU tests).

The oracle is this build's restatement of the OpenCV algorithms and no OpenCV is available to pin it, so these cases
pin it -- and the kernels -- to something other than each other: the impulse response of pyrDown, the FAST score of a
constructed ring, Hamming distances by construction, and Lucas-Kanade on an image shifted by whole pixels."""
import numpy as np
import pytest

from oracle import oracle as ora
from vi_slam_amd import synth


# ---------------------------------------------------------------- expected values, derived here and nowhere else

def pyr_down_impulse_expected(h, w, y, x, amp):
    """pyrDown of amp * delta(y, x): the 5x5 kernel [1 4 6 4 1] x [1 4 6 4 1] / 256 sampled at even positions, one rounding
    (v + 128) >> 8.  The impulse sits away from the rim, so no reflection is involved."""
    k = np.array([1, 4, 6, 4, 1], np.int64)
    out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.int64)
    for oy in range(out.shape[0]):
        for ox in range(out.shape[1]):
            dy, dx = y - 2 * oy, x - 2 * ox
            if -2 <= dy <= 2 and -2 <= dx <= 2:
                out[oy, ox] = (amp * k[dy + 2] * k[dx + 2] + 128) >> 8
    return out.astype(np.uint8)


CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1),
          (-2, 2), (-1, 3)]  # (dx, dy), the Bresenham circle of radius 3


def ring_image(centre, ring, n_arc, background, size=15):
    """A `size` x `size` image of `background` with one pixel `centre` in the middle whose circle carries `ring` on
    `n_arc` consecutive positions (starting at the first) and `centre` on the others."""
    im = np.full((size, size), background, np.uint8)
    c = size // 2
    im[c, c] = centre
    for i, (dx, dy) in enumerate(CIRCLE):
        im[c + dy, c + dx] = ring if i < n_arc else centre
    return im, c


# ---------------------------------------------------------------- CPU: the oracle

def test_pyr_down_impulse_oracle():
    for (y, x, amp) in [(20, 30, 255), (21, 31, 255), (20, 31, 200), (21, 30, 77)]:
        im = np.zeros((48, 64), np.uint8)
        im[y, x] = amp
        assert np.array_equal(ora.pyr_down(im), pyr_down_impulse_expected(48, 64, y, x, amp))
    # the kernel sums to 256: a constant image stays constant, rims (reflect-101) included
    assert (ora.pyr_down(np.full((37, 53), 131, np.uint8)) == 131).all()


def test_fast_ring_score_oracle():
    """Centre 100, an arc of 9 ring pixels at 160, the other 7 circle pixels at 100: the centre is a corner for every
    threshold below 60 and for none from 60 on, so its score (the largest threshold that still fires) is 59.  An arc of 8
    never fires."""
    im, c = ring_image(100, 160, 9, 100)
    for thr in (1, 20, 59):
        k = ora.fast9(im, thr, nms=False)
        hit = [tuple(r) for r in k if r[0] == c and r[1] == c]
        assert hit == [(c, c, 59)], (thr, k)
    assert not any(r[0] == c and r[1] == c for r in ora.fast9(im, 60, nms=False))
    im8, _ = ring_image(100, 160, 8, 100)
    assert not any(r[0] == c and r[1] == c for r in ora.fast9(im8, 20, nms=False))
    # darker arc: same score by symmetry
    imd, _ = ring_image(100, 40, 9, 100)
    assert [tuple(r) for r in ora.fast9(imd, 20, nms=False) if r[0] == c and r[1] == c] == [(c, c, 59)]


def hamming_cases():
    """Queries and a train set whose distances are known by construction: train row j of query i differs in exactly
    `flips[i][j]` bits."""
    rng = np.random.default_rng(7)
    q = rng.integers(0, 256, (6, 32), dtype=np.uint8)
    flips = [[0, 3, 3, 200], [256, 255, 1, 1], [17, 16, 16, 15], [5, 5, 5, 5], [128, 64, 32, 16], [1, 0, 0, 2]]
    sets = []
    for i in range(6):
        rows = []
        for f in flips[i]:
            bits = np.unpackbits(q[i])
            pos = rng.permutation(256)[:f]
            bits[pos] ^= 1
            rows.append(np.packbits(bits))
        sets.append(np.stack(rows))
    return q, sets, flips


def expected_top2(f):
    order = sorted(range(len(f)), key=lambda j: (f[j], j))  # strict '<' while scanning upwards: ties keep the lower index
    return [order[0], order[1]], [f[order[0]], f[order[1]]]


def test_hamming_by_construction_oracle():
    q, sets, flips = hamming_cases()
    for i in range(len(q)):
        idx, dist = ora.match_knn2(q[i:i + 1], sets[i])
        ei, ed = expected_top2(flips[i])
        assert list(idx[0]) == ei and list(dist[0]) == ed, i
    # ratio test: d0 < 0.7 * d1 in float
    idx = np.array([[0, 1], [0, 1], [0, 1], [0, -1]], np.int32)
    dist = np.array([[7, 10], [6, 10], [0, 0], [3, 0x7fffffff]], np.int32)
    assert list(ora.ratio_test(idx, dist, 0.7)) == [0, 1, 0, 0]  # 7 < 7.0 is false; a missing second neighbour fails


def smooth_image(w, h):
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    v = 128 + 50 * np.sin(x / 9.0) * np.cos(y / 11.0) + 40 * np.sin((x + 2 * y) / 17.0) + 25 * np.cos((3 * x - y) / 23.0)
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def shifted_pair(w, h, sx, sy):
    """Two crops of one smooth image, the second displaced by (sx, sy) whole pixels: content at (x, y) in the first is at
    (x + sx, y + sy) in the second, exactly."""
    big = smooth_image(w + 32, h + 32)
    a = big[16:16 + h, 16:16 + w].copy()
    b = big[16 - sy:16 - sy + h, 16 - sx:16 - sx + w].copy()
    return a, b


# The iteration stops once a step is shorter than 0.01 px (OpenCV's default criterion), so the position it stops at is the
# true one to within a few thousandths, not to rounding: the bound below is half the stopping step.
LK_TOL = 5e-3
LK_POINTS = np.array([[60.0, 50.0], [100.5, 80.25], [150.0, 120.0], [200.75, 60.5], [90.0, 140.0], [230.0, 150.0]], np.float32)


def test_klt_recovers_integer_shift_oracle():
    for sx, sy in [(3, -2), (-5, 4), (0, 0), (7, 6)]:
        a, b = shifted_pair(320, 240, sx, sy)
        out, st, _ = ora.klt_track(a, b, LK_POINTS, 21, 3)
        assert st.all()
        assert np.abs(out - (LK_POINTS + np.array([sx, sy], np.float32))).max() < LK_TOL, (sx, sy, out - LK_POINTS)


def test_shi_tomasi_single_square_oracle():
    """One bright square on a dark field: four corners, nothing along the edges (min eigenvalue ~ 0 there), all four
    within two pixels of the true corners, mutually >= min distance apart."""
    im = np.full((120, 160), 30, np.uint8)
    im[40:81, 50:111] = 220
    pts, sc = ora.good_features(im, 10, 0.05, 10.0)
    assert len(pts) == 4
    want = np.array([[50, 40], [110, 40], [50, 80], [110, 80]], np.float32)
    for p in pts:
        assert np.abs(want - p).max(axis=1).min() <= 2.0
    assert (np.diff(sc) <= 0).all()


# ---------------------------------------------------------------- GPU: the HIP path gives the same hand-derived answers

def _to_dev(imgs, dev):
    import torch
    from vi_slam_amd import capi
    t = capi.alloc_image(len(imgs), imgs[0].shape[0], imgs[0].shape[1], dev)
    t.copy_(torch.from_numpy(np.stack(imgs)).to(dev))
    return t


@pytest.mark.gpu
def test_pyr_down_impulse_hip(cuda):
    from vi_slam_amd import capi
    cases = [(20, 30, 255), (21, 31, 255), (20, 31, 200), (21, 30, 77)]
    ims = []
    for (y, x, amp) in cases:
        im = np.zeros((48, 64), np.uint8)
        im[y, x] = amp
        ims.append(im)
    out = capi.pyr_down(_to_dev(ims, cuda)).cpu().numpy()
    for i, (y, x, amp) in enumerate(cases):
        assert np.array_equal(out[i], pyr_down_impulse_expected(48, 64, y, x, amp))
    assert (capi.pyr_down(_to_dev([np.full((37, 53), 131, np.uint8)], cuda)).cpu().numpy() == 131).all()


@pytest.mark.gpu
def test_fast_ring_score_hip(cuda):
    from vi_slam_amd import capi
    im, c = ring_image(100, 160, 9, 100)
    imd, _ = ring_image(100, 40, 9, 100)
    im8, _ = ring_image(100, 160, 8, 100)
    # vfe_dev_fast9 applies the 3x3 non-maximum suppression: the centre's neighbours score less (their own circles see
    # at most a few ring pixels), so the centre survives with its score
    for thr, want in ((1, True), (20, True), (59, True), (60, False)):
        kps, cnt = capi.fast9(_to_dev([im, imd, im8], cuda), thr)
        kps, cnt = kps.cpu().numpy(), cnt.cpu().numpy()
        for f in range(2):
            hit = [tuple(r) for r in kps[f, :cnt[f]] if r[0] == c and r[1] == c]
            assert hit == ([(c, c, 59)] if want else []), (thr, f, kps[f, :cnt[f]])
        assert not any(r[0] == c and r[1] == c for r in kps[2, :cnt[2]])


@pytest.mark.gpu
def test_hamming_by_construction_hip(cuda):
    import torch
    from vi_slam_amd import capi
    q, sets, flips = hamming_cases()
    for i in range(len(q)):
        idx, dist = capi.match_knn2(torch.from_numpy(q[i:i + 1].copy()).to(cuda), torch.from_numpy(sets[i]).to(cuda))
        ei, ed = expected_top2(flips[i])
        assert idx.cpu().numpy()[0].tolist() == ei and dist.cpu().numpy()[0].tolist() == ed, i
    idx = torch.tensor([[0, 1], [0, 1], [0, 1], [0, -1]], dtype=torch.int32, device=cuda)
    dist = torch.tensor([[7, 10], [6, 10], [0, 0], [3, 0x7fffffff]], dtype=torch.int32, device=cuda)
    assert capi.ratio_test(idx, dist, 0.7).cpu().numpy().tolist() == [0, 1, 0, 0]


@pytest.mark.gpu
def test_klt_recovers_integer_shift_hip(cuda):
    import torch
    from vi_slam_amd import capi
    for sx, sy in [(3, -2), (-5, 4), (0, 0), (7, 6)]:
        a, b = shifted_pair(320, 240, sx, sy)
        pa = capi.build_pyramid(capi.alloc_pyramid(_to_dev([a], cuda), 4))
        pb = capi.build_pyramid(capi.alloc_pyramid(_to_dev([b], cuda), 4))
        pts = torch.from_numpy(LK_POINTS[None].copy()).to(cuda)
        out, st, _ = capi.klt_track(pa, pb, pts)
        out, st = out.cpu().numpy()[0], st.cpu().numpy()[0]
        assert st.all()
        assert np.abs(out - (LK_POINTS + np.array([sx, sy], np.float32))).max() < LK_TOL, (sx, sy)


@pytest.mark.gpu
def test_shi_tomasi_single_square_hip(cuda):
    from vi_slam_amd import capi
    im = np.full((120, 160), 30, np.uint8)
    im[40:81, 50:111] = 220
    gf = capi.GoodFeatures(1, 120, 160, 10, device=cuda)
    c, s, n = gf(_to_dev([im], cuda), 0.05, 10.0)
    n = int(n[0])
    pts = c[0, :n].cpu().numpy()
    assert n == 4
    want = np.array([[50, 40], [110, 40], [50, 80], [110, 80]], np.float32)
    for p in pts:
        assert np.abs(want - p).max(axis=1).min() <= 2.0


@pytest.mark.gpu
def test_corner_properties_at_1080p_hip(cuda):
    """The structural promises of goodFeaturesToTrack on a full-size frame, read off the HIP output alone: at most
    max_corners, every pair at least min_distance apart, every corner inside the 1-pixel rim, scores descending and above
    quality x the best score."""
    from vi_slam_amd import capi
    w, h, n, q, d = 1920, 1080, 1000, 0.01, 20.0
    gf = capi.GoodFeatures(2, h, w, n, device=cuda)
    c, s, cnt = gf(_to_dev([synth.scene(synth.BASE_SEED + 11, w, h, 0), synth.noise(5, w, h)], cuda), q, d)
    c, s, cnt = c.cpu().numpy(), s.cpu().numpy(), cnt.cpu().numpy()
    for f in range(2):
        m = int(cnt[f])
        p, sc = c[f, :m].astype(np.float64), s[f, :m]
        assert 0 < m <= n
        assert (p[:, 0] >= 1).all() and (p[:, 0] <= w - 2).all() and (p[:, 1] >= 1).all() and (p[:, 1] <= h - 2).all()
        assert (np.diff(sc) <= 0).all() and (sc > q * sc[0]).all()
        dd = ((p[:, None, :] - p[None, :, :]) ** 2).sum(axis=2) + np.eye(m) * 1e9
        assert dd.min() >= d * d
