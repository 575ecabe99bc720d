"""ctypes front for the CPU oracle (oracle/libvfe_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by the
cpu_baseline leg of bench.py -- never by the vi_slam_amd package.  The oracle restates the
OpenCV algorithms behind the reference's front-end call sites
(src/slam/adapters/openvins_adapter.cpp:70-80, basalt_adapter.cpp:42-53,
orbslam3_adapter.cpp:69-79); parity is unpinned (no OpenCV, no reference golden vectors).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(fast=False):
    target = "libvfe_oracle_fast.so" if fast else "libvfe_oracle.so"
    # the -march=native build is always redone on the machine that times it
    subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if fast else []) + [target])
    _LIBS.pop(fast, None)
    return os.path.join(_HERE, target)


def lib(fast=False):
    if fast not in _LIBS:
        name = "libvfe_oracle_fast.so" if fast else "libvfe_oracle.so"
        path = os.path.join(_HERE, name)
        override = os.environ.get("VFE_ORACLE_LIB")  # e.g. the sanitizer build (make -C oracle asan-test)
        if override and not fast:
            path = override
        elif not os.path.exists(path):
            build(fast)
        _LIBS[fast] = C.CDLL(path)
    return _LIBS[fast]


class OrbKp(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int), ("lx", C.c_int), ("ly", C.c_int)]


ORB_KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                         ("response", "<f4"), ("octave", "<i4"), ("lx", "<i4"), ("ly", "<i4")])


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def bgr2gray(bgr):
    bgr, p = _u8(bgr)
    h, w, _ = bgr.shape
    out = np.empty((h, w), np.uint8)
    lib().ora_bgr2gray(p, w, h, w * 3, _p(out, C.c_uint8), w)
    return out


def equalize_hist(img):
    img, p = _u8(img)
    h, w = img.shape
    out = np.empty((h, w), np.uint8)
    lib().ora_equalize_hist(p, w, h, w, _p(out, C.c_uint8), w)
    return out


def clahe(img, clip_limit=10.0, tiles_x=8, tiles_y=8):
    img, p = _u8(img)
    h, w = img.shape
    out = np.empty((h, w), np.uint8)
    L = lib()
    L.ora_clahe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int]
    if L.ora_clahe(p, w, h, w, _p(out, C.c_uint8), w, float(clip_limit), tiles_x, tiles_y) != 0:
        raise ValueError("ora_clahe: bad tile grid")
    return out


def pyr_down(img):
    img, p = _u8(img)
    h, w = img.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().ora_pyr_down(p, w, h, w, _p(out, C.c_uint8), out.shape[1])
    return out


def lk_max_level(w, h, win, max_level):
    return lib().ora_lk_max_level(w, h, win, max_level)


def build_pyramid(img, win=21, max_level=3):
    lv = [np.ascontiguousarray(img, np.uint8)]
    for _ in range(lk_max_level(img.shape[1], img.shape[0], win, max_level)):
        lv.append(pyr_down(lv[-1]))
    return lv


def resize_linear_exact(img, dw, dh):
    img, p = _u8(img)
    h, w = img.shape
    out = np.empty((dh, dw), np.uint8)
    lib().ora_resize_linear_exact(p, w, h, w, _p(out, C.c_uint8), dw, dh, dw)
    return out


def gauss7(img):
    img, p = _u8(img)
    h, w = img.shape
    out = np.empty((h, w), np.uint8)
    lib().ora_gauss7(p, w, h, w, _p(out, C.c_uint8), w)
    return out


def gauss7_kernel():
    k = np.zeros(7, np.int32)
    lib().ora_gauss7_kernel(_p(k, C.c_int))
    return k


def min_eigen_map(img, opt=False):
    img, p = _u8(img)
    h, w = img.shape
    out = np.empty((h, w), np.float32)
    (lib().ora_min_eigen_map_opt if opt else lib().ora_min_eigen_map)(p, w, h, w, _p(out, C.c_float))
    return out


def good_features(img, max_corners, quality, min_distance, seeds=None, fast=False, opt=False):
    """opt: the row-streaming organisation of the same arithmetic (ora_opt.cpp), bit-identical results."""
    img, p = _u8(img)
    h, w = img.shape
    cap = max_corners if max_corners > 0 else w * h
    xy = np.zeros((cap, 2), np.float32)
    sc = np.zeros(cap, np.float32)
    if seeds is not None and len(seeds):
        seeds = np.ascontiguousarray(seeds, np.float32)
        sp, ns = _p(seeds, C.c_float), len(seeds)
    else:
        sp, ns = None, 0
    L = lib(fast)
    fn = L.ora_good_features_opt if opt else L.ora_good_features
    fn.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                   C.c_double, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float),
                   C.POINTER(C.c_float), C.c_int]
    n = fn(p, w, h, w, max_corners, quality, min_distance, sp, ns, _p(xy, C.c_float), _p(sc, C.c_float), cap)
    assert n >= 0
    return xy[:n].copy(), sc[:n].copy()


def klt_track(prev, nxt, pts, win=21, max_level=3, max_iter=30, eps=0.01, min_eig_thr=1e-4,
              init=None, fast=False, want_err=True, opt=False):
    """opt: one derivative image per level and border-extended images (ora_opt.cpp), bit-identical results."""
    prev, pp = _u8(prev)
    nxt, pn = _u8(nxt)
    h, w = prev.shape
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    n = len(pts)
    out = np.zeros((n, 2), np.float32) if init is None else np.ascontiguousarray(init, np.float32).copy()
    st = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    L = lib(fast)
    fn = L.ora_klt_track_opt if opt else L.ora_klt_track
    fn.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int,
                   C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint8),
                   C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]
    fn(pp, pn, w, h, w, _p(pts, C.c_float), n, _p(out, C.c_float), _p(st, C.c_uint8),
                    _p(err, C.c_float) if want_err else None, win, max_level, max_iter, eps, min_eig_thr,
                    0 if init is None else 1)
    return out, st, err


def klt_track_bidir(prev, nxt, pts, max_recovered_dist2=0.04, win=21, max_level=3):
    """Forward track, backward track from the original positions, float32 distance test."""
    fwd, st, err = klt_track(prev, nxt, pts, win, max_level)
    back, bst, _ = klt_track(nxt, prev, fwd, win, max_level, init=np.ascontiguousarray(pts, np.float32).reshape(-1, 2))
    p = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    dx = back[:, 0] - p[:, 0]
    dy = back[:, 1] - p[:, 1]
    d2 = dx * dx + dy * dy  # float32, one rounding per operation
    ok = st.astype(bool) & bst.astype(bool) & (d2 <= np.float32(max_recovered_dist2))
    return fwd, ok.astype(np.uint8), err, back


def undistort_points(pts, camera, dist):
    p = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    out = np.empty_like(p)
    cam = (C.c_double * 4)(*camera)
    d5 = (C.c_double * 5)(*dist)
    L = lib()
    L.ora_undistort_points.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_longlong,
                                       C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ora_undistort_points.restype = None
    L.ora_undistort_points(_p(p, C.c_float), _p(out, C.c_float), len(p), cam, d5)
    return out


def corner_subpix(img, pts, win=5, zero_zone=-1, max_iter=30, eps=0.01):
    img, ip = _u8(img)
    h, w = img.shape
    p = np.array(pts, np.float32).reshape(-1, 2).copy()
    L = lib()
    L.ora_corner_subpix.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_double]
    if L.ora_corner_subpix(ip, w, h, w, _p(p, C.c_float), len(p), win, zero_zone, max_iter, float(eps)) != 0:
        raise ValueError("ora_corner_subpix: bad parameters")
    return p


def fundamental_ransac(p1, p2, status=None, threshold=1.0, confidence=0.99, max_iters=1000, seed=0x5EED):
    """-> (F [3,3] float64, status [n] uint8, inlier count)"""
    a = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
    b = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
    st = np.ones(len(a), np.uint8) if status is None else np.array(status, np.uint8).copy()
    F = np.zeros(9, np.float64)
    L = lib()
    L.ora_fundamental_ransac.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.c_int,
                                         C.c_double, C.c_double, C.c_int, C.c_uint32, C.POINTER(C.c_double)]
    cnt = L.ora_fundamental_ransac(_p(a, C.c_float), _p(b, C.c_float), _p(st, C.c_uint8), len(a), float(threshold),
                                   float(confidence), int(max_iters), int(seed), _p(F, C.c_double))
    return F.reshape(3, 3), st, cnt


def fast9(img, threshold, nms=True):
    img, p = _u8(img)
    h, w = img.shape
    cap = w * h
    k = np.zeros((cap, 3), np.int32)
    n = lib().ora_fast9(p, w, h, w, threshold, 1 if nms else 0, _p(k, C.c_int), cap)
    assert n >= 0
    return k[:n].copy()


def fast_grid(img, threshold, grid_x, grid_y, num_features, min_px_dist, seeds=None):
    img, p = _u8(img)
    h, w = img.shape
    cap = num_features + grid_x * grid_y + 16
    xy = np.zeros((cap, 2), np.float32)
    sc = np.zeros(cap, np.int32)
    if seeds is not None and len(seeds):
        seeds = np.ascontiguousarray(seeds, np.float32)
        sp, ns = _p(seeds, C.c_float), len(seeds)
    else:
        sp, ns = None, 0
    n = lib().ora_fast_grid(p, w, h, w, threshold, grid_x, grid_y, num_features, min_px_dist, sp, ns,
                            _p(xy, C.c_float), _p(sc, C.c_int), cap)
    assert n >= 0
    return xy[:n].copy(), sc[:n].copy()


def orb_level_sizes(w, h, scale_factor=1.2, nlevels=8):
    wh = np.zeros((nlevels, 2), np.int32)
    sc = np.zeros(nlevels, np.float32)
    L = lib()
    L.ora_orb_level_sizes.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float)]
    L.ora_orb_level_sizes(w, h, scale_factor, nlevels, _p(wh, C.c_int), _p(sc, C.c_float))
    return wh, sc


def orb_quotas(nfeatures, scale_factor=1.2, nlevels=8):
    q = np.zeros(nlevels, np.int32)
    L = lib()
    L.ora_orb_quotas.argtypes = [C.c_int, C.c_float, C.c_int, C.POINTER(C.c_int)]
    L.ora_orb_quotas(nfeatures, scale_factor, nlevels, _p(q, C.c_int))
    return q


def orb_extract(img, nfeatures, scale_factor=1.2, nlevels=8, fast_threshold=20, fast=False):
    img, p = _u8(img)
    h, w = img.shape
    cap = 4 * nfeatures + 1024
    kps = np.zeros(cap, ORB_KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    L = lib(fast)
    L.ora_orb_extract.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                  C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_int]
    n = L.ora_orb_extract(p, w, h, w, nfeatures, scale_factor, nlevels, fast_threshold,
                          kps.ctypes.data, _p(desc, C.c_uint8), cap)
    assert n >= 0
    return kps[:n].copy(), desc[:n].copy()


def match_knn2(query, train, base_index=0, fast=False):
    query, pq = _u8(query)
    train, pt = _u8(train)
    nq, nt = len(query), len(train)
    idx = np.zeros((nq, 2), np.int32)
    dist = np.zeros((nq, 2), np.int32)
    lib(fast).ora_match_knn2(pq, nq, pt, nt, base_index, _p(idx, C.c_int), _p(dist, C.c_int))
    return idx, dist


def ratio_test(idx, dist, ratio=0.7):
    idx = np.ascontiguousarray(idx, np.int32)
    dist = np.ascontiguousarray(dist, np.int32)
    good = np.zeros(len(idx), np.uint8)
    L = lib()
    L.ora_ratio_test.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_float, C.POINTER(C.c_uint8)]
    L.ora_ratio_test(_p(idx, C.c_int), _p(dist, C.c_int), len(idx), ratio, _p(good, C.c_uint8))
    return good
