"""ctypes binding of the C ABI in include/vfe.h (vi_slam_amd/lib/libvfe.so).

PyTorch is only plumbing here: it owns device memory and streams; every computation goes
through the C entry points.  There is no CPU fallback -- a missing library or GPU raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VFE_LIB_PATH: a variant build of the same library (kernel experiments); the default is the in-tree build
LIB_PATH = os.environ.get("VFE_LIB_PATH") or os.path.join(_HERE, "lib", "libvfe.so")
_LIB = None

VFE_MAX_LEVELS = 16


class VfeError(RuntimeError):
    pass


class DImg(C.Structure):
    """struct vfe_dimg"""
    _fields_ = [("data", C.c_void_p), ("width", C.c_int), ("height", C.c_int), ("pitch", C.c_int),
                ("batch", C.c_int), ("frame_stride", C.c_size_t), ("border", C.c_int)]


class OrbKp(C.Structure):
    """struct vfe_orb_kp"""
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int), ("lx", C.c_int), ("ly", C.c_int)]


ORB_KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                         ("response", "<f4"), ("octave", "<i4"), ("lx", "<i4"), ("ly", "<i4")])


class RejectOpts(C.Structure):
    _fields_ = [("f_threshold", C.c_double), ("confidence", C.c_double), ("max_iters", C.c_int), ("seed", C.c_uint),
                ("camera4", C.POINTER(C.c_double)), ("dist5", C.POINTER(C.c_double)), ("virtual_focal", C.c_double),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class Config(C.Structure):
    """struct vfe_config"""
    _fields_ = [("device", C.c_int), ("max_width", C.c_int), ("max_height", C.c_int), ("max_features", C.c_int),
                ("pyramid_levels", C.c_int), ("win_size", C.c_int), ("orb_levels", C.c_int)]


class FrameStats(C.Structure):
    """struct vfe_frame_stats"""
    _fields_ = [("n_tracked", C.c_int), ("n_detected", C.c_int), ("n_total", C.c_int)]


# name -> (restype, argtypes); every symbol include/vfe.h declares
_P = C.c_void_p
_DI = C.POINTER(DImg)
SIGNATURES = {
    "vfe_status_string": (C.c_char_p, [C.c_int]),
    "vfe_last_error": (C.c_char_p, []),
    "vfe_version": (C.c_char_p, []),
    "vfe_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "vfe_dev_bgr2gray": (C.c_int, [_DI, _DI, _P]),
    "vfe_dev_pyr_down": (C.c_int, [_DI, _DI, _P]),
    "vfe_dev_build_pyramid": (C.c_int, [_DI, C.c_int, _P]),
    "vfe_lk_max_level": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "vfe_good_features_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vfe_dev_good_features": (C.c_int, [_DI, C.c_int, C.c_double, C.c_double, _P, _P, C.c_int, _P, _P, _P, _P,
                                        C.c_size_t, _P]),
    "vfe_dev_min_eigen_map": (C.c_int, [_DI, _P, _P]),
    "vfe_dev_klt_track": (C.c_int, [_DI, _DI, C.c_int, _P, _P, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_double,
                                    C.c_double, C.c_int, _P]),
    "vfe_klt_track_workspace": (C.c_size_t, [C.c_int, C.c_int]),
    "vfe_dev_klt_track_ws": (C.c_int, [_DI, _DI, C.c_int, _P, _P, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_double,
                                       C.c_double, C.c_int, _P, C.c_size_t, _P]),
    "vfe_dev_klt_track_bidir": (C.c_int, [_DI, _DI, C.c_int, _P, _P, C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int,
                                          C.c_double, C.c_double, C.c_double, _P]),
    "vfe_dev_undistort_points": (C.c_int, [_P, _P, C.c_longlong, C.POINTER(C.c_double), C.POINTER(C.c_double), _P]),
    "vfe_dev_corner_subpix": (C.c_int, [_DI, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _P]),
    "vfe_fundamental_workspace": (C.c_size_t, [C.c_int]),
    "vfe_dev_fundamental_ransac": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, C.c_double, C.c_double, C.c_int, C.c_uint, _P, _P,
                                             _P, C.c_size_t, _P]),
    "vfe_dev_compact_tracks": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "vfe_dev_append_points": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, C.c_int, _P]),
    "vfe_frontend_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "vfe_reject_workspace": (C.c_size_t, [C.c_int, C.c_int]),
    "vfe_dev_frontend_step_reject": (C.c_int, [C.POINTER(DImg), C.POINTER(DImg), C.c_int, _P, _P, _P, C.c_int, C.c_double, C.c_double,
                                               C.c_int, C.c_int, C.c_double, C.c_double, _P, C.c_size_t, _P, _P]),
    "vfe_set_f_rejection": (C.c_int, [_P, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double]),
    "vfe_dev_frontend_step": (C.c_int, [_DI, _DI, C.c_int, _P, _P, _P, C.c_int, C.c_double, C.c_double, C.c_int,
                                        C.c_int, C.c_double, C.c_double, _P, C.c_size_t, _P]),
    "vfe_set_overlap": (C.c_int, [C.c_int]),
    "vfe_dev_frontend_candidates": (C.c_int, [_DI, C.c_int, C.c_int, C.c_double, C.c_double, _P, C.c_size_t, _P]),
    "vfe_dev_frontend_track": (C.c_int, [_DI, _DI, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                         _P, C.c_size_t, _P, _P]),
    "vfe_dev_frontend_refill": (C.c_int, [_DI, _P, _P, _P, C.c_int, C.c_double, C.c_double, _P, C.c_size_t, _P]),
    "vfe_frontend_fast_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vfe_dev_frontend_step_fast": (C.c_int, [_DI, _DI, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_double, C.c_double, _P, C.c_size_t, _P]),
    "vfe_set_detector": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vfe_prof_enable": (C.c_int, [C.c_int]),
    "vfe_prof_read": (C.c_int, [_P, _P, C.c_int]),
    "vfe_prof_kernel_name": (C.c_char_p, [C.c_int]),
    "vfe_fast_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "vfe_dev_fast9": (C.c_int, [_DI, C.c_int, _P, C.c_int, _P, _P, C.c_size_t, _P]),
    "vfe_fast_grid_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vfe_dev_fast_grid": (C.c_int, [_DI, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P,
                                    C.c_int, _P, _P, C.c_size_t, _P]),
    "vfe_orb_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]),
    "vfe_orb_level_sizes": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_int, _P, _P]),
    "vfe_orb_quotas": (C.c_int, [C.c_int, C.c_float, C.c_int, _P]),
    "vfe_dev_orb_extract": (C.c_int, [_DI, C.c_int, C.c_float, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P,
                                      C.c_size_t, _P]),
    "vfe_match_workspace": (C.c_size_t, [C.c_int, C.c_longlong]),
    "vfe_dev_match_knn2": (C.c_int, [_P, C.c_int, _P, C.c_longlong, C.c_int, _P, _P, _P, C.c_size_t, _P]),
    "vfe_dev_match_knn2_noalloc": (C.c_int, [_P, C.c_int, _P, C.c_longlong, C.c_int, _P, _P, _P]),
    "vfe_dev_match_knn2_batch": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, C.c_int, _P, _P, C.c_float, _P, _P]),
    "vfe_dev_merge_knn2": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P]),
    "vfe_dev_ratio_test": (C.c_int, [_P, _P, C.c_int, C.c_float, _P, _P]),
    "vfe_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "vfe_destroy": (None, [_P]),
    "vfe_reset": (C.c_int, [_P]),
    "vfe_ctx_set_overlap": (C.c_int, [_P, C.c_int]),
    "vfe_ctx_prof_enable": (C.c_int, [_P, C.c_int]),
    "vfe_ctx_prof_read": (C.c_int, [_P, _P, _P, C.c_int]),
    "vfe_set_equalize": (C.c_int, [_P, C.c_int]),
    "vfe_equalize_workspace": (C.c_size_t, [C.c_int]),
    "vfe_dev_equalize_hist": (C.c_int, [_DI, _DI, _P, C.c_size_t, _P]),
    "vfe_clahe_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "vfe_dev_clahe": (C.c_int, [_DI, _DI, C.c_double, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "vfe_detect_shi_tomasi": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_double,
                                        C.c_double, _P, C.POINTER(C.c_int)]),
    "vfe_klt_track": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_size_t, _P, C.c_int, _P, _P, _P, C.c_int, C.c_int,
                                C.c_double, C.c_double]),
    "vfe_orb_extract": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_float, C.c_int,
                                  C.c_int, _P, _P, C.c_int, C.POINTER(C.c_int)]),
    "vfe_match_knn2": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.c_float, _P, _P, _P]),
    "vfe_process_frame": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int64, C.c_int, C.c_double,
                                    C.c_double, C.POINTER(FrameStats)]),
    "vfe_get_points": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int)]),
}


def lib():
    """Load libvfe.so (built in-tree by `make`).  Raises if it is missing: there is no fallback."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise VfeError(f"{LIB_PATH} not found: build it with `make` (hipcc --offload-arch=gfx950)")
        L = C.CDLL(LIB_PATH)
        missing = [name for name in SIGNATURES if not hasattr(L, name)]
        if missing:
            raise VfeError(f"{LIB_PATH} does not export {missing}: stale build, run `make`")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        L = lib()
        raise VfeError(f"{L.vfe_status_string(rc).decode()}: {L.vfe_last_error().decode()}")


def dimg(t):
    """vfe_dimg view of a torch uint8 CUDA tensor [B, H, W] (row stride may exceed W).  Tensors from
    alloc_image(..., border=n) carry their reflect-101 border width in `_vfe_border`; views derived
    from them lose the tag and are described as unpadded, which is always safe."""
    assert t.is_cuda and t.dim() == 3 and t.element_size() == 1 and t.stride(2) == 1
    b, h, w = t.shape
    return DImg(t.data_ptr(), w, h, t.stride(1), b, max(t.stride(0), t.stride(1) * h), getattr(t, "_vfe_border", 0))


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


# ---- thin helpers over the device entry points (torch tensors in, torch tensors out) ----------

LK_BORDER = 32  # VFE_LK_BORDER


def alloc_image(batch, h, w, device="cuda", align=256, border=0):
    """uint8 [B, H, W] view whose rows and frames start on `align`-byte boundaries.  With `border`
    the view sits inside a buffer with that many spare pixels on every side (the reflect-101 margin
    pyramid levels carry for the tracker)."""
    import torch
    pitch = (w + 2 * border + align - 1) // align * align
    buf = torch.empty((batch, h + 2 * border, pitch), dtype=torch.uint8, device=device)
    view = buf[:, border:border + h, border:border + w]
    view._vfe_border = border
    return view


def pyr_down(src, dst=None):
    b, h, w = src.shape
    if dst is None:
        dst = alloc_image(b, (h + 1) // 2, (w + 1) // 2, src.device)
    check(lib().vfe_dev_pyr_down(C.byref(dimg(src)), C.byref(dimg(dst)), _stream()))
    return dst


def lk_max_level(w, h, win, max_level):
    return lib().vfe_lk_max_level(w, h, win, max_level)


def alloc_pyramid(img, nlevels):
    lv = [img]
    for _ in range(1, nlevels):
        b, h, w = lv[-1].shape
        lv.append(alloc_image(b, (h + 1) // 2, (w + 1) // 2, img.device, border=LK_BORDER))
    return lv


def build_pyramid(levels):
    """Fill levels[1:] from levels[0] (tensors from alloc_pyramid)."""
    arr = (DImg * len(levels))(*[dimg(t) for t in levels])
    check(lib().vfe_dev_build_pyramid(arr, len(levels), _stream()))
    return levels


class GoodFeatures:
    """Reusable buffers for vfe_dev_good_features on [B, H, W] batches."""

    def __init__(self, batch, h, w, max_corners, seed_cap=0, device="cuda"):
        import torch
        self.batch, self.h, self.w, self.max_corners, self.seed_cap = batch, h, w, max_corners, seed_cap
        nbytes = lib().vfe_good_features_workspace(w, h, batch, max_corners, seed_cap)
        if nbytes == 0:
            raise VfeError("vfe_good_features_workspace: invalid shape")
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.ws_bytes = nbytes
        self.corners = torch.zeros((batch, max_corners, 2), dtype=torch.float32, device=device)
        self.scores = torch.zeros((batch, max_corners), dtype=torch.float32, device=device)
        self.counts = torch.zeros(batch, dtype=torch.int32, device=device)

    def __call__(self, img, quality, min_distance, seeds=None, n_seeds=None):
        assert tuple(img.shape) == (self.batch, self.h, self.w)
        check(lib().vfe_dev_good_features(C.byref(dimg(img)), self.max_corners, quality, min_distance,
                                          _ptr(seeds), _ptr(n_seeds), self.seed_cap if seeds is not None else 0,
                                          _ptr(self.corners), _ptr(self.scores), _ptr(self.counts),
                                          _ptr(self.ws), self.ws_bytes, _stream()))
        return self.corners, self.scores, self.counts


def min_eigen_map(img):
    import torch
    b, h, w = img.shape
    out = torch.empty((b, h, w), dtype=torch.float32, device=img.device)
    check(lib().vfe_dev_min_eigen_map(C.byref(dimg(img)), _ptr(out), _stream()))
    return out


def klt_track(prev_levels, next_levels, prev_pts, n_pts=None, next_pts=None, status=None, err=None,
              win=21, max_iter=30, eps=0.01, min_eig_thr=1e-4, use_initial_flow=False):
    """prev_pts [B, cap, 2] float32 cuda; returns (next_pts, status, err)."""
    import torch
    b, cap, _ = prev_pts.shape
    if next_pts is None:
        next_pts = torch.zeros_like(prev_pts)
    if status is None:
        status = torch.zeros((b, cap), dtype=torch.uint8, device=prev_pts.device)
    if err is None:
        err = torch.zeros((b, cap), dtype=torch.float32, device=prev_pts.device)
    n = len(prev_levels)
    pa = (DImg * n)(*[dimg(t) for t in prev_levels])
    na = (DImg * n)(*[dimg(t) for t in next_levels])
    check(lib().vfe_dev_klt_track(pa, na, n, _ptr(prev_pts), _ptr(n_pts), cap, _ptr(next_pts), _ptr(status),
                                  _ptr(err), win, max_iter, eps, min_eig_thr, 1 if use_initial_flow else 0, _stream()))
    return next_pts, status, err


def klt_track_ws(prev_levels, next_levels, prev_pts, n_pts, next_pts, status, err=None, win=21, max_iter=30, eps=0.01,
                 min_eig_thr=1e-4, use_initial_flow=False, workspace=None):
    """vfe_dev_klt_track_ws: the tracker with a workspace for its own processing order (same results)."""
    import torch
    b, cap, _ = prev_pts.shape
    if workspace is None:
        workspace = torch.empty(lib().vfe_klt_track_workspace(cap, b), dtype=torch.uint8, device=prev_pts.device)
    n = len(prev_levels)
    pa = (DImg * n)(*[dimg(t) for t in prev_levels])
    na = (DImg * n)(*[dimg(t) for t in next_levels])
    check(lib().vfe_dev_klt_track_ws(pa, na, n, _ptr(prev_pts), _ptr(n_pts), cap, _ptr(next_pts), _ptr(status), _ptr(err),
                                     win, max_iter, eps, min_eig_thr, 1 if use_initial_flow else 0,
                                     _ptr(workspace), workspace.numel(), _stream()))
    return next_pts, status, err


def klt_track_no_err(prev_levels, next_levels, prev_pts, n_pts, next_pts, status, win=21, max_iter=30, eps=0.01, min_eig_thr=1e-4):
    """vfe_dev_klt_track without the err output (NULL): the tracker then skips the residual pass at the final position."""
    n = len(prev_levels)
    pa = (DImg * n)(*[dimg(t) for t in prev_levels])
    na = (DImg * n)(*[dimg(t) for t in next_levels])
    check(lib().vfe_dev_klt_track(pa, na, n, _ptr(prev_pts), _ptr(n_pts), prev_pts.shape[1], _ptr(next_pts), _ptr(status),
                                  None, win, max_iter, eps, min_eig_thr, 0, _stream()))
    return next_pts, status


def klt_track_bidir(prev_levels, next_levels, prev_pts, n_pts=None, max_recovered_dist2=0.04,
                    win=21, max_iter=30, eps=0.01, min_eig_thr=1e-4):
    """Forward track + backward check; returns (next_pts, status, err, back_pts)."""
    import torch
    b, cap, _ = prev_pts.shape
    next_pts = torch.zeros_like(prev_pts)
    back = torch.zeros_like(prev_pts)
    status = torch.zeros((b, cap), dtype=torch.uint8, device=prev_pts.device)
    bstat = torch.zeros((b, cap), dtype=torch.uint8, device=prev_pts.device)
    err = torch.zeros((b, cap), dtype=torch.float32, device=prev_pts.device)
    n = len(prev_levels)
    pa = (DImg * n)(*[dimg(t) for t in prev_levels])
    na = (DImg * n)(*[dimg(t) for t in next_levels])
    check(lib().vfe_dev_klt_track_bidir(pa, na, n, _ptr(prev_pts), _ptr(n_pts), cap, _ptr(next_pts), _ptr(status), _ptr(err),
                                        _ptr(back), _ptr(bstat), win, max_iter, eps, min_eig_thr, max_recovered_dist2, _stream()))
    return next_pts, status, err, back


def undistort_points(pts, camera, dist):
    """pts [..., 2] float32 cuda; camera (fx, fy, cx, cy); dist (k1, k2, p1, p2, k3)."""
    import torch
    out = torch.empty_like(pts)
    cam = (C.c_double * 4)(*camera)
    d5 = (C.c_double * 5)(*dist)
    check(lib().vfe_dev_undistort_points(_ptr(pts.contiguous()), _ptr(out), pts.numel() // 2, cam, d5, _stream()))
    return out


def corner_subpix(img, pts, n_pts=None, win=5, zero_zone=-1, max_iter=30, eps=0.01):
    """cv::cornerSubPix on a [B, H, W] uint8 cuda batch; pts [B, cap, 2] float32 cuda, refined in place."""
    b, cap, _ = pts.shape
    assert pts.is_contiguous() and b == img.shape[0]
    check(lib().vfe_dev_corner_subpix(C.byref(dimg(img)), _ptr(pts), _ptr(n_pts) if n_pts is not None else None, cap, win,
                                      zero_zone, max_iter, float(eps), _stream()))
    return pts


def fundamental_ransac(pts1, pts2, status, n_pts=None, threshold=1.0, confidence=0.99, max_iters=1000, seed=0x5EED):
    """Epipolar outlier rejection; pts1/pts2 [B, cap, 2] float32 cuda, status [B, cap] uint8 cuda (in/out).
    -> (F [B, 3, 3] float64, inliers [B] int32)"""
    import torch
    b, cap, _ = pts1.shape
    F = torch.zeros((b, 3, 3), dtype=torch.float64, device=pts1.device)
    inl = torch.zeros(b, dtype=torch.int32, device=pts1.device)
    nbytes = lib().vfe_fundamental_workspace(b)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=pts1.device)
    check(lib().vfe_dev_fundamental_ransac(_ptr(pts1), _ptr(pts2), _ptr(n_pts) if n_pts is not None else None, cap, b,
                                           _ptr(status), float(threshold), float(confidence), int(max_iters), int(seed),
                                           _ptr(F), _ptr(inl), _ptr(ws), nbytes, _stream()))
    return F, inl


def compact_tracks(pts, status, n_pts, width, height, border=1):
    import torch
    b, cap, _ = pts.shape
    out = torch.zeros_like(pts)
    out_n = torch.zeros(b, dtype=torch.int32, device=pts.device)
    check(lib().vfe_dev_compact_tracks(_ptr(pts), _ptr(status), _ptr(n_pts), cap, b, width, height, border,
                                       _ptr(out), _ptr(out_n), _stream()))
    return out, out_n


class FrontendStep:
    """Buffers + call wrapper for vfe_dev_frontend_step on `batch` streams of h x w frames."""

    def __init__(self, batch, h, w, max_corners, nlevels=4, win=21, max_iter=30, eps=0.01, min_eig_thr=1e-4,
                 quality=0.01, min_distance=20.0, device="cuda"):
        import torch
        self.batch, self.h, self.w, self.cap, self.nlevels = batch, h, w, max_corners, nlevels
        self.win, self.max_iter, self.eps, self.min_eig_thr = win, max_iter, eps, min_eig_thr
        self.quality, self.min_distance = quality, min_distance
        nbytes = lib().vfe_frontend_workspace(w, h, batch, max_corners)
        if nbytes == 0:
            raise VfeError("vfe_frontend_workspace: invalid shape")
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.ws_bytes = nbytes
        self.pts = torch.zeros((batch, max_corners, 2), dtype=torch.float32, device=device)
        self.n_pts = torch.zeros(batch, dtype=torch.int32, device=device)
        self.n_tracked = torch.zeros(batch, dtype=torch.int32, device=device)

    def set_f_rejection(self, threshold, camera=None, dist=None, virtual_focal=460.0, confidence=0.99, max_iters=1000, seed=0x5EED):
        """Epipolar outlier filter between tracking and refill (threshold <= 0 turns it off)."""
        import torch
        if not threshold or threshold <= 0:
            self.reject = None
            return
        nbytes = lib().vfe_reject_workspace(self.batch, self.cap)
        if nbytes == 0:
            raise VfeError("vfe_reject_workspace: at most 2048 points per frame")
        self.rj_ws = torch.empty(nbytes, dtype=torch.uint8, device=self.pts.device)
        self.rj_cam = (C.c_double * 4)(*camera) if camera is not None else None
        self.rj_dist = (C.c_double * 5)(*dist) if camera is not None else None
        self.reject = RejectOpts(float(threshold), float(confidence), int(max_iters), int(seed),
                                 C.cast(self.rj_cam, C.POINTER(C.c_double)) if camera is not None else None,
                                 C.cast(self.rj_dist, C.POINTER(C.c_double)) if camera is not None else None,
                                 float(virtual_focal), self.rj_ws.data_ptr(), nbytes)

    def step(self, prev_levels, next_levels):
        n = self.nlevels
        pa = (DImg * n)(*[dimg(t) for t in prev_levels])
        na = (DImg * n)(*[dimg(t) for t in next_levels])
        if getattr(self, "reject", None) is not None:
            check(lib().vfe_dev_frontend_step_reject(pa, na, n, _ptr(self.pts), _ptr(self.n_pts), _ptr(self.n_tracked), self.cap,
                                                     self.quality, self.min_distance, self.win, self.max_iter, self.eps,
                                                     self.min_eig_thr, _ptr(self.ws), self.ws_bytes, C.byref(self.reject), _stream()))
            return self.pts, self.n_pts, self.n_tracked
        check(lib().vfe_dev_frontend_step(pa, na, n, _ptr(self.pts), _ptr(self.n_pts), _ptr(self.n_tracked), self.cap,
                                          self.quality, self.min_distance, self.win, self.max_iter, self.eps,
                                          self.min_eig_thr, _ptr(self.ws), self.ws_bytes, _stream()))
        return self.pts, self.n_pts, self.n_tracked


class FrontendPipeline(FrontendStep):
    """The front-end step stage by stage (vfe_dev_build_pyramid, vfe_dev_frontend_candidates, vfe_dev_frontend_track,
    vfe_dev_frontend_refill) with what a pipeline needs to run the first two stages of frame t + 1 beside the last two of
    frame t: a second workspace, a side stream and the events that order the stages.  advance(prev_levels, next_levels,
    coming_levels) tracks and refills for next_levels on the current stream and enqueues pyramid and candidates for
    coming_levels (whose level 0 holds the frame after next; None: nothing is coming) on the side stream.  Results are
    those of FrontendStep.step; the caller keeps three pyramids in rotation."""

    def __init__(self, *args, **kw):
        import torch
        super().__init__(*args, **kw)
        dev = self.pts.device
        self.wss = [self.ws, torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)]
        self.side = torch.cuda.Stream(device=dev)
        self.pyramid_done = [torch.cuda.Event(), torch.cuda.Event()]     # per workspace slot
        self.candidates_done = [torch.cuda.Event(), torch.cuda.Event()]
        self.consumed = [torch.cuda.Event(), torch.cuda.Event()]         # refill done: the slot's workspace is free again
        self.slot = 0  # workspace of the frame the next advance() consumes
        self.primed = False

    def _first_stages(self, levels, slot, after):
        import torch
        n = self.nlevels
        na = (DImg * n)(*[dimg(t) for t in levels])
        self.side.wait_event(after)
        self.side.wait_event(self.consumed[slot])  # the workspace's previous reader
        with torch.cuda.stream(self.side):
            check(lib().vfe_dev_build_pyramid(na, n, _stream()))
            self.pyramid_done[slot].record()
            check(lib().vfe_dev_frontend_candidates(na, n, self.cap, self.quality, self.min_distance, _ptr(self.wss[slot]),
                                                    self.ws_bytes, _stream()))
            self.candidates_done[slot].record()

    def advance(self, prev_levels, next_levels, coming_levels=None):
        import torch
        cur = torch.cuda.current_stream()
        here = torch.cuda.Event()
        here.record()
        if not self.primed:
            self._first_stages(next_levels, self.slot, here)
        if coming_levels is not None:
            # the coming frame's buffers were last read by the step before this one, which precedes `here`
            self._first_stages(coming_levels, self.slot ^ 1, here)
        n = self.nlevels
        ws = self.wss[self.slot]
        pa = (DImg * n)(*[dimg(t) for t in prev_levels])
        na = (DImg * n)(*[dimg(t) for t in next_levels])
        # The tracker needs the pyramid only, but it is held until the candidates are there as well: started earlier it
        # fills the chip before the side stream's eigenvalue pass has run, which then ends up behind it, at the serial
        # end of the step (64 streams: 0.68 against 0.55 ms per step).
        cur.wait_event(self.pyramid_done[self.slot])
        cur.wait_event(self.candidates_done[self.slot])
        rj = C.byref(self.reject) if getattr(self, "reject", None) is not None else None
        check(lib().vfe_dev_frontend_track(pa, na, n, _ptr(self.pts), _ptr(self.n_pts), self.cap, self.win, self.max_iter, self.eps,
                                           self.min_eig_thr, _ptr(ws), self.ws_bytes, rj, _stream()))
        check(lib().vfe_dev_frontend_refill(na, _ptr(self.pts), _ptr(self.n_pts), _ptr(self.n_tracked), self.cap, self.quality,
                                            self.min_distance, _ptr(ws), self.ws_bytes, _stream()))
        self.consumed[self.slot].record()
        self.slot ^= 1
        self.primed = coming_levels is not None
        return self.pts, self.n_pts, self.n_tracked


class FrontendStepFast:
    """Buffers + call wrapper for vfe_dev_frontend_step_fast (KLT + FAST-grid refill) on `batch` streams of h x w frames."""

    def __init__(self, batch, h, w, max_corners, nlevels=4, win=21, fast_threshold=20, grid_x=5, grid_y=5, min_px_dist=10,
                 max_iter=30, eps=0.01, min_eig_thr=1e-4, device="cuda"):
        import torch
        self.batch, self.h, self.w, self.cap, self.nlevels = batch, h, w, max_corners, nlevels
        self.win, self.max_iter, self.eps, self.min_eig_thr = win, max_iter, eps, min_eig_thr
        self.det = (fast_threshold, grid_x, grid_y, min_px_dist)
        nbytes = lib().vfe_frontend_fast_workspace(w, h, batch, max_corners, grid_x, grid_y, min_px_dist)
        if nbytes == 0:
            raise VfeError("vfe_frontend_fast_workspace: invalid shape")
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.ws_bytes = nbytes
        self.pts = torch.zeros((batch, max_corners, 2), dtype=torch.float32, device=device)
        self.n_pts = torch.zeros(batch, dtype=torch.int32, device=device)
        self.n_tracked = torch.zeros(batch, dtype=torch.int32, device=device)

    def step(self, prev_levels, next_levels):
        n = self.nlevels
        pa = (DImg * n)(*[dimg(t) for t in prev_levels])
        na = (DImg * n)(*[dimg(t) for t in next_levels])
        thr, gx, gy, mpd = self.det
        check(lib().vfe_dev_frontend_step_fast(pa, na, n, _ptr(self.pts), _ptr(self.n_pts), _ptr(self.n_tracked), self.cap, thr, gx, gy,
                                               mpd, self.win, self.max_iter, self.eps, self.min_eig_thr, _ptr(self.ws), self.ws_bytes,
                                               _stream()))
        return self.pts, self.n_pts, self.n_tracked


K_NAMES = None


def set_overlap(on):
    """Detector beside the tracker on a helper stream (default) or everything in order on the caller's stream."""
    check(lib().vfe_set_overlap(1 if on else 0))


def prof_enable(on=True):
    check(lib().vfe_prof_enable(1 if on else 0))


def prof_slots():
    """Number of kernel-timing slots (vfe_prof_kernel_name answers "?" past the last one)."""
    L = lib()
    n = 0
    while n < 64 and L.vfe_prof_kernel_name(n) != b"?":
        n += 1
    return n


def prof_read(reset=True):
    """{kernel name: (total ms, launches)} for every kernel launched since the last reset."""
    L = lib()
    n = prof_slots()
    ms = (C.c_double * n)()
    cnt = (C.c_longlong * n)()
    check(L.vfe_prof_read(ms, cnt, 1 if reset else 0))
    return {L.vfe_prof_kernel_name(i).decode(): (ms[i], cnt[i]) for i in range(n) if cnt[i]}


def bgr2gray(bgr):
    """bgr: uint8 cuda [B, H, W*3] (rows of interleaved BGR) -> gray [B, H, W]."""
    b, h, w3 = bgr.shape
    gray = alloc_image(b, h, w3 // 3, bgr.device)
    src = DImg(bgr.data_ptr(), w3 // 3, h, bgr.stride(1), b, max(bgr.stride(0), bgr.stride(1) * h))
    check(lib().vfe_dev_bgr2gray(C.byref(src), C.byref(dimg(gray)), _stream()))
    return gray


def fast9(img, threshold, cap=None):
    """(kps [B, cap, 3] int32 (x, y, score) in raster order, counts [B])."""
    import torch
    b, h, w = img.shape
    cap = cap or (w * h // 4 + 64)
    nbytes = lib().vfe_fast_workspace(w, h, b)
    ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=img.device)
    kps = torch.zeros((b, cap, 3), dtype=torch.int32, device=img.device)
    counts = torch.zeros(b, dtype=torch.int32, device=img.device)
    check(lib().vfe_dev_fast9(C.byref(dimg(img)), threshold, _ptr(kps), cap, _ptr(counts), _ptr(ws), nbytes, _stream()))
    return kps, counts


class Orb:
    """Reusable buffers for vfe_dev_orb_extract on [B, H, W] batches."""

    def __init__(self, batch, h, w, nfeatures, scale_factor=1.2, nlevels=8, fast_threshold=20, cap=None, device="cuda"):
        import torch
        self.args = (nfeatures, scale_factor, nlevels, fast_threshold)
        self.batch, self.h, self.w = batch, h, w
        self.cap = cap or (2 * nfeatures + 256)
        nbytes = lib().vfe_orb_workspace(w, h, batch, scale_factor, nlevels)
        if nbytes == 0:
            raise VfeError("vfe_orb_workspace: invalid shape")
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.ws_bytes = nbytes
        self.kps = torch.zeros((batch, self.cap, 8), dtype=torch.int32, device=device)  # 32-byte vfe_orb_kp records
        self.desc = torch.zeros((batch, self.cap, 32), dtype=torch.uint8, device=device)
        self.counts = torch.zeros(batch, dtype=torch.int32, device=device)

    def __call__(self, img):
        assert tuple(img.shape) == (self.batch, self.h, self.w)
        nf, sf, nl, ft = self.args
        check(lib().vfe_dev_orb_extract(C.byref(dimg(img)), nf, sf, nl, ft, _ptr(self.kps), _ptr(self.desc), self.cap,
                                        _ptr(self.counts), _ptr(self.ws), self.ws_bytes, _stream()))
        return self.kps, self.desc, self.counts

    def host_keypoints(self, b):
        n = int(self.counts[b])
        raw = self.kps[b, :n].cpu().numpy()
        return raw.view(ORB_KP_DTYPE).reshape(n), self.desc[b, :n].cpu().numpy()


def match_knn2(query, train, base_index=0):
    """query [nq, 32], train [nt, 32] uint8 cuda -> (idx [nq, 2], dist [nq, 2]) int32."""
    import torch
    nq, nt = query.shape[0], train.shape[0]
    nbytes = lib().vfe_match_workspace(nq, nt)
    ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=query.device)
    idx = torch.empty((nq, 2), dtype=torch.int32, device=query.device)
    dist = torch.empty((nq, 2), dtype=torch.int32, device=query.device)
    check(lib().vfe_dev_match_knn2(_ptr(query), nq, _ptr(train) if nt else None, nt, base_index, _ptr(idx), _ptr(dist),
                                   _ptr(ws), nbytes, _stream()))
    return idx, dist


MATCH_NOALLOC_BYTES = 16 << 20  # VFE_MATCH_NOALLOC_BYTES


def match_knn2_noalloc(query, train, base_index=0, out=None):
    """vfe_dev_match_knn2_noalloc: partial lists in the library's built-in scratch; `out` = (idx, dist) to reuse."""
    import torch
    nq, nt = query.shape[0], train.shape[0]
    if out is None:
        out = (torch.empty((nq, 2), dtype=torch.int32, device=query.device),
               torch.empty((nq, 2), dtype=torch.int32, device=query.device))
    idx, dist = out
    check(lib().vfe_dev_match_knn2_noalloc(_ptr(query), nq, _ptr(train) if nt else None, nt, base_index, _ptr(idx), _ptr(dist),
                                           _stream()))
    return idx, dist


def match_knn2_batch(query, nq, train, nt, ratio=0.7):
    """query [B, qcap, 32], train [B, tcap, 32] uint8 cuda, nq/nt [B] int32 cuda ->
    (idx [B, qcap, 2], dist [B, qcap, 2] int32, good [B, qcap] uint8)."""
    import torch
    b, qcap, _ = query.shape
    tcap = train.shape[1]
    assert query.is_contiguous() and train.is_contiguous()
    idx = torch.empty((b, qcap, 2), dtype=torch.int32, device=query.device)
    dist = torch.empty((b, qcap, 2), dtype=torch.int32, device=query.device)
    good = torch.empty((b, qcap), dtype=torch.uint8, device=query.device)
    check(lib().vfe_dev_match_knn2_batch(_ptr(query), _ptr(nq), qcap, _ptr(train), _ptr(nt), tcap, b, _ptr(idx), _ptr(dist),
                                         ratio, _ptr(good), _stream()))
    return idx, dist, good


def merge_knn2(shard_idx, shard_dist):
    """[S, nq, 2] per-shard top-2 lists -> global top-2 by (dist, idx)."""
    import torch
    s, nq, _ = shard_idx.shape
    idx = torch.empty((nq, 2), dtype=torch.int32, device=shard_idx.device)
    dist = torch.empty((nq, 2), dtype=torch.int32, device=shard_idx.device)
    check(lib().vfe_dev_merge_knn2(_ptr(shard_idx.contiguous()), _ptr(shard_dist.contiguous()), s, nq, _ptr(idx), _ptr(dist), _stream()))
    return idx, dist


def ratio_test(idx, dist, ratio=0.7):
    import torch
    nq = idx.shape[0]
    good = torch.empty(nq, dtype=torch.uint8, device=idx.device)
    check(lib().vfe_dev_ratio_test(_ptr(idx), _ptr(dist), nq, ratio, _ptr(good), _stream()))
    return good


def equalize_hist(img, out=None):
    """cv::equalizeHist on a [B, H, W] uint8 cuda batch (in place when out is img)."""
    import torch
    if out is None:
        out = alloc_image(img.shape[0], img.shape[1], img.shape[2], img.device)
    nbytes = lib().vfe_equalize_workspace(img.shape[0])
    ws = torch.empty(nbytes, dtype=torch.uint8, device=img.device)
    check(lib().vfe_dev_equalize_hist(C.byref(dimg(img)), C.byref(dimg(out)), _ptr(ws), nbytes, _stream()))
    return out


def clahe(img, clip_limit=10.0, tiles_x=8, tiles_y=8, out=None):
    """cv::createCLAHE(clip_limit, (tiles_x, tiles_y)).apply on a [B, H, W] uint8 cuda batch (in place when out is img)."""
    import torch
    if out is None:
        out = alloc_image(img.shape[0], img.shape[1], img.shape[2], img.device)
    nbytes = lib().vfe_clahe_workspace(img.shape[0], tiles_x, tiles_y)
    ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=img.device)
    check(lib().vfe_dev_clahe(C.byref(dimg(img)), C.byref(dimg(out)), float(clip_limit), tiles_x, tiles_y, _ptr(ws), nbytes,
                              _stream()))
    return out


def fast_grid(img, threshold, grid_x, grid_y, num_features, min_px_dist, seeds=None, n_seeds=None):
    """(pts [B, cap, 2] float32, scores [B, cap] int32, counts [B])."""
    import torch
    b, h, w = img.shape
    cap = num_features + grid_x * grid_y + 16
    nbytes = lib().vfe_fast_grid_workspace(w, h, b, grid_x, grid_y, min_px_dist)
    ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=img.device)
    pts = torch.zeros((b, cap, 2), dtype=torch.float32, device=img.device)
    sc = torch.zeros((b, cap), dtype=torch.int32, device=img.device)
    counts = torch.zeros(b, dtype=torch.int32, device=img.device)
    check(lib().vfe_dev_fast_grid(C.byref(dimg(img)), threshold, grid_x, grid_y, num_features, min_px_dist,
                                  _ptr(seeds), _ptr(n_seeds), seeds.shape[1] if seeds is not None else 0,
                                  _ptr(pts), _ptr(sc), cap, _ptr(counts), _ptr(ws), nbytes, _stream()))
    return pts, sc, counts


class Context:
    """vfe_ctx: the host-pointer API the framework adapters use (numpy in, numpy out)."""

    def __init__(self, max_width, max_height, max_features=1000, pyramid_levels=4, win_size=21, orb_levels=8, device=0):
        self._h = C.c_void_p()
        cfg = Config(device, max_width, max_height, max_features, pyramid_levels, win_size, orb_levels)
        check(lib().vfe_create(C.byref(cfg), C.byref(self._h)))
        self.max_features = max_features

    def close(self):
        if self._h:
            lib().vfe_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _img(a):
        a = np.ascontiguousarray(a, np.uint8)
        ch = 1 if a.ndim == 2 else a.shape[2]
        return a, a.ctypes.data, a.shape[0], a.shape[1], a.strides[0], ch

    def reset(self):
        check(lib().vfe_reset(self._h))

    def set_f_rejection(self, threshold, camera=None, dist=None, virtual_focal=460.0):
        cam = (C.c_double * 4)(*camera) if camera is not None else None
        d5 = (C.c_double * 5)(*dist) if camera is not None else None
        check(lib().vfe_set_f_rejection(self._h, float(threshold), cam, d5, float(virtual_focal)))

    def set_detector(self, mode, fast_threshold=20, grid_x=5, grid_y=5):
        """0 / "shi_tomasi" (default) or 1 / "fast_grid" (OpenVINS's FAST per grid cell) behind process_frame."""
        m = {"shi_tomasi": 0, "fast_grid": 1}.get(mode, mode)
        check(lib().vfe_set_detector(self._h, int(m), fast_threshold, grid_x, grid_y))

    def set_equalize(self, mode):
        """False/0 off, True/1 cv::equalizeHist, 2 or "clahe" CLAHE (clip 10, 8 x 8 tiles)."""
        m = 2 if mode == "clahe" else int(mode)
        check(lib().vfe_set_equalize(self._h, m))

    def detect_shi_tomasi(self, img, max_corners, quality, min_distance):
        a, p, rows, cols, step, ch = self._img(img)
        out = np.zeros((max_corners, 2), np.float32)
        n = C.c_int(0)
        check(lib().vfe_detect_shi_tomasi(self._h, p, rows, cols, step, ch, max_corners, quality, min_distance,
                                          out.ctypes.data, C.byref(n)))
        return out[:n.value].copy()

    def klt_track(self, prev, nxt, pts, max_level=3, max_iter=30, eps=0.01, min_eig_thr=1e-4):
        a, pa, rows, cols, step, _ = self._img(prev)
        b, pb, _, _, step_b, _ = self._img(nxt)
        assert step == step_b
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        n = len(pts)
        out = np.zeros((n, 2), np.float32)
        st = np.zeros(n, np.uint8)
        err = np.zeros(n, np.float32)
        check(lib().vfe_klt_track(self._h, pa, pb, rows, cols, step, pts.ctypes.data, n, out.ctypes.data,
                                  st.ctypes.data, err.ctypes.data, max_level, max_iter, eps, min_eig_thr))
        return out, st, err

    def orb_extract(self, img, nfeatures, scale_factor=1.2, nlevels=8, fast_threshold=20):
        a, p, rows, cols, step, ch = self._img(img)
        cap = 2 * nfeatures + 256
        kps = np.zeros(cap, ORB_KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        check(lib().vfe_orb_extract(self._h, p, rows, cols, step, ch, nfeatures, scale_factor, nlevels, fast_threshold,
                                    kps.ctypes.data, desc.ctypes.data, cap, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def match_knn2(self, query, train, ratio=0.7):
        q = np.ascontiguousarray(query, np.uint8)
        t = np.ascontiguousarray(train, np.uint8)
        nq, nt = len(q), len(t)
        idx = np.zeros((nq, 2), np.int32)
        dist = np.zeros((nq, 2), np.int32)
        good = np.zeros(nq, np.uint8)
        check(lib().vfe_match_knn2(self._h, q.ctypes.data, nq, t.ctypes.data if nt else None, nt, ratio,
                                   idx.ctypes.data, dist.ctypes.data, good.ctypes.data))
        return idx, dist, good

    def process_frame(self, img, timestamp_ns, max_corners, quality, min_distance):
        a, p, rows, cols, step, ch = self._img(img)
        st = FrameStats()
        check(lib().vfe_process_frame(self._h, p, rows, cols, step, ch, timestamp_ns, max_corners, quality,
                                      min_distance, C.byref(st)))
        return st.n_tracked, st.n_detected, st.n_total

    def points(self):
        out = np.zeros((self.max_features, 2), np.float32)
        n = C.c_int(0)
        check(lib().vfe_get_points(self._h, out.ctypes.data, self.max_features, C.byref(n)))
        return out[:n.value].copy()
