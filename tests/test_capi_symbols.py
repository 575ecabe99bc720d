"""CPU: the C ABI library loads without a GPU and exports every function include/vfe.h declares;
argument validation works without touching the device."""
import ctypes as C
import os
import re

import numpy as np

from vi_slam_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "vfe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vfe_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 40
    L = C.CDLL(capi.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    unbound = [n for n in names if n not in capi.SIGNATURES]
    assert not unbound, unbound
    capi.lib()


def test_status_strings_and_version():
    L = capi.lib()
    assert L.vfe_status_string(0) == b"ok"
    assert b"device" in L.vfe_status_string(-5)
    assert b"gfx950" in L.vfe_version()


def test_workspace_queries_are_host_only():
    L = capi.lib()
    assert L.vfe_good_features_workspace(1920, 1080, 1, 1000, 1000) > 1920 * 1080 * 8
    assert L.vfe_good_features_workspace(0, 1080, 1, 1000, 0) == 0
    assert L.vfe_frontend_workspace(1920, 1080, 2, 1000) > 2 * L.vfe_good_features_workspace(1920, 1080, 1, 1000, 1000) - 4096
    assert L.vfe_fast_workspace(752, 480, 1) > 0 and L.vfe_orb_workspace(752, 480, 1, 1.2, 8) > 0
    assert L.vfe_match_workspace(2000, 2000) > 0
    # one top-2 list per (query, chunk) in two int arrays; a large train set adds the sample's lists and a floor per query,
    # and the relocalisation shape (2000 queries, 10 M rows) still fits the array of the entry point that allocates nothing
    assert L.vfe_match_workspace(0, 100) == 0 and L.vfe_match_workspace(5, 0) >= 5 * 2 * 4 * 2
    small, large = L.vfe_match_workspace(2000, 100_000), L.vfe_match_workspace(2048, 10_000_000)
    assert 2000 * 200 * 16 <= small < large <= capi.MATCH_NOALLOC_BYTES < L.vfe_match_workspace(6000, 10_000_000)
    assert L.vfe_lk_max_level(752, 480, 21, 3) == 3 and L.vfe_lk_max_level(40, 40, 21, 3) == 0


def test_argument_validation_without_gpu():
    L = capi.lib()
    bad = capi.DImg(None, 0, 0, 0, 0, 0)
    assert L.vfe_dev_pyr_down(C.byref(bad), C.byref(bad), None) == -1
    assert b"invalid" in L.vfe_last_error() or b"bad image" in L.vfe_last_error()
    q = (C.c_int * 8)()
    assert L.vfe_orb_quotas(2000, 1.2, 8, q) == 0 and list(q) == [434, 362, 302, 251, 209, 175, 145, 122]
    assert L.vfe_orb_quotas(2000, 1.2, 99, q) == -1
    cfg = capi.Config(0, 4, 4, 10, 4, 21, 0)   # image too small: rejected before any device call
    h = C.c_void_p()
    assert L.vfe_create(C.byref(cfg), C.byref(h)) == -1 and not h.value


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        return
    L = capi.lib()
    cfg = capi.Config(0, 640, 480, 100, 3, 21, 0)
    h = C.c_void_p()
    rc = L.vfe_create(C.byref(cfg), C.byref(h))
    assert rc in (-5, -2) and not h.value
    assert L.vfe_last_error() != b""
