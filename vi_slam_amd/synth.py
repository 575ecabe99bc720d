"""Seeded synthetic frames and descriptors (host-side, numpy out).

The reference's own timing fixture feeds unseeded uniform-random 640x480 frames
(tests/e2e/e2e_test_fixture.cpp:200-215); these generators are its reproducible stand-in.
"""
import ctypes as C
import os
import numpy as np

_LIB = None
BASE_SEED = 0x5EED0000
MAP_SEED = 0x4D415000


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libvfe_synth.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make` (or __graft_entry__.build()) first")
        _LIB = C.CDLL(path)
        _LIB.vfe_synth_scene.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                         C.c_void_p, C.c_int]
        _LIB.vfe_synth_noise.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_int]
        _LIB.vfe_synth_descriptors.argtypes = [C.c_uint32, C.c_longlong, C.c_longlong, C.c_void_p]
        _LIB.vfe_synth_flip_bits.argtypes = [C.c_uint32, C.c_void_p, C.c_int, C.c_void_p]
    return _LIB


def noise(seed, w, h):
    out = np.empty((h, w), np.uint8)
    _lib().vfe_synth_noise(seed & 0xFFFFFFFF, w, h, out.ctypes.data, w)
    return out


def scene(seed, w, h, t, vx=1.5, vy=0.75, rot_deg=0.1):
    out = np.empty((h, w), np.uint8)
    _lib().vfe_synth_scene(seed & 0xFFFFFFFF, w, h, t, vx, vy, rot_deg, out.ctypes.data, w)
    return out


def descriptors(seed, n, first=0):
    out = np.empty((n, 32), np.uint8)
    _lib().vfe_synth_descriptors(seed & 0xFFFFFFFF, first, n, out.ctypes.data)
    return out


def flip_bits(seed, desc, nflip):
    src = np.ascontiguousarray(desc, np.uint8)
    out = np.empty(32, np.uint8)
    _lib().vfe_synth_flip_bits(seed & 0xFFFFFFFF, src.ctypes.data, nflip, out.ctypes.data)
    return out
