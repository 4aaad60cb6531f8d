"""ctypes binding of oracle/liboracle.so (the CPU restatement, rt_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the package.  It is the checker, not a
rendering path of the product.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_DIR, "liboracle.so")
REF_DRIVER = os.path.join(_DIR, "_ref", "ref_driver")
_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not built (make -C oracle)")
        l = C.CDLL(LIB_PATH)
        l.orc_render.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        l.orc_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        l.orc_rng_stream.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        l.orc_kat_aabb.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double]
        l.orc_kat_node_hit.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        l.orc_kat_scatter.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        l.orc_kat_texture.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        l.orc_kat_get_ray.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib = l
    return _lib


def render(desc_ptr: int, cam, seed: int = 1, threads: int = 0):
    """Oracle render of the whole image: (linear f64 HxWx3, rgb8 HxWx3, counters dict)."""
    h, w = cam.image_height, cam.image_width
    linear = np.zeros((h, w, 3), np.float64)
    rgb8 = np.zeros((h, w, 3), np.uint8)
    cnt = (C.c_uint64 * 12)()
    rc = lib().orc_render(desc_ptr, C.addressof(cam), seed, threads, linear.ctypes.data, rgb8.ctypes.data, C.addressof(cnt))
    if rc != 0:
        raise RuntimeError(f"orc_render failed ({rc})")
    names = ("samples", "segments", "box_tests", "sphere_tests", "quad_tests", "triangle_tests", "xform_enters", "medium_tests",
             "surface_hits", "noise_calls", "texel_fetches", "rng_draws")
    return linear, rgb8, {n: int(v) for n, v in zip(names, cnt)}


def rng_stream(seed: int, pixel: int, sample: int, n: int) -> np.ndarray:
    out = np.zeros(n, np.float64)
    lib().orc_rng_stream(seed, pixel, sample, n, out.ctypes.data)
    return out
