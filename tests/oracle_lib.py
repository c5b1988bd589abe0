"""ctypes binding of oracle/libflutl_oracle.so -- the CHECKER, never the
product.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
import this."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libflutl_oracle.so")

_lib = None


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(ORACLE_DIR, n) for n in ("flutl_oracle.c", "flutl_tor_oracle.c")]
        if (not os.path.exists(LIB_PATH) or
                os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs)):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "libflutl_oracle.so"],
                                  stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(LIB_PATH)
        u32p = np.ctypeslib.ndpointer(np.uint32, flags="C")
        i32p = np.ctypeslib.ndpointer(np.int32, flags="C")
        vp, u32, sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_size_t
        L.flutl_oracle_blend_n.argtypes = [u32, u32p, u32p, u32p, u32p, sz]
        L.flutl_oracle_over_n.argtypes = [u32p, u32p, u32p, sz]
        L.flutl_oracle_group_n.argtypes = [u32p, u32p, u32p, u32p, sz]
        L.flutl_oracle_color_rgba8888_n.argtypes = [u32p, u32p, sz]
        L.flutl_oracle_color_rgba8888.argtypes = [u32]
        L.flutl_oracle_color_rgba8888.restype = u32
        L.flutl_oracle_color.argtypes = [ctypes.c_double] * 4
        L.flutl_oracle_color.restype = u32
        L.flutl_oracle_opacity_to_alpha.argtypes = [ctypes.c_double]
        L.flutl_oracle_opacity_to_alpha.restype = u32
        L.flutl_oracle_fixed_from_double.argtypes = [ctypes.c_double]
        L.flutl_oracle_fixed_from_double.restype = ctypes.c_int32
        L.flutl_oracle_rect_coverage.argtypes = [ctypes.c_int] * 6
        L.flutl_oracle_rect_coverage.restype = u32
        L.flutl_oracle_render.argtypes = [u32, u32, vp, sz, vp, u32, vp, vp]
        L.flutl_oracle_blur.argtypes = [u32p, u32p, u32, u32, sz, i32p, ctypes.c_int]
        L.flutl_oracle_blur_kernel.argtypes = [ctypes.c_int, ctypes.c_double, i32p]
        L.flutl_tor_oracle.argtypes = [i32p, ctypes.c_int] + [ctypes.c_int] * 5 + [
            np.ctypeslib.ndpointer(np.uint8, flags="C"), ctypes.c_int]
        L.flutl_tor_polygon_box.argtypes = [i32p, ctypes.c_int, i32p]
        _lib = L
    return _lib


def color_rgba8888(c):
    c = np.ascontiguousarray(c, np.uint32)
    o = np.empty_like(c)
    lib().flutl_oracle_color_rgba8888_n(c.ravel(), o.ravel(), c.size)
    return o


def blend(rule, s, m, d):
    s, m, d = (np.ascontiguousarray(np.broadcast_to(a, np.shape(d)), np.uint32).ravel()
               for a in (s, m, d))
    o = np.empty_like(d)
    lib().flutl_oracle_blend_n(rule, s, m, d, o, d.size)
    return o


def over(s, d):
    s = np.ascontiguousarray(s, np.uint32).ravel()
    d = np.ascontiguousarray(d, np.uint32).ravel()
    o = np.empty_like(d)
    lib().flutl_oracle_over_n(s, d, o, d.size)
    return o


def group(g, alpha, d):
    g = np.ascontiguousarray(g, np.uint32).ravel()
    d = np.ascontiguousarray(d, np.uint32).ravel()
    alpha = np.ascontiguousarray(np.broadcast_to(alpha, d.shape), np.uint32).ravel()
    o = np.empty_like(d)
    lib().flutl_oracle_group_n(g, alpha, d, o, d.size)
    return o


def render(width, height, regions, prims, atlas):
    """CPU oracle frame for a display list (numpy structured arrays)."""
    regions = np.ascontiguousarray(regions)
    prims = np.ascontiguousarray(prims)
    atlas = np.ascontiguousarray(atlas, np.uint8)
    out = np.empty((height, width), np.uint32)
    rc = lib().flutl_oracle_render(
        width, height, out.ctypes.data_as(ctypes.c_void_p), width * 4,
        regions.ctypes.data_as(ctypes.c_void_p), len(regions),
        prims.ctypes.data_as(ctypes.c_void_p),
        atlas.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise RuntimeError(f"flutl_oracle_render: {rc}")
    return out


def blur(src, kernel, radius):
    src = np.ascontiguousarray(src, np.uint32)
    h, w = src.shape
    out = np.empty_like(src)
    lib().flutl_oracle_blur(src.ravel(), out.ravel(), w, h, w * 4,
                            np.ascontiguousarray(kernel, np.int32), radius)
    return out


def frac_origin(origin):
    """device origin (double) -> (integer destination of resampled pixel 0, 7-bit weight)"""
    import ctypes
    base, dist = ctypes.c_int(), ctypes.c_uint32()
    L = lib()
    L.flutl_oracle_frac_origin.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_int),
                                           ctypes.POINTER(ctypes.c_uint32)]
    L.flutl_oracle_frac_origin.restype = None
    L.flutl_oracle_frac_origin(float(origin), ctypes.byref(base), ctypes.byref(dist))
    return base.value, dist.value


def resample(src, distx, disty, box=None):
    """pixman bilinear resample of an ARGB32 image by constant 7-bit weights -> (h+1, w+1);
    box = 24.8 {x1, y1, x2, y2} of the fill it is painted through (edge coverage), or None"""
    import ctypes
    src = np.ascontiguousarray(src, np.uint32)
    h, w = src.shape
    out = np.empty((h + 1, w + 1), np.uint32)
    L = lib()
    u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
    L.flutl_oracle_resample.argtypes = [u32p, ctypes.c_uint32, ctypes.c_uint32,
                                        ctypes.c_uint32, ctypes.c_uint32,
                                        ctypes.c_void_p, u32p]
    L.flutl_oracle_resample.restype = None
    b = None if box is None else np.ascontiguousarray(box, np.int32)
    L.flutl_oracle_resample(src, w, h, int(distx), int(disty),
                            None if b is None else b.ctypes.data, out)
    return out


def fixed_24_8(d):
    """_cairo_fixed_from_double: nearest 1/256, ties to even"""
    import math
    v = d * 256.0
    f = math.floor(v)
    r = v - f
    if r > 0.5 or (r == 0.5 and (int(f) & 1)):
        f += 1
    return int(f)


def blur_kernel(radius, sigma):
    out = np.empty((2 * radius + 1) ** 2, np.int32)
    lib().flutl_oracle_blur_kernel(radius, float(sigma), out)
    return out


def scan_polygon(edges, box, even_odd=False):
    """oracle/flutl_tor_oracle.c: coverage (h x w uint8) of a polygon (n x 7 int32 edges,
    24.8 fixed point) inside box = (x0, y0, x1, y1), as Cairo 1.16's tor converter"""
    edges = np.ascontiguousarray(edges, np.int32).reshape(-1, 7)
    x0, y0, x1, y1 = (int(v) for v in box)
    out = np.zeros((y1 - y0, x1 - x0), np.uint8)
    rc = lib().flutl_tor_oracle(edges.ravel(), len(edges), x0, y0, x1, y1, int(even_odd),
                                out.ravel(), x1 - x0)
    assert rc == 0
    return out


def polygon_box(edges):
    edges = np.ascontiguousarray(edges, np.int32).reshape(-1, 7)
    box = np.zeros(4, np.int32)
    return box if lib().flutl_tor_polygon_box(edges.ravel(), len(edges), box) else None
