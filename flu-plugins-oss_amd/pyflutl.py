"""ctypes binding of the fluttml-hip C-ABI (include/flutl.h).

Plumbing for tests and bench.py only: the product is the C library
(lib/libflutl.so) and the GStreamer element in host/.  PyTorch is used by the
callers of this module for device memory and streams; nothing here computes.

The library must have been built (``python -c 'import __graft_entry__ as g;
g.build()'`` or ``make -C flu-plugins-oss_amd``); there is no fallback.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# FLUTL_LIB: profiling builds of the same library (tools/ablate.sh); never a fallback
LIB_PATH = os.environ.get("FLUTL_LIB") or os.path.join(HERE, "lib", "libflutl.so")

# ---- IR mirrors (include/flutl.h) -----------------------------------------
OP_RECT, OP_MASK, OP_IMAGE, OP_GROUP_BEGIN, OP_GROUP_END = 1, 2, 3, 4, 5
RULE_PIXMAN, RULE_LERP = 0, 1
TILE_W, TILE_H = 64, 8
ATLAS_SLACK = 16

PRIM_DTYPE = np.dtype([
    ("op", "u1"), ("rule", "u1"), ("param", "<u2"),
    ("cx0", "<i2"), ("cy0", "<i2"), ("cx1", "<i2"), ("cy1", "<i2"),
    ("color", "<u4"),
    ("a", "<i4"), ("b", "<i4"), ("c", "<i4"), ("d", "<i4"),
])
REGION_DTYPE = np.dtype([
    ("x0", "<i2"), ("y0", "<i2"), ("x1", "<i2"), ("y1", "<i2"),
    ("first", "<u4"), ("count", "<u4"),
])
FRAME_DTYPE = np.dtype([
    ("region_first", "<u4"), ("region_count", "<u4"), ("dst_offset", "<u8"),
])
assert PRIM_DTYPE.itemsize == 32 and REGION_DTYPE.itemsize == 16
assert FRAME_DTYPE.itemsize == 16

EXPORTS = [
    "flutl_abi_version", "flutl_device_count", "flutl_strerror",
    "flutl_list_validate", "flutl_launch_frame", "flutl_launch_batch",
    "flutl_launch_batch_over",
    "flutl_launch_blur", "flutl_blur_kernel", "flutl_launch_resample",
    "flutl_launch_rotate_cw",
    "flutl_ctx_create", "flutl_ctx_destroy", "flutl_ctx_sync",
    "flutl_atlas_put", "flutl_atlas_reset", "flutl_atlas_size",
    "flutl_list_create", "flutl_list_destroy",
    "flutl_render", "flutl_render_to_host",
    "flutl_host_alloc", "flutl_host_free", "flutl_render_frame_to_host",
    "flutl_render_frame_rows_to_host", "flutl_render_frame_window_to_host",
    "flutl_atlas_put_blurred",
    "flutl_launch_delta", "flutl_launch_delta_window", "flutl_delta_segment_bytes",
    "flutl_launch_frame_delta",
    "flutl_launch_frame_rows",
    "flutl_launch_frame_over_rows", "flutl_blend_over_host",
    "flutl_launch_frame_ordered", "flutl_row_order",
    "flutl_frame_create", "flutl_frame_destroy", "flutl_frame_host",
    "flutl_frame_device", "flutl_frame_submit", "flutl_frame_wait", "flutl_ctx_delta_bytes",
    "flutl_atlas_put_polygon", "flutl_polygon_entry_bytes", "flutl_ctx_scan_fallbacks",
    "flutl_scan_scratch_bytes", "flutl_scan_zero_bytes", "flutl_ctx_edge_block_acquire",
    "flutl_ctx_edge_block_release", "flutl_frame_mark_written", "flutl_frame_create_quiet", "flutl_ctx_reserve_frames", "flutl_frame_stage", "flutl_frame_submit_staged", "flutl_launch_scan", "flutl_scan_status",
]


class FlutlError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        super().__init__(f"{what}: {code} ({strerror(code)})")


_lib = None


def lib():
    """Load lib/libflutl.so; raises loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} missing: build it with __graft_entry__.build() "
                "(there is no CPU fallback for the HIP path)")
        L = ctypes.CDLL(LIB_PATH)
        vp, u32, sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_size_t
        L.flutl_strerror.restype = ctypes.c_char_p
        L.flutl_strerror.argtypes = [ctypes.c_int]
        L.flutl_list_validate.argtypes = [u32, u32, vp, u32, vp, u32, sz]
        L.flutl_launch_frame.argtypes = [vp, u32, u32, vp, sz, vp, u32, vp, vp]
        L.flutl_launch_batch.argtypes = [vp, u32, u32, vp, sz, vp, u32, vp, vp, vp]
        L.flutl_launch_batch_over.argtypes = [vp, u32, u32, vp, vp, sz, vp, u32, vp, vp, vp]
        L.flutl_launch_blur.argtypes = [vp, vp, vp, u32, u32, sz, vp, ctypes.c_int]
        L.flutl_blur_kernel.argtypes = [ctypes.c_int, ctypes.c_double, vp]
        L.flutl_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
        L.flutl_ctx_destroy.argtypes = [vp]
        L.flutl_ctx_destroy.restype = None
        L.flutl_ctx_sync.argtypes = [vp]
        L.flutl_ctx_reserve_frames.argtypes = [vp, u32, u32, u32]
        L.flutl_atlas_put.argtypes = [vp, vp, sz, ctypes.POINTER(u32)]
        L.flutl_atlas_reset.argtypes = [vp]
        L.flutl_atlas_reset.restype = None
        L.flutl_atlas_size.argtypes = [vp]
        L.flutl_atlas_size.restype = sz
        L.flutl_list_create.argtypes = [vp, u32, u32, vp, u32, vp, u32,
                                        ctypes.POINTER(vp)]
        L.flutl_list_destroy.argtypes = [vp, vp]
        L.flutl_list_destroy.restype = None
        L.flutl_render.argtypes = [vp, vp, vp, sz]
        L.flutl_render_to_host.argtypes = [vp, vp, vp, sz]
        L.flutl_render_frame_to_host.argtypes = [vp, u32, u32, vp, u32, vp, u32, vp, sz]
        L.flutl_render_frame_rows_to_host.argtypes = [vp, u32, u32, vp, u32, vp, u32, vp, sz,
                                                      vp, u32]
        L.flutl_render_frame_window_to_host.argtypes = [vp, u32, u32, vp, u32, vp, u32, vp, sz,
                                                        vp, u32, u32, u32]
        L.flutl_atlas_put_blurred.argtypes = [vp, u32, u32, vp, u32, vp, u32, ctypes.c_int,
                                              ctypes.c_double, u32, u32, vp, ctypes.c_int,
                                              vp, ctypes.POINTER(u32)]
        L.flutl_atlas_put_polygon.argtypes = [vp, vp, u32, vp, ctypes.c_int, ctypes.POINTER(u32)]
        L.flutl_polygon_entry_bytes.argtypes = [u32, u32]
        L.flutl_polygon_entry_bytes.restype = sz
        L.flutl_ctx_scan_fallbacks.argtypes = [vp]
        L.flutl_ctx_scan_fallbacks.restype = u32
        L.flutl_scan_scratch_bytes.argtypes = [u32, u32, u32]
        L.flutl_scan_scratch_bytes.restype = sz
        L.flutl_ctx_edge_block_acquire.argtypes = [vp, sz, ctypes.POINTER(vp)]
        L.flutl_ctx_edge_block_release.argtypes = [vp, vp]
        L.flutl_ctx_edge_block_release.restype = None
        L.flutl_frame_stage.argtypes = [vp, vp, vp, u32, vp, u32]
        L.flutl_frame_submit_staged.argtypes = [vp, vp]
        L.flutl_frame_mark_written.argtypes = [vp]
        L.flutl_frame_mark_written.restype = None
        L.flutl_scan_zero_bytes.argtypes = []
        L.flutl_scan_zero_bytes.restype = sz
        L.flutl_launch_scan.argtypes = [vp, vp, u32, vp, ctypes.c_int, vp, vp, sz, ctypes.c_int]
        L.flutl_scan_status.argtypes = [vp, vp, u32, u32, ctypes.POINTER(u32)]
        L.flutl_launch_rotate_cw.argtypes = [vp, vp, vp, u32, u32, sz, sz, vp]
        L.flutl_launch_resample.argtypes = [vp, vp, vp, u32, u32, sz, sz, u32, u32, vp]
        L.flutl_launch_delta.argtypes = [vp, vp, vp, vp, u32, u32, u32, u32, vp]
        L.flutl_launch_delta_window.argtypes = [vp, vp, vp, vp, u32, u32, u32, u32, u32, u32, vp]
        L.flutl_launch_frame_rows.argtypes = [vp, u32, u32, vp, sz, vp, u32, vp, vp, u32, u32]
        L.flutl_launch_frame_over_rows.argtypes = [vp, u32, u32, vp, vp, sz, vp, u32, vp, vp,
                                                   u32, u32]
        L.flutl_blend_over_host.argtypes = [vp, u32, u32, vp, u32, vp, u32, vp, sz]
        L.flutl_launch_frame_ordered.argtypes = [vp, u32, u32, vp, sz, vp, u32, vp, vp, u32, u32, vp]
        L.flutl_row_order.argtypes = [u32, u32, vp, u32, vp, u32, u32, u32, vp]
        L.flutl_launch_frame_delta.argtypes = [vp, u32, u32, vp, vp, vp, sz, vp, u32, vp, vp,
                                               u32, u32, vp, vp]
        L.flutl_delta_segment_bytes.restype = u32
        L.flutl_row_order.restype = u32
        L.flutl_frame_create.argtypes = [vp, u32, u32, ctypes.POINTER(vp)]
        L.flutl_frame_destroy.argtypes = [vp]
        L.flutl_frame_destroy.restype = None
        L.flutl_frame_host.argtypes = [vp]
        L.flutl_frame_host.restype = vp
        L.flutl_frame_device.argtypes = [vp]
        L.flutl_frame_device.restype = vp
        L.flutl_frame_submit.argtypes = [vp, vp, vp, u32, vp, u32]
        L.flutl_frame_wait.argtypes = [vp]
        L.flutl_ctx_delta_bytes.argtypes = [vp]
        L.flutl_ctx_delta_bytes.restype = ctypes.c_uint64
        L.flutl_host_alloc.argtypes = [sz, ctypes.POINTER(vp)]
        L.flutl_host_free.argtypes = [vp]
        L.flutl_host_free.restype = None
        _lib = L
    return _lib


def strerror(code):
    return lib().flutl_strerror(int(code)).decode()


def _check(rc, what):
    if rc != 0:
        raise FlutlError(rc, what)


def _np_ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def validate(width, height, regions, prims, atlas_bytes):
    regions = np.ascontiguousarray(regions, REGION_DTYPE)
    prims = np.ascontiguousarray(prims, PRIM_DTYPE)
    return lib().flutl_list_validate(width, height, _np_ptr(regions), len(regions),
                                     _np_ptr(prims), len(prims), atlas_bytes)


def blur_kernel(radius, sigma):
    out = np.empty((2 * radius + 1) ** 2, np.int32)
    _check(lib().flutl_blur_kernel(radius, float(sigma), _np_ptr(out)),
           "flutl_blur_kernel")
    return out


# ---- stateless launches on torch-owned device memory ----------------------

def _dev_ptr(t):
    return ctypes.c_void_p(t.data_ptr() if hasattr(t, "data_ptr") else int(t))


def launch_frame(stream, width, height, dst, regions_dev, n_regions, prims_dev,
                 atlas_dev, stride=None):
    """dst / *_dev: torch CUDA tensors (or raw int device pointers)."""
    p = _dev_ptr
    _check(lib().flutl_launch_frame(ctypes.c_void_p(stream), width, height, p(dst),
                                    stride or width * 4, p(regions_dev), n_regions,
                                    p(prims_dev), p(atlas_dev)),
           "flutl_launch_frame")


def launch_frame_rows(stream, width, height, dst, regions_dev, n_regions, prims_dev,
                      atlas_dev, row0, row1, stride=None):
    """only the supertile rows that [row0, row1) touches are written"""
    p = _dev_ptr
    _check(lib().flutl_launch_frame_rows(ctypes.c_void_p(stream), width, height, p(dst),
                                         stride or width * 4, p(regions_dev), n_regions,
                                         p(prims_dev), p(atlas_dev), row0, row1),
           "flutl_launch_frame_rows")


def row_order(width, height, regions, prims, row0=0, row1=None):
    """supertile rows of [row0, row1), heaviest first (uint16 indices from row0 // 16)"""
    regions = np.ascontiguousarray(regions, REGION_DTYPE)
    prims = np.ascontiguousarray(prims, PRIM_DTYPE)
    out = np.zeros((height + 15) // 16, np.uint16)
    n = lib().flutl_row_order(width, height, _np_ptr(regions), len(regions), _np_ptr(prims),
                              len(prims), row0, height if row1 is None else row1, _np_ptr(out))
    return out[:n]


def launch_frame_ordered(stream, width, height, dst, regions_dev, n_regions, prims_dev,
                         atlas_dev, row0, row1, order_dev, stride=None):
    p = _dev_ptr
    _check(lib().flutl_launch_frame_ordered(ctypes.c_void_p(stream), width, height, p(dst),
                                            stride or width * 4, p(regions_dev), n_regions,
                                            p(prims_dev), p(atlas_dev), row0, row1,
                                            p(order_dev)),
           "flutl_launch_frame_ordered")


def launch_frame_delta(stream, width, height, twin, host, park, regions_dev, n_regions,
                       prims_dev, atlas_dev, row0, row1, order_dev, changed_dev=None, stride=None):
    """composite rows [row0, row1), compare with `twin`, write what differs into `twin`
    and `host` (any device-accessible memory); `park`: a scratch frame"""
    p = _dev_ptr
    _check(lib().flutl_launch_frame_delta(ctypes.c_void_p(stream), width, height, p(twin), p(host),
                                          p(park), stride or width * 4, p(regions_dev),
                                          n_regions, p(prims_dev), p(atlas_dev), row0, row1,
                                          p(order_dev),
                                          p(changed_dev) if changed_dev is not None else None),
           "flutl_launch_frame_delta")


def delta_segment_bytes():
    return int(lib().flutl_delta_segment_bytes())


def launch_batch(stream, width, height, dst, frames_dev, n_frames, regions_dev,
                 prims_dev, atlas_dev, stride=None):
    p = _dev_ptr
    _check(lib().flutl_launch_batch(ctypes.c_void_p(stream), width, height, p(dst),
                                    stride or width * 4, p(frames_dev), n_frames,
                                    p(regions_dev), p(prims_dev), p(atlas_dev)),
           "flutl_launch_batch")


def launch_batch_over(stream, width, height, dst, video, frames_dev, n_frames,
                      regions_dev, prims_dev, atlas_dev, stride=None):
    """overlay_i OVER video_i in one launch; video has the geometry of dst."""
    p = _dev_ptr
    _check(lib().flutl_launch_batch_over(ctypes.c_void_p(stream), width, height, p(dst),
                                         p(video), stride or width * 4, p(frames_dev),
                                         n_frames, p(regions_dev), p(prims_dev),
                                         p(atlas_dev)),
           "flutl_launch_batch_over")


def launch_resample(stream, src, dst, w, h, distx, disty, box=None, src_stride=None,
                    dst_stride=None):
    """pixman bilinear resample by constant 7-bit weights: w x h -> (w+1) x (h+1);
    box = 24.8 rectangle whose edge coverage multiplies the samples, or None"""
    p = _dev_ptr
    b = None if box is None else np.ascontiguousarray(box, np.int32)
    _check(lib().flutl_launch_resample(ctypes.c_void_p(stream), p(src), p(dst), w, h,
                                       src_stride or w * 4, dst_stride or (w + 1) * 4,
                                       distx, disty, None if b is None else _np_ptr(b)),
           "flutl_launch_resample")


def launch_blur(stream, src, dst, w, h, kernel_dev, radius, stride=None):
    p = _dev_ptr
    _check(lib().flutl_launch_blur(ctypes.c_void_p(stream), p(src), p(dst), w, h,
                                   stride or w * 4, p(kernel_dev), radius),
           "flutl_launch_blur")


def scan_scratch_bytes(n_edges, width, height):
    return int(lib().flutl_scan_scratch_bytes(n_edges, width, height))


SCAN_CLEAN = 16     # mode flag: the scratch's first scan_zero_bytes() are zero (include/flutl.h)


def scan_zero_bytes():
    return int(lib().flutl_scan_zero_bytes())


def launch_scan(stream, edges_dev, n_edges, box, even_odd, scratch, dst, stride, mode=0):
    """polygon (n x 7 int32 edges in device memory) -> A8 coverage of `box` in dst"""
    p = _dev_ptr
    b = (ctypes.c_int32 * 4)(*[int(v) for v in box])
    _check(lib().flutl_launch_scan(ctypes.c_void_p(stream), p(edges_dev), n_edges, b,
                                   int(even_odd), p(scratch), p(dst), stride, mode),
           "flutl_launch_scan")


def scan_status(stream, scratch, n_edges, height):
    st = ctypes.c_uint32(0)
    _check(lib().flutl_scan_status(ctypes.c_void_p(stream), _dev_ptr(scratch), n_edges, height,
                                   ctypes.byref(st)), "flutl_scan_status")
    return st.value


def launch_delta(stream, fresh, twin, host, width, height, row0, row1, counter=None):
    """segments of rows [row0, row1) where fresh != twin are written to twin and host"""
    p = _dev_ptr
    _check(lib().flutl_launch_delta(ctypes.c_void_p(stream), p(fresh), p(twin), p(host),
                                    width, height, row0, row1,
                                    None if counter is None else p(counter)),
           "flutl_launch_delta")


# ---- context API (library-owned HBM), as the GStreamer element uses it ----

class Context:
    def __init__(self, device=0):
        h = ctypes.c_void_p()
        _check(lib().flutl_ctx_create(device, ctypes.byref(h)), "flutl_ctx_create")
        self._h = h

    def close(self):
        if self._h:
            lib().flutl_ctx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def atlas_put(self, data):
        data = np.ascontiguousarray(data)
        off = ctypes.c_uint32()
        _check(lib().flutl_atlas_put(self._h, _np_ptr(data), data.nbytes,
                                     ctypes.byref(off)), "flutl_atlas_put")
        return off.value

    def atlas_put_polygon(self, edges, box, even_odd=False, via_block=False):
        """scan converts the polygon on the device into a new atlas entry (MASK layout:
        pixel (0,0) of the box at offset + 4, stride width + 4); returns the offset"""
        edges = np.ascontiguousarray(edges, np.int32).reshape(-1, 7)
        b = (ctypes.c_int32 * 4)(*[int(v) for v in box])
        off = ctypes.c_uint32()
        src = _np_ptr(edges)
        if via_block:
            # as the element's stroke workers do: the edges written straight into one of the
            # context's page-locked blocks, which the device reads
            blk = ctypes.c_void_p()
            rc = lib().flutl_ctx_edge_block_acquire(self._h, edges.nbytes, ctypes.byref(blk))
            if rc == 0:
                ctypes.memmove(blk, src, edges.nbytes)
                src = blk
            elif via_block == "must":
                # (acquire never waits for the device; a caller that insists does)
                self.sync()
                _check(lib().flutl_ctx_edge_block_acquire(self._h, edges.nbytes, ctypes.byref(blk)),
                       "flutl_ctx_edge_block_acquire")
                ctypes.memmove(blk, src, edges.nbytes)
                src = blk
        _check(lib().flutl_atlas_put_polygon(self._h, src, len(edges), b,
                                             int(even_odd), ctypes.byref(off)),
               "flutl_atlas_put_polygon")
        return off.value

    def atlas_put_blurred(self, w, h, regions, prims, radius, sigma, distx=0, disty=0,
                          box=None, rotate_cw=False):
        """returns (atlas offset, host copy of the entry: the blurred w x h ARGB32
        image, or -- distx/disty != 0 -- its (h+1) x (w+1) bilinear resample)"""
        regions = np.ascontiguousarray(regions, REGION_DTYPE)
        prims = np.ascontiguousarray(prims, PRIM_DTYPE)
        grow = 1 if (distx or disty or box is not None) else 0
        b = None if box is None else np.ascontiguousarray(box, np.int32)
        out = np.empty((w + grow, h + grow) if rotate_cw else (h + grow, w + grow), np.uint32)
        off = ctypes.c_uint32()
        _check(lib().flutl_atlas_put_blurred(
            self._h, w, h, _np_ptr(regions), len(regions), _np_ptr(prims), len(prims),
            radius, float(sigma), distx, disty, None if b is None else _np_ptr(b),
            1 if rotate_cw else 0, _np_ptr(out), ctypes.byref(off)),
            "flutl_atlas_put_blurred")
        return off.value, out

    def atlas_reset(self):
        lib().flutl_atlas_reset(self._h)

    def atlas_size(self):
        return lib().flutl_atlas_size(self._h)

    def list_create(self, width, height, regions, prims):
        regions = np.ascontiguousarray(regions, REGION_DTYPE)
        prims = np.ascontiguousarray(prims, PRIM_DTYPE)
        h = ctypes.c_void_p()
        _check(lib().flutl_list_create(self._h, width, height, _np_ptr(regions),
                                       len(regions), _np_ptr(prims), len(prims),
                                       ctypes.byref(h)), "flutl_list_create")
        return h

    def list_destroy(self, lst):
        lib().flutl_list_destroy(self._h, lst)

    def render_to_host(self, lst, width, height):
        out = np.empty((height, width), np.uint32)
        _check(lib().flutl_render_to_host(self._h, lst, _np_ptr(out), width * 4),
               "flutl_render_to_host")
        return out

    def render_frame_rows_into(self, width, height, regions, prims, out, bands, cols=None):
        """dirty-rectangle form: only the row bands [(y0, y1), ...] of `out` (a C-contiguous
        (height, width) uint32 array) are written -- and of those only the columns
        cols = (x0, x1), if given"""
        regions = np.ascontiguousarray(regions, REGION_DTYPE)
        prims = np.ascontiguousarray(prims, PRIM_DTYPE)
        b = np.ascontiguousarray(bands, np.uint32).reshape(-1)
        if cols is None:
            _check(lib().flutl_render_frame_rows_to_host(
                self._h, width, height, _np_ptr(regions), len(regions), _np_ptr(prims),
                len(prims), _np_ptr(out), width * 4, _np_ptr(b) if b.size else None,
                b.size // 2), "flutl_render_frame_rows_to_host")
        else:
            _check(lib().flutl_render_frame_window_to_host(
                self._h, width, height, _np_ptr(regions), len(regions), _np_ptr(prims),
                len(prims), _np_ptr(out), width * 4, _np_ptr(b) if b.size else None,
                b.size // 2, cols[0], cols[1]), "flutl_render_frame_window_to_host")

    def render_frame_to_host(self, width, height, regions, prims, pinned=True):
        """the element's per-buffer call; with pinned=True the frame lands in
        page-locked memory from flutl_host_alloc (as the element does)"""
        regions = np.ascontiguousarray(regions, REGION_DTYPE)
        prims = np.ascontiguousarray(prims, PRIM_DTYPE)
        nbytes = width * height * 4
        if pinned:
            ptr = ctypes.c_void_p()
            _check(lib().flutl_host_alloc(nbytes, ctypes.byref(ptr)), "flutl_host_alloc")
            dst = ptr
        else:
            out = np.empty((height, width), np.uint32)
            dst = _np_ptr(out)
        try:
            _check(lib().flutl_render_frame_to_host(
                self._h, width, height, _np_ptr(regions), len(regions), _np_ptr(prims),
                len(prims), dst, width * 4), "flutl_render_frame_to_host")
            if pinned:
                out = np.ctypeslib.as_array(
                    ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint32)),
                    shape=(height, width)).copy()
        finally:
            if pinned:
                lib().flutl_host_free(ptr)
        return out

    def render(self, lst, dst_ptr, stride):
        _check(lib().flutl_render(self._h, lst, ctypes.c_void_p(dst_ptr), stride),
               "flutl_render")

    def sync(self):
        _check(lib().flutl_ctx_sync(self._h), "flutl_ctx_sync")

    def blend_over_host(self, width, height, regions, prims, video):
        """overlay OVER `video` (a C-contiguous (height, width) uint32 array), in place"""
        regions = np.ascontiguousarray(regions, REGION_DTYPE)
        prims = np.ascontiguousarray(prims, PRIM_DTYPE)
        _check(lib().flutl_blend_over_host(self._h, width, height, _np_ptr(regions),
                                           len(regions), _np_ptr(prims), len(prims),
                                           _np_ptr(video), video.strides[0]),
               "flutl_blend_over_host")

    def delta_bytes(self):
        return int(lib().flutl_ctx_delta_bytes(self._h))

    def reserve_frames(self, width, height, count):
        """flutl_ctx_reserve_frames: the memory of `count` frames in one piece"""
        _check(lib().flutl_ctx_reserve_frames(self._h, width, height, count), "flutl_ctx_reserve_frames")

    def frame(self, width, height):
        return Frame(self, width, height)


class Frame:
    """A pooled output frame of the element: page-locked host block + HBM twin,
    filled asynchronously (flutl_frame_submit) by delta read-back."""

    def __init__(self, ctx, width, height):
        h = ctypes.c_void_p()
        _check(lib().flutl_frame_create(ctx._h, width, height, ctypes.byref(h)),
               "flutl_frame_create")
        self._h, self.ctx, self.width, self.height = h, ctx, width, height

    def close(self):
        if self._h:
            lib().flutl_frame_destroy(self._h)
            self._h = None

    def submit(self, regions, prims):
        regions = np.ascontiguousarray(regions, REGION_DTYPE)
        prims = np.ascontiguousarray(prims, PRIM_DTYPE)
        _check(lib().flutl_frame_submit(self.ctx._h, self._h, _np_ptr(regions), len(regions),
                                        _np_ptr(prims), len(prims)), "flutl_frame_submit")

    def stage(self, regions, prims):
        """submit in two steps: stage, then the caller makes the frame's atlas entries
        (atlas_put_polygon ...), then submit_staged"""
        regions = np.ascontiguousarray(regions, REGION_DTYPE)
        prims = np.ascontiguousarray(prims, PRIM_DTYPE)
        _check(lib().flutl_frame_stage(self.ctx._h, self._h, _np_ptr(regions), len(regions),
                                       _np_ptr(prims), len(prims)), "flutl_frame_stage")

    def submit_staged(self):
        _check(lib().flutl_frame_submit_staged(self.ctx._h, self._h), "flutl_frame_submit_staged")

    def mark_written(self):
        lib().flutl_frame_mark_written(self._h)

    def wait(self):
        _check(lib().flutl_frame_wait(self._h), "flutl_frame_wait")

    def host(self):
        """the host block as a (height, width) uint32 array (a view, not a copy)"""
        ptr = lib().flutl_frame_host(self._h)
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint32)),
                                     shape=(self.height, self.width))

    def device_ptr(self):
        return int(lib().flutl_frame_device(self._h))
