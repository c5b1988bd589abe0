#!/usr/bin/env python3
"""GPU-box tool: what one flutl_frame_submit costs the calling thread and the device,
outside GStreamer.  The states of bench_data/cfg2 are submitted in turn into a few
pooled frames (the element's shape: one frame per buffer, delta read-back):
    python tools/submit_bench.py [n_submits] [n_frames]
Host time = wall per call; device time = wall per call once the calls are back to back
(the submit blocks on its list ring when the device is behind).  Under rocprofv3
--kernel-trace --stats the per-kernel times of this shape come out as well."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flu-plugins-oss_amd"))
import numpy as np      # noqa: E402
import pyflutl as F     # noqa: E402
import scene_pack       # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pk = scene_pack.find_pack("cfg2", None)
W, H = pk.width, pk.height
states = pk.states()[:64]
with F.Context(0) as ctx:
    ctx.atlas_put(pk.atlas[F.ATLAS_SLACK:])
    frames = [ctx.frame(W, H) for _ in range(nf)]
    lib = F.lib()
    import ctypes
    args = []
    for R, P in states:
        R = np.ascontiguousarray(R, F.REGION_DTYPE)
        P = np.ascontiguousarray(P, F.PRIM_DTYPE)
        args.append((R, P, R.ctypes.data_as(ctypes.c_void_p), len(R), P.ctypes.data_as(ctypes.c_void_p), len(P)))
    for i in range(200):        # warm: every frame written once, caches hot
        a = args[i % len(args)]
        lib.flutl_frame_submit(ctx._h, frames[i % nf]._h, a[2], a[3], a[4], a[5])
    ctx.sync()
    t0 = time.perf_counter()
    for i in range(n):
        a = args[i % len(args)]
        lib.flutl_frame_submit(ctx._h, frames[i % nf]._h, a[2], a[3], a[4], a[5])
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    print(f"{n} submits into {nf} frame(s): {1e6 * (t1 - t0) / n:.1f} us per call, "
          f"{1e6 * (t2 - t0) / n:.1f} us per frame drained, delta {ctx.delta_bytes() / (n + 200) / 1e6:.2f} MB per frame")
    # paced: the calling thread waits for the device before each call -> pure host cost
    t_host = 0.0
    for i in range(300):
        a = args[i % len(args)]
        ctx.sync()
        t = time.perf_counter()
        lib.flutl_frame_submit(ctx._h, frames[i % nf]._h, a[2], a[3], a[4], a[5])
        t_host += time.perf_counter() - t
    ctx.sync()
    print(f"paced (device idle at each call): {1e6 * t_host / 300:.1f} us per call")
    for f in frames:
        f.close()
