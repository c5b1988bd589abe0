#!/usr/bin/env python3
"""GPU-box diagnosis: one single-frame launch of bench.py's workload with the
-DFLUTL_EXP=20 build of the compositor, in which every wave leaves its start and
end clock in the first 16 bytes of its supertile.  Prints when waves start, when
they end and how long they run, by how many coverage primitives hit their tile.
    python tools/wave_timeline.py [config] [state]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = "/tmp/libflutl_exp20.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                       "-I" + os.path.join(ROOT, "include"), "-DFLUTL_EXP=20", "-shared", "-o", so,
                       os.path.join(ROOT, "flu-plugins-oss_amd/csrc/flutl_kernels.hip"), "-x", "hip",
                       os.path.join(ROOT, "flu-plugins-oss_amd/csrc/flutl_api.cpp")],
                      stderr=subprocess.DEVNULL)
os.environ["FLUTL_LIB"] = so
sys.path[:0] = [os.path.join(ROOT, "flu-plugins-oss_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import pyflutl as F
import scene_pack

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
pk = scene_pack.load(os.path.join(ROOT, "bench_data", cfg + ".flpk.z"))
W, H = pk.width, pk.height
R, P = pk.states()[int(sys.argv[2]) if len(sys.argv) > 2 else 5]
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()
rd, pd, ad = dev(R), dev(P), dev(pk.atlas)
dst = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
order = dev(F.row_order(W, H, R, P))
for ordered in (False, True):
    for it in range(5):     # warm
        F.launch_frame(st, W, H, dst, rd, len(R), pd, ad)
    torch.cuda.synchronize()
    if ordered:
        F.launch_frame_ordered(st, W, H, dst, rd, len(R), pd, ad, 0, H, order)
    else:
        F.launch_frame(st, W, H, dst, rd, len(R), pd, ad)
    torch.cuda.synchronize()
    fr = dst.cpu().numpy().view(np.uint64).reshape(H, W // 2)
    SX, SY = W // 64, (H + 15) // 16
    t0 = fr[0:SY * 16:16, 0:SX * 32:32].astype(np.int64)
    t1 = fr[0:SY * 16:16, 1:SX * 32:32].astype(np.int64)
    base = t0.min()
    t0, t1 = t0 - base, t1 - base
    nm = np.zeros((SY, SX), int)
    for p in P:
        if p["op"] == 2 and p["cx1"] > p["cx0"]:
            nm[p["cy0"] // 16:(p["cy1"] + 15) // 16, p["cx0"] // 64:(p["cx1"] + 63) // 64] += 1
    print(f"--- {cfg} {W}x{H} ordered={ordered}: clock ticks (s_memtime units) ---")
    print("kernel span (first start -> last end):", int(t1.max()))
    print("starts: p50 %d p90 %d p99 %d max %d" % tuple(np.percentile(t0, [50, 90, 99, 100])))
    print("ends  : p10 %d p50 %d p90 %d p99 %d max %d" % tuple(np.percentile(t1, [10, 50, 90, 99, 100])))
    d = t1 - t0
    for k in range(nm.max() + 1):
        sel = nm == k
        if sel.any():
            print(f"  tiles with {k} mask prims: n={sel.sum():5d} duration mean {d[sel].mean():8.0f} "
                  f"p90 {np.percentile(d[sel], 90):8.0f} max {d[sel].max():8.0f}  end mean {t1[sel].mean():8.0f}")
