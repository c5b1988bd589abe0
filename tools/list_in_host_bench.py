#!/usr/bin/env python3
"""GPU box: may the DELTA launch read its display list straight from page-locked host memory
instead of from a copy in HBM?  cfg2's states; the regions, primitives and row order of each
either in device memory (as now, after the slot's hipMemcpyAsync), in coherent page-locked
memory (every read a PCIe round trip) or in NON-coherent page-locked memory (cacheable in the
L2: visible at kernel boundaries, which is all a list slot needs).  us per launch, back to back.
    python tools/list_in_host_bench.py [n]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "flu-plugins-oss_amd"))
import pyflutl as F  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
hip = ctypes.CDLL("libamdhip64.so")
W, H, states, atlas, _ = bench.load_states("cfg2_2160p_8regions_outline", 64, None)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()
A_dev = dev(atlas)
nf = 3
twins = [torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda") for _ in range(nf)]
hosts = [torch.zeros(W * H * 4, dtype=torch.uint8).pin_memory() for _ in range(nf)]
park = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
changed = torch.zeros(1, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream


def host_block(arr, flags):
    b = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
    p = ctypes.c_void_p()
    assert hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(max(64, len(b))), ctypes.c_uint(flags)) == 0
    ctypes.memmove(p, b.ctypes.data, len(b))
    d = ctypes.c_void_p()
    assert hip.hipHostGetDevicePointer(ctypes.byref(d), p, 0) == 0
    return d.value


def build(kind):
    per = []
    for (R, P) in states:
        live = [r for r in R if r["y1"] > r["y0"]]
        y0 = min(int(r["y0"]) for r in live)
        y1 = max(int(r["y1"]) for r in live)
        order = F.row_order(W, H, R, P, y0, y1)
        if kind == "device":
            per.append((dev(R), len(R), dev(P), y0, y1, dev(order)))
        else:
            flags = 0x3 | (0x80000000 if kind == "host non-coherent" else 0x40000000)
            per.append((host_block(R, flags), len(R), host_block(P, flags), y0, y1, host_block(order, flags)))
    return per


x = torch.randn(4096, 4096, device="cuda")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    y = x @ x
torch.cuda.synchronize()
ref = None
dev_hosts = [torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda") for _ in range(nf)]
pinned_hosts = hosts
for kind in ("device", "host non-coherent", "host coherent", "device", "device, frames' host blocks in HBM"):
    per = build("device" if kind.startswith("device") else kind)
    hosts = dev_hosts if "HBM" in kind else pinned_hosts
    for t in twins:
        t.zero_()

    def launch(i):
        rd, nr, pd, y0, y1, od = per[i % len(per)]
        F.launch_frame_delta(stream, W, H, twins[i % nf], hosts[i % nf], park, rd, nr, pd, A_dev,
                             y0, y1, od, changed)
    for i in range(3 * len(per)):
        launch(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        launch(i)
    b.record()
    torch.cuda.synchronize()
    got = [t.cpu().numpy().copy() for t in twins]
    if ref is None:
        ref = got
    same = all(np.array_equal(g, r) for g, r in zip(got, ref))
    print(f"list in {kind:18s}: {a.elapsed_time(b) / n * 1000:6.2f} us per launch, frames equal: {same}", flush=True)
