#!/usr/bin/env python3
"""GPU box: per-row timing of the polygon scan converter (development aid)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flu-plugins-oss_amd"))
import pyflutl as F  # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "polygons.npz"))
meta = z["meta"]
which = sys.argv[1] if len(sys.argv) > 1 else "big"
if which == "bench":
    b = np.load(os.path.join(ROOT, "bench_data", "outline_4k.npz"))
    edges, box = b["edges0"], tuple(int(v) for v in b["meta"][0][1:5])
elif which == "big":
    text = [i for i in range(len(meta)) if meta[i][0] == 2]
    parts, x = [], 0
    for i in text:
        e = z[f"edges{i}"].copy()
        x0, x1 = int(meta[i][1]), int(meta[i][3])
        if x + (x1 - x0) > 3800:
            break
        e[:, 0] += (x - x0) * 256
        e[:, 2] += (x - x0) * 256
        parts.append(e)
        x += x1 - x0 + 4
    edges, box = np.concatenate(parts), (0, 0, min(x, 4090), 320)
else:
    i = int(which)
    edges, box = z[f"edges{i}"], tuple(int(v) for v in meta[i][1:5])
w, h = box[2] - box[0], box[3] - box[1]
ed = torch.from_numpy(np.ascontiguousarray(edges, np.int32).view(np.uint8).reshape(-1).copy()).cuda()
scratch = torch.zeros(F.scan_scratch_bytes(len(edges), w, h) + 256, dtype=torch.uint8, device="cuda")
dst = torch.empty(h * (w + 4), dtype=torch.uint8, device="cuda")
dbg = torch.zeros(h * 16, dtype=torch.int32, device="cuda")
L = F.lib()
L.flutl_scan_set_debug.argtypes = [ctypes.c_void_p]
L.flutl_scan_set_debug.restype = None
stream = torch.cuda.current_stream().cuda_stream
# clocks up first: half a second of busy work, then conversions back to back; the timed
# one comes right behind them
import time
x = torch.randn(4096, 4096, device="cuda")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    y = x @ x
for _ in range(200):
    F.launch_scan(stream, ed, len(edges), box, 0, scratch, dst, w + 4, 1)
L.flutl_scan_set_debug(ctypes.c_void_p(dbg.data_ptr()))
F.launch_scan(stream, ed, len(edges), box, 0, scratch, dst, w + 4, 1)
torch.cuda.synchronize()
L.flutl_scan_set_debug(None)
d = dbg.cpu().numpy().reshape(h, 16)
print(f"{len(edges)} edges, box {w}x{h}; rows by time (10 ns ticks):")
order = np.argsort(-d[:, 0])
for r in order[:12]:
    print(f"  row {r}: {d[r,0]/100:.1f} us (setup {d[r,3]/100:.1f}), {d[r,1]} edges, mode {d[r,2]}; "
          f"state {d[r,8]/100:.1f} count {d[r,4]/100:.1f} prefix {d[r,5]/100:.1f} place {d[r,6]/100:.1f} read+clear {d[r,7]/100:.1f}")
for m in sorted(set(d[:, 2])):
    sel = d[:, 2] == m
    print(f"mode {m}: {sel.sum()} rows, mean {d[sel,0].mean()/100:.1f} us, max {d[sel,0].max()/100:.1f} us, mean edges {d[sel,1].mean():.0f}")

