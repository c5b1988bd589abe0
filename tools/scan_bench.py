#!/usr/bin/env python3
"""GPU box: the polygon scan converter on bench.py's outline (bench_data/outline_4k.npz) with
the GPU's clocks up (half a second of busy work first, then 300 conversions back to back):
us per conversion from HIP events, bit exactness against Cairo's coverage, and -- under
rocprofv3 --kernel-trace --stats -- the kernels' own times at those clocks.
    python tools/scan_bench.py [bench|big|<index into tests/golden/polygons.npz>] [n]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flu-plugins-oss_amd"))
import pyflutl as F  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "bench"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
want = None
if which == "bench":
    b = np.load(os.path.join(ROOT, "bench_data", "outline_4k.npz"))
    edges, box, want = b["edges0"], tuple(int(v) for v in b["meta"][0][1:5]), b["cov0"]
else:
    z = np.load(os.path.join(ROOT, "tests", "golden", "polygons.npz"))
    i = int(which)
    edges, box, want = z[f"edges{i}"], tuple(int(v) for v in z["meta"][i][1:5]), z[f"cov{i}"]
w, h = box[2] - box[0], box[3] - box[1]
ed = torch.from_numpy(np.ascontiguousarray(edges, np.int32).view(np.uint8).reshape(-1).copy()).cuda()
scratch = torch.zeros(F.scan_scratch_bytes(len(edges), w, h) + 256, dtype=torch.uint8, device="cuda")
dst = torch.zeros(h * (w + 4), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
go = lambda: F.launch_scan(stream, ed, len(edges), box, 0, scratch, dst, w + 4, F.SCAN_CLEAN)
go()
torch.cuda.synchronize()
got = dst.cpu().numpy().reshape(h, w + 4)[:, :w]
exact = bool(np.array_equal(got, want))
# clocks up: a busy GPU for half a second
x = torch.randn(4096, 4096, device="cuda")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    y = x @ x
torch.cuda.synchronize()
for _ in range(20):
    go()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(n):
    go()
b.record()
torch.cuda.synchronize()
print(f"{len(edges)} edges, box {w}x{h}: {a.elapsed_time(b) / n * 1000:.1f} us per conversion "
      f"({n} back to back), bit exact vs Cairo: {exact}")
