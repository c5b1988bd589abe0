#!/usr/bin/env python3
"""GPU box, under rocprofv3 --kernel-trace --stats: the polygon scan converter on one
polygon (argv[1]: 'bench' = bench_data/outline_4k.npz, 'big' = the synthetic 4K line of
tools/scan_timing.py, or an index into
tests/golden/polygons.npz), 50 conversions."""
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
    # the outline bench.py's outline_scan block converts (a cue of a typical 4K stream)
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
scratch = torch.empty(F.scan_scratch_bytes(len(edges), w, h) + 256, dtype=torch.uint8, device="cuda")
dst = torch.empty(h * (w + 4), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
scratch[:F.scan_zero_bytes()].zero_()
for _ in range(50):
    F.launch_scan(stream, ed, len(edges), box, 0, scratch, dst, w + 4, F.SCAN_CLEAN)
torch.cuda.synchronize()
print(f"{len(edges)} edges, box {w}x{h}")
