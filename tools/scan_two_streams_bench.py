#!/usr/bin/env python3
"""GPU box: the bench outline's scan conversion queued on ONE stream against alternating between
TWO (each with a scratch buffer of its own), alone and with the element's DELTA launches on a
third stream beside them: would a context's conversions overlap each other?
    python tools/scan_two_streams_bench.py [n]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
b = np.load(os.path.join(ROOT, "bench_data", "outline_4k.npz"))
edges, box = b["edges0"], tuple(int(v) for v in b["meta"][0][1:5])
w, h = box[2] - box[0], box[3] - box[1]
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()
ed = dev(np.ascontiguousarray(edges, np.int32))
scratch = [torch.zeros(F.scan_scratch_bytes(len(edges), w, h) + 256, dtype=torch.uint8, device="cuda") for _ in range(2)]
dst = [torch.zeros(h * (w + 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
W, H, states, atlas, _ = bench.load_states("cfg2_2160p_8regions_outline", 64, None)
A_dev = dev(atlas)
nf = 3
twins = [torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda") for _ in range(nf)]
hosts = [torch.zeros(W * H * 4, dtype=torch.uint8).pin_memory() for _ in range(nf)]
park = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
changed = torch.zeros(1, dtype=torch.int64, device="cuda")
per = []
for (R, P) in states:
    live = [r for r in R if r["y1"] > r["y0"]]
    y0 = min(int(r["y0"]) for r in live)
    y1 = max(int(r["y1"]) for r in live)
    per.append((dev(R), len(R), dev(P), y0, y1, dev(F.row_order(W, H, R, P, y0, y1))))
sa = [torch.cuda.Stream(), torch.cuda.Stream()]
sm = torch.cuda.Stream()


def scan(i, two):
    k = (i & 1) if two else 0
    F.launch_scan(sa[k].cuda_stream, ed, len(edges), box, 0, scratch[k], dst[k], w + 4, F.SCAN_CLEAN)


def delta(i):
    rd, nr, pd, y0, y1, od = per[i % len(per)]
    F.launch_frame_delta(sm.cuda_stream, W, H, twins[i % nf], hosts[i % nf], park, rd, nr, pd, A_dev,
                         y0, y1, od, changed)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


x = torch.randn(4096, 4096, device="cuda")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    y = x @ x
torch.cuda.synchronize()
for i in range(40):
    scan(i, True)
    delta(i)
for rep in range(2):
    a1 = timed(lambda: [scan(i, False) for i in range(n)])
    a2 = timed(lambda: [scan(i, True) for i in range(n)])
    b1 = timed(lambda: [(scan(i, False), delta(i)) for i in range(n)])
    b2 = timed(lambda: [(scan(i, True), delta(i)) for i in range(n)])
    print(f"us per conversion: one stream {a1:.1f}, two {a2:.1f}; with a DELTA launch beside each: one {b1:.1f}, two {b2:.1f}", flush=True)
