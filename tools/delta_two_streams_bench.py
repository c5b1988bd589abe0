#!/usr/bin/env python3
"""GPU box: the element's DELTA launches (cfg2's states into 4 twin + page-locked host pairs)
queued on ONE stream against alternating between TWO (a frame always on the same one): does a
launch's wait for PCIe hide behind the next launch's compositing?
    python tools/delta_two_streams_bench.py [n]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
W, H, states, atlas, _ = bench.load_states("cfg2_2160p_8regions_outline", 64, None)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()
A_dev = dev(atlas)
nf = 4
twins = [torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda") for _ in range(nf)]
hosts = [torch.zeros(W * H * 4, dtype=torch.uint8).pin_memory() for _ in range(nf)]
parks = [torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda") for _ in range(2)]
changed = torch.zeros(1, dtype=torch.int64, device="cuda")
per = []
for (R, P) in states:
    live = [r for r in R if r["y1"] > r["y0"]]
    y0 = min(int(r["y0"]) for r in live)
    y1 = max(int(r["y1"]) for r in live)
    per.append((dev(R), len(R), dev(P), y0, y1, dev(F.row_order(W, H, R, P, y0, y1))))
ss = [torch.cuda.Stream(), torch.cuda.Stream()]


def launch(i, two):
    rd, nr, pd, y0, y1, od = per[i % len(per)]
    f = i % nf
    k = (f & 1) if two else 0
    F.launch_frame_delta(ss[k].cuda_stream, W, H, twins[f], hosts[f], parks[k], rd, nr, pd, A_dev,
                         y0, y1, od, changed)


x = torch.randn(4096, 4096, device="cuda")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    y = x @ x
torch.cuda.synchronize()
for rep in range(2):
    for two in (False, True):
        for i in range(3 * len(per)):
            launch(i, two)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            launch(i, two)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / n * 1e6
        print(f"{'two streams' if two else 'one stream '}: {us:.1f} us per launch", flush=True)
