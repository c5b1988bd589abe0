#!/usr/bin/env python3
"""GPU box: flutl_frame_submit of cfg2's states into 3 frames in rotation -- frames with
page-locked blocks of their own against frames of one reservation (flutl_ctx_reserve_frames):
does where the host block lies change the DELTA launch's time?
    python tools/arena_bench.py [n]"""
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
W, H, states, atlas, _ = bench.load_states("cfg2_2160p_8regions_outline", 64, None)
x = torch.randn(4096, 4096, device="cuda")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    y = x @ x
torch.cuda.synchronize()
for rep in range(2):
    for reserve in (0, 18, 3):
        with F.Context(0) as ctx:
            ctx.atlas_put(atlas[F.ATLAS_SLACK:])
            t0 = time.perf_counter()
            if reserve:
                ctx.reserve_frames(W, H, reserve)
            frames = [ctx.frame(W, H) for _ in range(3)]
            for f in frames:
                f.wait()
            made = time.perf_counter() - t0
            for i in range(60):
                frames[i % 3].submit(*states[i % len(states)])
            ctx.sync()
            t0 = time.perf_counter()
            for i in range(n):
                frames[i % 3].submit(*states[i % len(states)])
            ctx.sync()
            us = (time.perf_counter() - t0) / n * 1e6
            print(f"reservation of {reserve}: 3 frames made in {made * 1e3:.1f} ms; {us:.1f} us per submit", flush=True)
            for f in frames:
                f.close()
