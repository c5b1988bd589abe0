#!/usr/bin/env python3
"""GPU-box tool: many more random display lists than the test suite holds, kernel vs
oracle, bit for bit.   python tools/gpu_fuzz_lists.py [first_seed [count]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "flu-plugins-oss_amd"))
import gpu_util          # noqa: E402
import oracle_lib as O   # noqa: E402
import scenes            # noqa: E402

SIZES = [(320, 180), (200, 72), (97, 41), (512, 96), (130, 50), (257, 33), (640, 64), (64, 160)]
first = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 400
bad = []
for s in range(first, first + count):
    W, H = SIZES[s % len(SIZES)]
    R, P, A = scenes.random_scene(s, W, H, n_regions=2 + s % 11, prims_per_region=4 + (s * 7) % 50)
    exp = O.render(W, H, R, P, A)
    got = gpu_util.hip_render(W, H, R, P, A)
    if not np.array_equal(got, exp):
        bad.append((s, W, H, int((got != exp).sum())))
        print("MISMATCH", bad[-1], flush=True)
    if (s - first) % 100 == 99:
        print("seed", s, "done,", len(bad), "bad", flush=True)
print("failing:", bad)
sys.exit(1 if bad else 0)
