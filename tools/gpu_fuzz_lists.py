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
# the batch launch (frame-per-XCD / interleaved-row mappings) and the fused over-video
# epilogue, 13 frames at a time
import torch                 # noqa: E402
import pyflutl as F          # noqa: E402
import test_gpu_parity as T  # noqa: E402
for s0 in range(first, first + count, 13 * 4):
    W, H = SIZES[(s0 // 13) % len(SIZES)]
    lists = [scenes.random_scene(50000 + s0 + j, W, H, n_regions=2 + j % 9,
                                 prims_per_region=4 + (j * 5) % 40) for j in range(13)]
    R, P, atlas_all, Fd = T._concat_lists(lists, W, H)
    rng = np.random.default_rng(s0)
    video = rng.integers(0, 1 << 32, (13, H, W), dtype=np.uint64).astype(np.uint32)
    video[::2] |= 0xFF000000
    stream = torch.cuda.current_stream().cuda_stream
    args = (gpu_util.to_dev(Fd), 13, gpu_util.to_dev(R), gpu_util.to_dev(P),
            gpu_util.to_dev(atlas_all))
    d1 = torch.full((13 * W * H * 4,), 0xC3, dtype=torch.uint8, device="cuda")
    d2 = torch.full((13 * W * H * 4,), 0x5A, dtype=torch.uint8, device="cuda")
    F.launch_batch(stream, W, H, d1, *args)
    F.launch_batch_over(stream, W, H, d2, gpu_util.to_dev(video), *args)
    torch.cuda.synchronize()
    o1 = d1.cpu().numpy().view(np.uint32).reshape(13, H, W)
    o2 = d2.cpu().numpy().view(np.uint32).reshape(13, H, W)
    for i, (Ri, Pi, Ai) in enumerate(lists):
        exp = O.render(W, H, Ri, Pi, Ai)
        if not np.array_equal(o1[i], exp):
            bad.append(("batch", s0, i, W, H))
            print("MISMATCH", bad[-1], flush=True)
        expv = O.over(exp.reshape(-1), video[i].reshape(-1)).reshape(H, W)
        if not np.array_equal(o2[i], expv):
            bad.append(("over-video", s0, i, W, H))
            print("MISMATCH", bad[-1], flush=True)
print("failing:", bad)
sys.exit(1 if bad else 0)
