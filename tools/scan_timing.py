#!/usr/bin/env python3
"""GPU box: time the polygon scan converter (flutl_launch_scan) on the committed text
strokes and on a synthetic two-line 4K outline built from them, HIP events."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flu-plugins-oss_amd"))
import pyflutl as F  # noqa: E402


def main():
    z = np.load(os.path.join(ROOT, "tests", "golden", "polygons.npz"))
    meta = z["meta"]
    text = [i for i in range(len(meta)) if meta[i][0] == 2]
    # a wide synthetic polygon: text strokes side by side (shifted copies), as one outline
    parts, x = [], 0
    for i in text:
        e = z[f"edges{i}"].copy()
        x0, x1 = int(meta[i][1]), int(meta[i][3])
        if x + (x1 - x0) > 3800:
            break
        shift = (x - x0) * 256
        e[:, 0] += shift
        e[:, 2] += shift
        parts.append(e)
        x += x1 - x0 + 4
    big = np.concatenate(parts)
    cases = [("text stroke %d" % i, z[f"edges{i}"], tuple(int(v) for v in meta[i][1:5])) for i in text[:6]]
    cases.append(("synthetic 4K line", big, (0, 0, min(x, 4090), 320)))
    stream = torch.cuda.current_stream().cuda_stream
    for name, edges, box in cases:
        w, h = box[2] - box[0], box[3] - box[1]
        ed = torch.from_numpy(np.ascontiguousarray(edges, np.int32).view(np.uint8).reshape(-1).copy()).cuda()
        scratch = torch.empty(F.scan_scratch_bytes(len(edges), w, h) + 256, dtype=torch.uint8, device="cuda")
        dst = torch.empty(h * (w + 4), dtype=torch.uint8, device="cuda")
        for mode in (1, 0):
            for _ in range(3):
                F.launch_scan(stream, ed, len(edges), box, 0, scratch, dst, w + 4, mode)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            n = 20
            for _ in range(n):
                F.launch_scan(stream, ed, len(edges), box, 0, scratch, dst, w + 4, mode)
            b.record()
            torch.cuda.synchronize()
            st = F.scan_status(stream, scratch, len(edges), h) if mode == 1 else 0
            print(f"{name}: {len(edges)} edges, box {w}x{h}, mode {mode}: "
                  f"{a.elapsed_time(b) / n * 1000:.1f} us per polygon, status {st}")


if __name__ == "__main__":
    main()
