"""Helpers for the -m gpu tests: run a display list through the HIP library
(C-ABI, stateless launch on torch-owned HBM) and bring the frame back."""
import numpy as np
import torch

import pyflutl as F


def to_dev(arr):
    a = np.ascontiguousarray(arr)
    if a.size == 0:
        return torch.zeros(64, dtype=torch.uint8, device="cuda")
    return torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).cuda()


def hip_render(width, height, regions, prims, atlas, stride=None):
    rc = F.validate(width, height, regions, prims, atlas.size)
    assert rc == 0, f"invalid display list: {rc}"
    stride = stride or width * 4
    rd, pd, ad = to_dev(regions), to_dev(prims), to_dev(atlas)
    dst = torch.full((height * stride,), 0xAB, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    F.launch_frame(stream, width, height, dst, rd, len(regions), pd, ad, stride=stride)
    torch.cuda.synchronize()
    out = dst.cpu().numpy().reshape(height, stride)
    frame = out[:, :width * 4].copy().view(np.uint32).reshape(height, width)
    pad = out[:, width * 4:]
    assert (pad == 0xAB).all(), "kernel wrote into the row padding"
    return frame
