#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Builds oracle/kat_gen.c against the real libcairo /
libpixman of this image (/opt/conda), runs it and packs the tables into
tests/golden/kats.npz.  Run in the builder container only; the .npz is
committed so that the GPU box (no dev files needed) can check against it.

    python oracle/make_kats.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def main():
    subprocess.check_call(["make", "-C", HERE, "_build/kat_gen"])
    with tempfile.TemporaryDirectory() as d:
        ver = subprocess.check_output([os.path.join(HERE, "_build/kat_gen"), d])
        ver = ver.decode().strip()
        f = lambda n: os.path.join(d, n)
        out = {"versions": np.array(ver)}
        out["color"] = np.fromfile(f("color.u32"), "<u4").reshape(-1, 2)
        out["fill"] = np.fromfile(f("fill.u32"), "<u4").reshape(-1, 3)
        out["mask"] = np.fromfile(f("mask.u32"), "<u4").reshape(-1, 4)
        out["image"] = np.fromfile(f("image.u32"), "<u4").reshape(-1, 3)
        g = np.fromfile(f("group.bin"), np.dtype(
            [("op", "<f8"), ("g", "<u4"), ("d", "<u4"), ("o", "<u4"), ("pad", "<u4")]))
        out["group_opacity"] = g["op"]
        out["group"] = np.stack([g["g"], g["d"], g["o"]], 1)
        W = H = 12
        r = np.fromfile(f("rect.bin"), np.dtype(
            [("p", "<f8", 6), ("c", "<u4"), ("pad", "<u4"),
             ("d", "<u4", (H, W)), ("o", "<u4", (H, W))]))
        out["rect_params"] = r["p"]
        out["rect_color"] = r["c"]
        out["rect_dst"] = r["d"]
        out["rect_out"] = r["o"]
        W, H = 48, 32
        s = np.fromfile(f("stroke.bin"), np.dtype(
            [("c", "<u4"), ("pad", "<u4"), ("lw", "<f8"), ("m", "u1", (H, W)),
             ("d", "<u4", (H, W)), ("o", "<u4", (H, W))]))
        out["stroke_color"] = s["c"]
        out["stroke_mask"] = s["m"]
        out["stroke_dst"] = s["d"]
        out["stroke_out"] = s["o"]
        # blur: variable-size records
        buf = open(f("blur.bin"), "rb").read()
        W, H = 40, 24
        off, i = 0, 0
        while off < len(buf):
            rad = int(np.frombuffer(buf, "<u4", 1, off)[0])
            sigma = float(np.frombuffer(buf, "<f8", 1, off + 8)[0])
            off += 16
            n = (2 * rad + 1) ** 2
            out[f"blur{i}_radius_sigma"] = np.array([rad, sigma])
            out[f"blur{i}_kernel"] = np.frombuffer(buf, "<i4", n, off).copy()
            off += 4 * n
            out[f"blur{i}_src"] = np.frombuffer(buf, "<u4", W * H, off).reshape(H, W).copy()
            off += 4 * W * H
            out[f"blur{i}_out"] = np.frombuffer(buf, "<u4", W * H, off).reshape(H, W).copy()
            off += 4 * W * H
            i += 1
        out["blur_count"] = np.array(i)
        SW, SH, DW, DH = 10, 7, 24, 16
        b = np.fromfile(f("fracblit.bin"), np.dtype(
            [("t", "<f8", 2), ("o", "<i4", 2), ("src", "<u4", (SH, SW)),
             ("d", "<u4", (DH, DW)), ("out", "<u4", (DH, DW))]))
        out["fracblit_t"] = b["t"]
        out["fracblit_o"] = b["o"]
        out["fracblit_src"] = b["src"]
        out["fracblit_dst"] = b["d"]
        out["fracblit_out"] = b["out"]
    dst = os.path.join(ROOT, "tests", "golden", "kats.npz")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    np.savez_compressed(dst, **out)
    print(ver, "->", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    sys.exit(main())
