#!/usr/bin/env python3
"""TEST INFRASTRUCTURE, builder container only: builds oracle/polygon_kats.c against the
real libcairo under /opt/conda, runs it and packs its records into
tests/golden/polygons.npz (edges + box per polygon, Cairo's coverage as the answer).

    python oracle/make_polygons.py          tests/golden/polygons.npz
    python oracle/make_polygons.py bench    bench_data/outline_4k.npz: the outline of one cue
                                            of a typical 4K stream (bench.py's outline_scan)
"""
import os
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CONDA = os.environ.get("CONDA", "/opt/conda")


def main():
    os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
    exe = os.path.join(HERE, "_build", "polygon_kats")
    subprocess.check_call([
        "gcc", "-O2", "-o", exe, os.path.join(HERE, "polygon_kats.c"),
        os.path.join(HERE, "flutl_tor_oracle.c"),
        os.path.join(ROOT, "flu-plugins-oss_amd", "host", "flutl_stroker.c"),
        f"-I{CONDA}/include/cairo", f"-L{CONDA}/lib", f"-Wl,-rpath,{CONDA}/lib", "-lcairo", "-lm"])
    bench = len(sys.argv) > 1 and sys.argv[1] == "bench"
    raw = os.path.join(HERE, "_build", "polygons.bin")
    subprocess.check_call([exe, raw] + (["0", "0", "0", "1", "bench"] if bench else sys.argv[1:]))
    data = open(raw, "rb").read()
    pos, out, meta = 0, {}, []
    while pos < len(data):
        kind, n, x0, y0, x1, y1, eo = struct.unpack_from("<7i", data, pos)
        pos += 28
        edges = np.frombuffer(data, "<i4", n * 7, pos).reshape(n, 7).copy()
        pos += n * 28
        w, h = x1 - x0, y1 - y0
        cov = np.frombuffer(data, np.uint8, w * h, pos).reshape(h, w).copy()
        pos += w * h
        i = len(meta)
        out[f"edges{i}"] = edges
        out[f"cov{i}"] = cov
        meta.append((kind, x0, y0, x1, y1, eo))
    out["meta"] = np.array(meta, dtype=np.int32)
    out["versions"] = np.array(["cairo 1.16.0 (conda), DejaVu 2.37"])
    path = os.path.join(ROOT, "bench_data", "outline_4k.npz") if bench else \
        os.path.join(ROOT, "tests", "golden", "polygons.npz")
    np.savez_compressed(path, **out)
    print(f"{len(meta)} polygons -> {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


if __name__ == "__main__":
    main()
