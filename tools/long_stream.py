#!/usr/bin/env python3
"""GPU-box tool: the element on a cfg2-like stream of 2400 scene states (4801 buffers),
alone and four at once in one process.  python tools/long_stream.py [n_states]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import e2e_timing        # noqa: E402
import gen_config_ttml   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "long.ttml")
    with open(p, "w") as f:
        f.write(gen_config_ttml.gen("cfg2", 2, n_states=n))
    for _ in range(2):
        r = e2e_timing.measure(p, 3840, 2160)
        print({k: r.get(k) for k in ("buffers", "frames_per_s", "record_us_per_buffer",
                                     "device_us_per_buffer", "delta_bytes", "miss_us", "region_hits",
                                     "region_misses", "recoloured", "build_us", "layout_us")}, flush=True)
    for _ in range(2):
        print(e2e_timing.measure_streams(p, 3840, 2160, 4), flush=True)
