#!/usr/bin/env python3
"""GPU-box tool: the element end to end with libflutl built for other delta segment
sizes (flu-plugins-oss_amd/lib/dl16, dl8: built beforehand with hipcc -DFLUTL_DELTA_LANES=16 / 8 = 256 / 128 bytes (command in tools/b1_variants.sh); the default is 4 = 64).
    python tools/delta_segment_ab.py [cfg2|typical_4k] [runs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import e2e_timing  # noqa: E402
import gstenv      # noqa: E402

doc = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ttml = os.path.join(ROOT, "bench_data", doc + ".ttml")
base = gstenv.env_for(gstenv.HIP_PLUGIN_DIR)["LD_LIBRARY_PATH"]
for variant in ("", "dl16", "dl8"):
    d = os.path.join(ROOT, "flu-plugins-oss_amd", "lib", variant)
    if variant and not os.path.exists(os.path.join(d, "libflutl.so")):
        continue
    extra = {"LD_LIBRARY_PATH": d + ":" + base} if variant else {}
    e2e_timing.measure(ttml, 3840, 2160, extra_env=extra)      # warm
    for _ in range(runs):
        m = e2e_timing.measure(ttml, 3840, 2160, extra_env=extra)
        print(variant or "dl4 (default)", m["frames_per_s"], "frames/s  delta_bytes", m.get("delta_bytes"),
              "record", m["record_us_per_buffer"], "device", m["device_us_per_buffer"], flush=True)
