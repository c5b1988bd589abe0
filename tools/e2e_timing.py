#!/usr/bin/env python3
"""GPU-box tool: the drop-in element in a real pipeline (filesrc ! ttmlrender !
caps ! fakesink) with FLUTL_TIMING=1: push-to-push frames/s and the element's own
host / device split.   python tools/e2e_timing.py cfg2 3840 2160"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gstenv  # noqa: E402

cfg, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ttml = os.path.join(ROOT, "bench_data", cfg + ".ttml")
extra = {"GST_DEBUG": "ttmlbase:5", "GST_DEBUG_NO_COLOR": "1", "FLUTL_TIMING": "1"}
for k, v in os.environ.items():
    if k.startswith("FLUTL_"):
        extra[k] = v
env = gstenv.env_for(gstenv.HIP_PLUGIN_DIR, extra)
cmd = [os.path.join(gstenv.CONDA, "bin", "gst-launch-1.0"), "-q", "filesrc",
       f"location={ttml}", "!", "ttmlrender", "!", f"video/x-raw,width={W},height={H}",
       "!", "fakesink"]
r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=tempfile.gettempdir())
times = []
for line in r.stderr.split("\n"):
    if "Pushing buffer of" in line:
        h, m, s = line.split()[0].split(":")
        times.append(int(h) * 3600 + int(m) * 60 + float(s))
    if "ttmlrender-hip timing" in line:
        print(line)
if len(times) > 2:
    print(f"{cfg} {W}x{H}: {len(times)} buffers, {(len(times) - 1) / (times[-1] - times[0]):.1f} frames/s push-to-push")
else:
    print("rc", r.returncode, r.stderr[-1500:])
