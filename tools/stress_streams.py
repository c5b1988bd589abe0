#!/usr/bin/env python3
"""GPU box: n element pipelines in one gst-launch process, over and over, each run under
a timeout -- looking for a hang.  python tools/stress_streams.py [runs] [streams] [doc]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gstenv  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
doc = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
ttml = os.path.join(ROOT, "bench_data", doc + ".ttml")
extra = {"FLUTL_TIMING": "1"}
for k, v in os.environ.items():
    if k.startswith("FLUTL_"):
        extra[k] = v
env = gstenv.env_for(gstenv.HIP_PLUGIN_DIR, extra)
cmd = [os.path.join(gstenv.CONDA, "bin", "gst-launch-1.0"), "-q"]
for _ in range(n):
    cmd += ["filesrc", f"location={ttml}", "!", "ttmlrender", "!",
            "video/x-raw,width=3840,height=2160", "!", "fakesink"]
hung = 0
for i in range(runs):
    t0 = time.time()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=tempfile.gettempdir(),
                           timeout=25)
        print(f"run {i}: rc {r.returncode}, {time.time() - t0:.2f} s", flush=True)
        if r.returncode:
            print(r.stderr[-600:], flush=True)
    except subprocess.TimeoutExpired:
        hung += 1
        print(f"run {i}: HUNG", flush=True)
print(f"{hung} of {runs} hung")
