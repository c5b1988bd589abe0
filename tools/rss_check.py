#!/usr/bin/env python3
"""GPU box: peak resident memory of the element's process on streams of different lengths
(new outlined text in every cue): what grows with the stream leaks.   python tools/rss_check.py"""
import os
import resource
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gstenv  # noqa: E402
import sprof_stream  # noqa: E402

for cues in (1000, 4000, 8000):
    doc = os.path.join(tempfile.gettempdir(), f"typical_{cues}.ttml")
    sprof_stream.document(cues, doc)
    env = gstenv.env_for(gstenv.HIP_PLUGIN_DIR, {"FLUTL_TIMING": "1"})
    cmd = [os.path.join(gstenv.CONDA, "bin", "gst-launch-1.0"), "-q", "filesrc", f"location={doc}", "!",
           "ttmlrender", "!", "video/x-raw,width=3840,height=2160", "!", "fakesink"]
    before = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    after = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    print(f"{cues:6d} cues: rc {r.returncode}, peak RSS of the children so far {after / 1024:.0f} MB (before {before / 1024:.0f})", flush=True)
