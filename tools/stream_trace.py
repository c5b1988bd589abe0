#!/usr/bin/env python3
"""GPU box: rocprofv3 kernel + memory-copy trace of the element on a long stream with new
outlined text in every cue: what the device does per buffer.
    python tools/stream_trace.py [cues]   ->  gpurun_out/stream_trace/"""
import csv
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gstenv  # noqa: E402
import sprof_stream  # noqa: E402


def main():
    cues = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    doc = os.path.join(tempfile.gettempdir(), "typical_long.ttml")
    sprof_stream.document(cues, doc)
    out = os.path.join(ROOT, "gpurun_out", "stream_trace")
    os.makedirs(out, exist_ok=True)
    env = gstenv.env_for(gstenv.HIP_PLUGIN_DIR, {"FLUTL_TIMING": "1", "TMPDIR": "/tmp"})
    cmd = ["rocprofv3", "--kernel-trace", "--memory-copy-trace", "--stats", "--output-format", "csv",
           "-d", out, "-o", "t", "--", os.path.join(gstenv.CONDA, "bin", "gst-launch-1.0"), "-q",
           "filesrc", f"location={doc}", "!", "ttmlrender", "!", "video/x-raw,width=3840,height=2160",
           "!", "fakesink"]
    r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
    print([l for l in r.stderr.split("\n") if "timing" in l][-1:][0][:400] if "timing" in r.stderr else r.stderr[-1500:])
    for name in ("kernel_stats", "memory_copy_stats"):
        for f in glob.glob(os.path.join(out, "**", f"*{name}.csv"), recursive=True):
            print("==", os.path.relpath(f, ROOT))
            for row in csv.DictReader(open(f)):
                print("  %-60s calls %6s avg %9.1f ns total %6.2f ms" % (
                    row["Name"][:60], row["Calls"], float(row["AverageNs"]), float(row["TotalDurationNs"]) / 1e6))


if __name__ == "__main__":
    main()
