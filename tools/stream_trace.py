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
    if os.environ.get("SPROF_DOC") == "cfg2":      # (the cfg2 kind of stream, `cues` states long)
        import gen_config_ttml
        open(doc, "w").write(gen_config_ttml.gen("cfg2", 2, n_states=cues))
    else:
        sprof_stream.document(cues, doc)
    out = os.path.join(ROOT, "gpurun_out", "stream_trace")
    os.makedirs(out, exist_ok=True)
    env = gstenv.env_for(gstenv.HIP_PLUGIN_DIR, {"FLUTL_TIMING": "1", "TMPDIR": "/tmp"})
    # (FLUTL_TRACE_HIP=1: the runtime's API calls as well -- what each costs the calling thread)
    hip = ["--hip-runtime-trace"] if os.environ.get("FLUTL_TRACE_HIP") else []
    cmd = ["rocprofv3", "--kernel-trace", "--memory-copy-trace"] + hip + ["--stats", "--output-format", "csv",
           "-d", out, "-o", "t", "--", os.path.join(gstenv.CONDA, "bin", "gst-launch-1.0"), "-q",
           "filesrc", f"location={doc}", "!", "ttmlrender", "!", "video/x-raw,width=3840,height=2160",
           "!", "fakesink"]
    r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
    print([l for l in r.stderr.split("\n") if "timing" in l][-1:][0][:400] if "timing" in r.stderr else r.stderr[-1500:])
    for name in ("kernel_stats", "memory_copy_stats", "hip_api_stats"):
        for f in glob.glob(os.path.join(out, "**", f"*{name}.csv"), recursive=True):
            print("==", os.path.relpath(f, ROOT))
            for row in csv.DictReader(open(f)):
                print("  %-60s calls %6s avg %9.1f ns total %6.2f ms" % (
                    row["Name"][:60], row["Calls"], float(row["AverageNs"]), float(row["TotalDurationNs"]) / 1e6))
    timeline(out)


def timeline(out):
    """the stream's operations in time order over the middle of the run: per kind its mean
    duration and the mean idle gap in front of it, and the period of a cue"""
    ops = []
    for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            kind = ("composite" if "composite" in n else "scan_rows" if "scan_rows" in n else
                    "scan_prepare" if "scan_prepare" in n or "scan_bucket" in n else n[:24])
            ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
    for f in glob.glob(os.path.join(out, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")[:16]))
    ops.sort()
    if len(ops) < 100:
        return
    ops = ops[len(ops) // 4: 3 * len(ops) // 4]
    import collections
    dur, gap, cnt = collections.Counter(), collections.Counter(), collections.Counter()
    for (a, b) in zip(ops, ops[1:]):
        k = b[2]
        dur[k] += b[1] - b[0]
        gap[k] += max(0, b[0] - a[1])
        cnt[k] += 1
    cues = cnt.get("composite", 1)
    print(f"timeline over the middle half of the run ({cues} cues): period {(ops[-1][0] - ops[0][0]) / cues / 1000:.1f} us per cue")
    for k in sorted(cnt, key=lambda k: -dur[k]):
        print("  %-26s x%.2f per cue  mean %7.2f us   idle in front of it %6.2f us" % (
            k, cnt[k] / cues, dur[k] / cnt[k] / 1000, gap[k] / cnt[k] / 1000))
    print("  busy %.1f us + idle %.1f us per cue" % (sum(dur.values()) / cues / 1000, sum(gap.values()) / cues / 1000))
    # two streams: how much of the time two operations run side by side
    edges = sorted([(a, 1) for a, _, _ in ops] + [(b, -1) for _, b, _ in ops])
    depth, last, any_busy, two_busy = 0, edges[0][0], 0, 0
    for t, d in edges:
        if depth >= 1:
            any_busy += t - last
        if depth >= 2:
            two_busy += t - last
        depth += d
        last = t
    print("  the device busy %.1f us per cue, of which two operations side by side %.1f us" % (
        any_busy / cues / 1000, two_busy / cues / 1000))


if __name__ == "__main__":
    main()
