#!/usr/bin/env python3
"""GPU-box tool: the drop-in element in a real pipeline (filesrc ! ttmlrender !
caps ! fakesink) with FLUTL_TIMING=1.  Frames/s is taken from the element's own
clock: (buffers - 1) / (first gen_buffer return -> last frame complete in host
memory), the same push-to-push span oracle/ref_runner.py takes for the
reference from its "Pushing buffer" debug lines (gstttmlbase.c:183-191) -- that
log is not switched on here, it would cost more than a buffer does.

    python tools/e2e_timing.py cfg2 3840 2160        (bench_data/cfg2.ttml)
Importable: measure(ttml_path, W, H, extra_env=None, log_path=None) -> dict.
"""
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gstenv  # noqa: E402

PIPELINE = "filesrc ! ttmlrender ! video/x-raw,width={W},height={H} ! fakesink"


def measure(ttml, W, H, extra_env=None, log_path=None, sink="fakesink", props=()):
    extra = {"FLUTL_TIMING": "1"}
    for k, v in os.environ.items():
        if k.startswith("FLUTL_") and k != "FLUTL_BENCH_REHEARSAL":
            extra[k] = v
    extra.update(extra_env or {})
    env = gstenv.env_for(gstenv.HIP_PLUGIN_DIR, extra)
    cmd = [os.path.join(gstenv.CONDA, "bin", "gst-launch-1.0"), "-q", "filesrc",
           f"location={os.path.abspath(ttml)}", "!", "ttmlrender", *props, "!",
           f"video/x-raw,width={W},height={H}", "!"] + sink.split()
    t0 = time.perf_counter()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=tempfile.gettempdir(),
                       timeout=300)
    wall = time.perf_counter() - t0
    lines = [l for l in r.stderr.split("\n") if "ttmlrender-hip timing" in l]
    if log_path:
        with open(log_path, "a") as f:
            f.write(f"$ {' '.join(cmd[1:])}   (rc {r.returncode}, wall {wall:.2f} s)\n")
            f.write("\n".join(lines) + "\n")
            if r.returncode != 0:
                f.write(r.stderr[-2000:] + "\n")
    if r.returncode != 0 or not lines:
        raise RuntimeError(f"element pipeline failed (rc {r.returncode}): {r.stderr[-1500:]}")
    # one line per document; a pipeline plays one document
    kv = {k: float(v) for k, v in re.findall(r"(\w+)=([0-9.]+)", lines[-1])}
    n = int(kv["buffers"])
    out = {"buffers": n, "frames_per_s": round((n - 1) / (kv["drained_us"] * 1e-6), 2),
           "record_us_per_buffer": kv["record_us"], "device_us_per_buffer": kv["device_us"],
           "span_s": round(kv["span_us"] * 1e-6, 4), "drained_s": round(kv["drained_us"] * 1e-6, 4),
           "pipeline_wall_s": round(wall, 2)}
    for k in ("delta_bytes", "deferred", "resolve_wait_us", "atlas_upload_us", "build_us",
              "layout_us", "strokes", "stroke_us", "stroke_queued_us", "pure_wait_us", "submit_us",
              "push_block_us", "miss_us", "region_hits", "region_misses", "recoloured",
              "polygons", "polygon_edges", "polygons_sequential", "composed_paths", "composed_checked",
              "t_first_us", "t_end_us", "device"):
        if k in kv:
            out[k] = kv[k]
    return out


def measure_streams(ttml, W, H, n, extra_env=None, log_path=None):
    """n element pipelines in ONE gst-launch process (n unlinked chains: they go to
    PLAYING together), one HIP context each.  Frames/s = all buffers but the n first
    ones / (earliest end of a first gen_buffer -> latest frame complete), from the
    elements' CLOCK_MONOTONIC stamps; also how much of that span each stream was in."""
    extra = {"FLUTL_TIMING": "1"}
    for k, v in os.environ.items():
        if k.startswith("FLUTL_") and k != "FLUTL_BENCH_REHEARSAL":
            extra[k] = v
    extra.update(extra_env or {})
    env = gstenv.env_for(gstenv.HIP_PLUGIN_DIR, extra)
    cmd = [os.path.join(gstenv.CONDA, "bin", "gst-launch-1.0"), "-q"]
    for _ in range(n):
        cmd += ["filesrc", f"location={os.path.abspath(ttml)}", "!", "ttmlrender", "!",
                f"video/x-raw,width={W},height={H}", "!", "fakesink"]
    # (a pipeline that hangs must not hang the bench: 2 minutes is 40 times what it takes)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=tempfile.gettempdir(),
                       timeout=120)
    lines = [l for l in r.stderr.split("\n") if "ttmlrender-hip timing" in l and "buffers=0 " not in l]
    if log_path:
        with open(log_path, "a") as f:
            f.write(f"$ {n} x ({' '.join(cmd[2:9])})   (rc {r.returncode})\n" + "\n".join(lines) + "\n")
    if r.returncode != 0 or len(lines) < n:
        raise RuntimeError(f"{n}-stream pipeline failed (rc {r.returncode}, {len(lines)} reports): "
                           f"{r.stderr[-1500:]}")
    kvs = [{k: float(v) for k, v in re.findall(r"(\w+)=([0-9.]+)", l)} for l in lines[-n:]]
    t0 = min(kv["t_first_us"] for kv in kvs)
    t1 = max(kv["t_end_us"] for kv in kvs)
    frames = sum(int(kv["buffers"]) - 1 for kv in kvs)
    return {"frames_per_s": round(frames / ((t1 - t0) * 1e-6), 1), "streams": n,
            "span_s": round((t1 - t0) * 1e-6, 4),
            "per_stream": [round((kv["buffers"] - 1) / (kv["drained_us"] * 1e-6), 1) for kv in kvs],
            "in_span": [round(kv["drained_us"] / (t1 - t0), 2) for kv in kvs]}


if __name__ == "__main__":
    cfg, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    ttml = cfg if os.path.exists(cfg) else os.path.join(ROOT, "bench_data", cfg + ".ttml")
    measure(ttml, W, H)      # warm the file and font caches
    print(os.path.basename(ttml), f"{W}x{H}", measure(ttml, W, H))
