"""TEST/BENCH INFRASTRUCTURE: times the REFERENCE's own ttmlrender
(oracle/_ref/libgstfluttml.so, built from /root/reference by build_ref.sh and
shipped to the GPU box as a prebuilt file) on the host cores of the machine
bench.py runs on.  Used only by bench.py's cpu_baseline leg.

Method: `filesrc ! ttmlrender ! capsfilter ! fakesink` on the
config's generated TTML (bench_data/<cfg>.ttml); wall time of the whole
pipeline minus the wall time of the same pipeline on a one-interval document
of the same frame size (process start, plugin scan, parse, caps negotiation,
first buffer), divided into the number of additional buffers.  The reference is
single threaded per element instance: cores = 1.
"""
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools"))

CFG_OF = {"cfg0_720p_2regions": "cfg0", "cfg1_1080p_4regions": "cfg1",
          "cfg2_2160p_8regions_outline": "cfg2", "cfg4_4320p_16regions": "cfg4"}

ONE_STATE = """<?xml version="1.0" encoding="UTF-8"?>
<tt xmlns="http://www.w3.org/ns/ttml" xmlns:tts="http://www.w3.org/ns/ttml#styling">
 <head><layout><region xml:id="r" tts:origin="10% 10%" tts:extent="10% 10%"/></layout></head>
 <body><div><p region="r" begin="0s" end="1s" tts:fontFamily="DejaVu Sans">x</p></div></body>
</tt>
"""


def _frames_in(ttml_path):
    """number of buffers the base class pushes = timeline intervals + 1; counted
    cheaply from the committed metadata"""
    import json
    meta = ttml_path[:-5] + ".sha256.json"
    if os.path.exists(meta):
        return int(json.load(open(meta))["reference_frames_total"])
    return None


def _push_times(plugin_dir, ttml, W, H):
    """wall-clock times (s) at which the base class pushed each buffer:
    GST_DEBUG=ttmlbase:5 "Pushing buffer" lines (gstttmlbase.c:183-191)"""
    import subprocess
    import gstenv
    env = gstenv.env_for(plugin_dir, {"GST_DEBUG": "ttmlbase:5", "GST_DEBUG_NO_COLOR": "1"})
    cmd = [os.path.join(gstenv.CONDA, "bin", "gst-launch-1.0"), "-q", "filesrc",
           f"location={os.path.abspath(ttml)}", "!", "ttmlrender", "!",
           f"video/x-raw,width={W},height={H}", "!", "fakesink"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True,
                       cwd=tempfile.gettempdir())
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])
    times = []
    for line in r.stderr.split("\n"):
        if "Pushing buffer of" in line:
            h, m, sec = line.split()[0].split(":")
            times.append(int(h) * 3600 + int(m) * 60 + float(sec))
    return times


def measure_reference(config, W, H, plugin_dir=None):
    """frames/s = (N - 1) / (t_last_push - t_first_push): start-up, plugin scan,
    XML parse and the first buffer are excluded."""
    import gstenv
    plugin_dir = plugin_dir or gstenv.REF_PLUGIN_DIR
    cfg = CFG_OF.get(config, config)
    ttml = os.path.join(ROOT, "bench_data", cfg + ".ttml")
    if not (gstenv.available(plugin_dir) and os.path.exists(ttml)):
        return None
    _push_times(plugin_dir, ttml, W, H)          # warm file/font caches
    t = _push_times(plugin_dir, ttml, W, H)
    if len(t) < 3:
        return None
    fps = (len(t) - 1) / max(t[-1] - t[0], 1e-9)
    who = ("the reference's ttmlrender (oracle/_ref, cairo 1.16.0 / pango 1.45.3)"
           if plugin_dir == gstenv.REF_PLUGIN_DIR else "the fluttml-hip ttmlrender element")
    return {"value": round(fps, 3), "unit": "frames/s", "cores": 1,
            "kind": "reference" if plugin_dir == gstenv.REF_PLUGIN_DIR else "element",
            "sample": f"{len(t)} buffers of bench_data/{cfg}.ttml at {W}x{H} through {who} "
                      f"! fakesink, push-to-push {t[-1] - t[0]:.3f} s (first buffer and "
                      f"start-up excluded)"}


def measure_reference_all_cores(config, W, H, n_procs=None):
    """The reference is single threaded per element: its "all host cores" figure
     is n concurrent pipelines, one per core; frames/s are summed."""
    import concurrent.futures
    import gstenv
    cfg = CFG_OF.get(config, config)
    ttml = os.path.join(ROOT, "bench_data", cfg + ".ttml")
    if not (gstenv.available(gstenv.REF_PLUGIN_DIR) and os.path.exists(ttml)):
        return None
    n = n_procs or max(1, min(len(os.sched_getaffinity(0)), 32))
    with concurrent.futures.ThreadPoolExecutor(n) as ex:
        runs = list(ex.map(lambda _: _push_times(gstenv.REF_PLUGIN_DIR, ttml, W, H), range(n)))
    runs = [t for t in runs if len(t) >= 3]
    if not runs:
        return None
    fps = sum((len(t) - 1) / max(t[-1] - t[0], 1e-9) for t in runs)
    return {"value": round(fps, 3), "unit": "frames/s", "cores": len(runs),
            "kind": "reference",
            "sample": f"{len(runs)} concurrent reference pipelines (one per core) over "
                      f"bench_data/{cfg}.ttml at {W}x{H}, frames/s summed"}


if __name__ == "__main__":
    import gstenv
    which = sys.argv[4] if len(sys.argv) > 4 else "ref"
    print(measure_reference(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]),
                            plugin_dir=gstenv.HIP_PLUGIN_DIR if which == "hip" else None))
