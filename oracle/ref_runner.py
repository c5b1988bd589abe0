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


def measure_reference(config, W, H, budget_s=40.0):
    import gstenv
    cfg = CFG_OF.get(config, config)
    ttml = os.path.join(ROOT, "bench_data", cfg + ".ttml")
    if not (gstenv.available(gstenv.REF_PLUGIN_DIR) and os.path.exists(ttml)):
        return None
    n = _frames_in(ttml)
    if not n:
        return None
    with tempfile.NamedTemporaryFile("w", suffix=".ttml", delete=False) as f:
        f.write(ONE_STATE)
        one = f.name
    try:
        gstenv.render(gstenv.REF_PLUGIN_DIR, one, W, H, sink="fakesink")  # warm caches
        t0 = time.perf_counter()
        gstenv.render(gstenv.REF_PLUGIN_DIR, one, W, H, sink="fakesink")
        t_start = time.perf_counter() - t0
        t0 = time.perf_counter()
        gstenv.render(gstenv.REF_PLUGIN_DIR, ttml, W, H, sink="fakesink")
        t_all = time.perf_counter() - t0
    finally:
        os.unlink(one)
    n0 = 2   # the one-interval document yields 2 buffers (interval + trailing clear)
    fps = (n - n0) / max(t_all - t_start, 1e-9)
    return {"value": round(fps, 3), "unit": "frames/s", "cores": 1, "kind": "reference",
            "sample": f"{n} buffers of bench_data/{cfg}.ttml at {W}x{H} through the "
                      f"reference's ttmlrender (oracle/_ref, cairo 1.16.0/pango 1.45.3) "
                      f"! fakesink; {t_all:.2f} s wall minus {t_start:.2f} s start-up"}


if __name__ == "__main__":
    print(measure_reference(sys.argv[1], int(sys.argv[2]), int(sys.argv[3])))
