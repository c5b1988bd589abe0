#!/usr/bin/env python3
"""GPU box: what the consumer of ttmlrender's buffers pays (tools/map_cost.c): dropped
unmapped / mapped for reading / made writable and mapped READWRITE as an in-place element does,
with the element's READONLY memory (default) and with FLUTL_WRITABLE=1; 3840x2160, the metric's
kind of stream 2400 scene states long.   python tools/map_cost.py [states]  -> JSON"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_config_ttml  # noqa: E402
import gstenv           # noqa: E402


def build():
    exe = os.path.join(tempfile.gettempdir(), "flutl_map_cost")
    C = gstenv.CONDA
    inc = [f"-I{C}/include/gstreamer-1.0", f"-I{C}/include/glib-2.0", f"-I{C}/lib/glib-2.0/include",
           f"-I{C}/lib/gstreamer-1.0/include"]
    subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tools", "map_cost.c")] + inc +
                          [f"-L{C}/lib", f"-Wl,-rpath,{C}/lib", "-lgstbase-1.0", "-lgstreamer-1.0",
                           "-lgobject-2.0", "-lglib-2.0"])
    return exe


def measure(states=2400, W=3840, H=2160):
    exe = build()
    out = {"what": "frames/s seen by a consumer of ttmlrender's buffers (a GstBaseTransform behind it): passed on unmapped, mapped "
                   "for reading, mapped READWRITE in place; %dx%d, %d states"
                   % (W, H, states)}
    with tempfile.TemporaryDirectory() as d:
        doc = os.path.join(d, "cfg2_long.ttml")
        with open(doc, "w") as f:
            f.write(gen_config_ttml.gen("cfg2", 2, n_states=states))
        for label, extra in (("readonly", {}), ("FLUTL_WRITABLE", {"FLUTL_WRITABLE": "1"})):
            env = gstenv.env_for(gstenv.HIP_PLUGIN_DIR, extra)
            row = {}
            for mode in ("none", "none", "read", "write"):     # (the first run warms the caches)
                print(f"map_cost: {label} {mode}", file=sys.stderr, flush=True)
                try:
                    r = subprocess.run([exe, doc, str(W), str(H), mode], env=env, capture_output=True,
                                       text=True, cwd=d, timeout=90)
                except subprocess.TimeoutExpired as e:
                    row[mode] = {"error": "still running after 90 s: " + str((e.stderr or b"")[-300:])}
                    break       # (one attempt; nothing further on a pipeline that hung)
                m = re.search(r"buffers=(\d+) frames_per_s=([0-9.]+) memories_replaced_by_copies=(\d+)", r.stdout)
                row[mode] = ({"frames_per_s": float(m.group(2)), "buffers": int(m.group(1)),
                              "frames_copied_by_gstreamer": int(m.group(3))} if m and r.returncode == 0
                             else {"error": (r.stderr or r.stdout)[-300:]})
            out[label] = row
    return out


if __name__ == "__main__":
    print(json.dumps(measure(int(sys.argv[1]) if len(sys.argv) > 1 else 2400)))
