#!/bin/bash
# GPU box: the cfg2 kind of stream 2400 states long under settings given as arguments (A/B).
cd "$(dirname "$0")/.."
python3 - "$@" <<'PY'
import os, sys, tempfile
sys.path.insert(0, "tools")
import e2e_timing as E, gen_config_ttml
W, H = 3840, 2160
long_doc = os.path.join(tempfile.gettempdir(), "cfg2_2400_states.ttml")
open(long_doc, "w").write(gen_config_ttml.gen("cfg2", 2, n_states=2400))
variants = sys.argv[1:] or [""]
E.measure(long_doc, W, H)
for rep in range(2):
    for v in variants:
        env = dict(kv.split("=", 1) for kv in v.split()) if v else {}
        r = E.measure(long_doc, W, H, extra_env=env)
        print(f"[{v}] {r['frames_per_s']:.0f} frames/s  record {r['record_us_per_buffer']:.1f}  device {r['device_us_per_buffer']:.1f}  delta bytes {r.get('delta_bytes', 0):.0f}", flush=True)
PY
