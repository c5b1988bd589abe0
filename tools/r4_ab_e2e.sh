#!/bin/bash
# GPU box: the typical stream (3000 cues, new outlined text in every cue) end to end, A/B by
# environment: tools/r4_ab_e2e.sh "" "FLUTL_SCAN_SEPARATE_PREPARE=1" ...   (two runs each, alternating)
cd "$(dirname "$0")/.."
python3 - "$@" <<'PY'
import os, sys, tempfile
sys.path.insert(0, "tools")
import e2e_timing, sprof_stream
doc = os.path.join(tempfile.gettempdir(), "typical_3000.ttml")
sprof_stream.document(3000, doc)
variants = sys.argv[1:] or [""]
e2e_timing.measure(doc, 3840, 2160)
for rep in range(2):
    for v in variants:
        env = dict(kv.split("=", 1) for kv in v.split()) if v else {}
        r = e2e_timing.measure(doc, 3840, 2160, extra_env=env)
        print(f"[{v}] {r['frames_per_s']:.0f} frames/s  record {r['record_us_per_buffer']:.1f}  finisher {r['device_us_per_buffer']:.1f} us", flush=True)
PY
