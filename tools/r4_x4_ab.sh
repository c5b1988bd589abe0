#!/bin/bash
# GPU box: four elements in one process on the 2400-state stream under settings given as
# arguments (A/B), e.g. "" "GPU_MAX_HW_QUEUES=8".
cd "$(dirname "$0")/.."
python3 - "$@" <<'PY'
import os, sys, tempfile
sys.path.insert(0, "tools")
import e2e_timing as E, gen_config_ttml
W, H = 3840, 2160
long_doc = os.path.join(tempfile.gettempdir(), "cfg2_2400_states.ttml")
open(long_doc, "w").write(gen_config_ttml.gen("cfg2", 2, n_states=2400))
variants = sys.argv[1:] or [""]
E.measure_streams(long_doc, W, H, 4)
for rep in range(3):
    for v in variants:
        env = dict(kv.split("=", 1) for kv in v.split()) if v else {}
        r = E.measure_streams(long_doc, W, H, 4, extra_env=env)
        print(f"[{v}] {r['frames_per_s']:.0f} frames/s  in_span {r.get('in_span')}", flush=True)
PY
