#!/bin/bash
# GPU box: A/B of an environment setting over the element's end-to-end legs: cfg2.ttml (241
# buffers), the same kind 2400 states long, both with four streams in one process, and the
# 120-cue document of new outlined text.   tools/r4_queue_ab.sh "" "FLUTL_QUEUE_ALWAYS=1"
cd "$(dirname "$0")/.."
python3 - "$@" <<'PY'
import os, sys, tempfile
sys.path.insert(0, "tools")
import e2e_timing as E, gen_config_ttml
W, H = 3840, 2160
long_doc = os.path.join(tempfile.gettempdir(), "cfg2_2400_states.ttml")
open(long_doc, "w").write(gen_config_ttml.gen("cfg2", 2, n_states=2400))
variants = sys.argv[1:] or [""]
E.measure("bench_data/cfg2.ttml", W, H)
for rep in range(2):
    for v in variants:
        env = dict(kv.split("=", 1) for kv in v.split()) if v else {}
        a = E.measure("bench_data/cfg2.ttml", W, H, extra_env=env)["frames_per_s"]
        b = E.measure(long_doc, W, H, extra_env=env)["frames_per_s"]
        c = E.measure_streams("bench_data/cfg2.ttml", W, H, 4, extra_env=env)["frames_per_s"]
        d = E.measure_streams(long_doc, W, H, 4, extra_env=env)["frames_per_s"]
        e = E.measure("bench_data/typical_4k.ttml", W, H, extra_env=env)["frames_per_s"]
        print(f"[{v}] cfg2 {a:.0f}  2400 states {b:.0f}  cfg2 x4 {c:.0f}  2400 states x4 {d:.0f}  typical {e:.0f}", flush=True)
PY
