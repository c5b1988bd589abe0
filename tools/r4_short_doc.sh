#!/bin/bash
# GPU box: the 120-cue document end to end under a few run-ahead depths (twice each).
cd "$(dirname "$0")/.."
python3 - "$@" <<'PY'
import sys, os
sys.path.insert(0, "tools")
import e2e_timing
doc = os.path.join("bench_data", "typical_4k.ttml")
e2e_timing.measure(doc, 3840, 2160)
for v in sys.argv[1:] or [""]:
    env = dict(kv.split("=", 1) for kv in v.split()) if v else {}
    rs = [e2e_timing.measure(doc, 3840, 2160, extra_env=env) for _ in range(3)]
    print(f"[{v}]", [round(r["frames_per_s"]) for r in rs], "record", rs[-1]["record_us_per_buffer"], "push_block", rs[-1].get("push_block_us"), "resolve_wait", rs[-1].get("resolve_wait_us"), "flush", rs[-1].get("atlas_upload_us"), "submit", rs[-1].get("submit_us"), flush=True)
PY
