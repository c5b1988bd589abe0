#!/bin/bash
# GPU box: the cfg2 kind of stream 2400 states long, with the element's and the library's
# timing lines: where a buffer's time goes when nothing new has to be rasterised.
cd "$(dirname "$0")/.."
python3 - "$@" <<'PY'
import sys, os, subprocess, tempfile, re
sys.path.insert(0, "tools")
import gstenv, gen_config_ttml
doc = os.path.join(tempfile.gettempdir(), "cfg2_2400_states.ttml")
open(doc, "w").write(gen_config_ttml.gen("cfg2", 2, n_states=2400))
extra = {"FLUTL_API_TIMING": "1", "FLUTL_TIMING": "1"}
for kv in sys.argv[1:]:
    k, v = kv.split("=", 1)
    extra[k] = v
env = gstenv.env_for(gstenv.HIP_PLUGIN_DIR, extra)
cmd = [os.path.join(gstenv.CONDA, "bin", "gst-launch-1.0"), "-q", "filesrc", "location=" + doc, "!", "ttmlrender", "!", "video/x-raw,width=3840,height=2160", "!", "fakesink"]
subprocess.run(cmd, env=env, capture_output=True, text=True)
for rep in range(3):
    r = subprocess.run(cmd, env=env, capture_output=True, text=True)
    for l in r.stderr.split("\n"):
        if "api timing" in l:
            print(l)
        if "ttmlrender-hip timing" in l:
            kv = dict(re.findall(r"(\w+)=([0-9.]+)", l))
            print({k: kv[k] for k in ("buffers", "record_us", "device_us", "span_us", "resolve_wait_us", "atlas_upload_us", "submit_us", "push_block_us", "pure_wait_us", "build_us", "layout_us", "miss_us", "region_hits") if k in kv})
            print("period per buffer %.1f us" % (float(kv["span_us"]) / float(kv["buffers"])))
PY
