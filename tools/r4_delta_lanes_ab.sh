#!/bin/bash
# GPU box: the delta read-back's segment size, 64 bytes (as built) against 128 and 256: the
# library rebuilt on the box with -DFLUTL_DELTA_LANES=8 / 16, the 2400-state stream alone and
# four of them in one process each time.  (Nothing is copied back but the numbers.)
cd "$(dirname "$0")/.."
run () {
python3 - <<'PY'
import os, sys, tempfile
sys.path.insert(0, "tools")
import e2e_timing as E, gen_config_ttml
W, H = 3840, 2160
doc = os.path.join(tempfile.gettempdir(), "cfg2_2400_states.ttml")
open(doc, "w").write(gen_config_ttml.gen("cfg2", 2, n_states=2400))
E.measure(doc, W, H)
one = [E.measure(doc, W, H) for _ in range(3)]
four = [E.measure_streams(doc, W, H, 4) for _ in range(3)]
print("  one stream %s frames/s (delta bytes per frame %.0f)  four %s" % (
    [round(r["frames_per_s"]) for r in one], one[0].get("delta_bytes", 0),
    [round(r["frames_per_s"]) for r in four]), flush=True)
PY
}
echo "segments of 64 bytes"; run
for lanes in ${LANES:-8 16}; do
  make -C flu-plugins-oss_amd -B all CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../include -DFLUTL_DELTA_LANES=$lanes" > /dev/null 2>&1 || { echo "build failed"; exit 1; }
  echo "segments of $((lanes * 16)) bytes"; run
done
