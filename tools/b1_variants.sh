#!/bin/bash
# GPU-box tool: single-frame launch time of compile-time variants of the compositor.
#   tools/b1_variants.sh "-DFLUTL_WAVES=4" "-DFLUTL_WAVES=8" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/b1v
i=0
for flags in "$@"; do
  so=/tmp/libflutl_b1v$i.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude $flags \
    -shared -o $so flu-plugins-oss_amd/csrc/flutl_kernels.hip -x hip flu-plugins-oss_amd/csrc/flutl_api.cpp 2>/dev/null
  FLUTL_LIB=$so timeout -k 10 300 python bench.py --no-cpu-baseline --no-e2e --batch 1 --steps 3000 --warmup 300 \
    > gpurun_out/b1v/v$i.json 2> gpurun_out/b1v/v$i.err
  python - <<PY
import json
d = json.load(open("gpurun_out/b1v/v$i.json"))
r = d["roofline"]
print("[$flags] us/step %.2f  kernel_ms median %.4f p10 %.4f" % (d["ms_per_step"] * 1e3, r["kernel_ms_median"], r["kernel_ms_p10"]))
PY
  i=$((i+1))
done
