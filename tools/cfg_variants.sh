#!/bin/bash
# GPU-box tool: 64-frames-per-launch time of compile-time variants of the compositor on a config.
#   tools/cfg_variants.sh cfg4_4320p_16regions "-DFLUTL_WAVES=8" "-DFLUTL_LUT_MIN=1" ...
set -e
cd "$(dirname "$0")/.."
CFG=$1; shift
mkdir -p gpurun_out/cfgv
i=0
for flags in "$@"; do
  so=/tmp/libflutl_cfgv$i.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude $flags \
    -shared -o $so flu-plugins-oss_amd/csrc/flutl_kernels.hip -x hip flu-plugins-oss_amd/csrc/flutl_api.cpp 2>/dev/null
  FLUTL_LIB=$so timeout -k 10 300 python bench.py --no-cpu-baseline --no-e2e --config $CFG --steps 150 --warmup 20 \
    > gpurun_out/cfgv/v$i.json 2> gpurun_out/cfgv/v$i.err
  python - <<PY
import json
d = json.load(open("gpurun_out/cfgv/v$i.json"))
r = d["roofline"]
print("[$flags] steady %.0f fps  kernel_ms median %.4f p10 %.4f  frac %.3f" % (d["value_steady"], r["kernel_ms_median"], r["kernel_ms_p10"], r["frac"]))
PY
  i=$((i+1))
done
