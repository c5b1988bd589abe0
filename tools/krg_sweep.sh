#!/bin/bash
# GPU-box tool: supertile height sweep (-DFLUTL_KRG=n strips of 4 rows per wave) for
# single-frame, 8-frame and 64-frame launches of bench.py's workload.
#   tools/krg_sweep.sh "2 3 4" [config]
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/krg
CFG=${2:-cfg2_2160p_8regions_outline}
for k in $1; do
  so=/tmp/libflutl_krg$k.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DFLUTL_KRG=$k $FLUTL_CFLAGS \
    -shared -o $so flu-plugins-oss_amd/csrc/flutl_kernels.hip -x hip flu-plugins-oss_amd/csrc/flutl_api.cpp
  for b in 1 8 64; do
    FLUTL_LIB=$so timeout -k 10 300 python bench.py --no-cpu-baseline --no-e2e --config $CFG --batch $b \
      --steps 300 --warmup 50 > gpurun_out/krg/k${k}_b$b.json 2> gpurun_out/krg/k${k}_b$b.err
    python - <<PY
import json
d = json.load(open("gpurun_out/krg/k${k}_b$b.json"))
r = d["roofline"]
print("KRG $k batch $b: value_steady %.0f fps  kernel_ms mean %.4f median %.4f p10 %.4f  frac %.3f" % (
    d["value_steady"], r["kernel_ms"], r["kernel_ms_median"], r["kernel_ms_p10"], r["frac"]))
PY
  done
done
