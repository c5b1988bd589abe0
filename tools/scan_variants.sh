#!/bin/bash
# GPU-box tool: compile-time variants of the scan converter, timed by tools/scan_bench.py under
# rocprofv3 (per-kernel averages).   tools/scan_variants.sh "" "-DFLUTL_BUCKET_EDGES=256" ...
cd "$(dirname "$0")/.."
ROOT=$(pwd)
mkdir -p gpurun_out/scanv
i=0
for flags in "$@"; do
  so=/tmp/libflutl_scanv$i.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude $flags \
    -shared -o $so flu-plugins-oss_amd/csrc/flutl_kernels.hip flu-plugins-oss_amd/csrc/flutl_scan.hip \
    -x hip flu-plugins-oss_amd/csrc/flutl_api.cpp 2>/dev/null || { echo "[$flags] build failed"; i=$((i+1)); continue; }
  export FLUTL_LIB=$so
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv \
     -d "$ROOT/gpurun_out/scanv/v$i" -o p -- python3 "$ROOT/tools/scan_bench.py" bench \
     > "$ROOT/gpurun_out/scanv/v$i.out" 2>&1)
  echo "[$flags] $(grep 'us per conversion' gpurun_out/scanv/v$i.out)"
  python3 - "$ROOT/gpurun_out/scanv/v$i/p_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "scan_" in r["Name"]:
        print("     %-22s calls %4s avg %8.1f ns  min %6s max %6s" % (r["Name"].split("::")[1].split("(")[0], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
  i=$((i+1))
done
unset FLUTL_LIB
