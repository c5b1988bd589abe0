#!/bin/bash
# GPU-box tool: per-row phase timing (tools/scan_rows_debug.py bench, clocks up) of round 3's
# scan converter (tools/variants/scan_r3.hip) and of the current one.
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -shared -o /tmp/libflutl_r3.so \
  flu-plugins-oss_amd/csrc/flutl_kernels.hip tools/variants/scan_r3.hip -x hip tools/variants/api_r3.cpp 2>/dev/null || echo "r3 build failed"
echo "== round 3"; FLUTL_LIB=/tmp/libflutl_r3.so FLUTL_R3=1 timeout -k 10 120 python3 tools/scan_rows_debug.py bench 2>&1 | tail -16
echo "== now"; timeout -k 10 120 python3 tools/scan_rows_debug.py bench 2>&1 | tail -16
