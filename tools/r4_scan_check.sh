#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the scan converter after a change -- its parity tests, where a
# row's time goes (tools/scan_rows_debug.py), and rocprofv3's per-kernel times on bench.py's outline.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" && mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_scan.py -x -q > gpurun_out/scan_tests.txt 2>&1 || { tail -30 gpurun_out/scan_tests.txt; exit 1; }
tail -3 gpurun_out/scan_tests.txt
timeout -k 10 120 python3 tools/scan_rows_debug.py bench > gpurun_out/scan_rows_debug.txt 2>&1 || { tail -20 gpurun_out/scan_rows_debug.txt; exit 1; }
cat gpurun_out/scan_rows_debug.txt
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv \
   -d "$ROOT/gpurun_out/prof_scan_bench" -o p -- python3 "$ROOT/tools/scan_profile.py" bench \
   > "$ROOT/gpurun_out/prof_scan_bench.out" 2>&1) || { tail -20 gpurun_out/prof_scan_bench.out; exit 1; }
find gpurun_out/prof_scan_bench -name "*kernel_stats.csv" | head -1 | xargs cat | cut -d, -f1-7
