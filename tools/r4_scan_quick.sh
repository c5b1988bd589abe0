#!/bin/bash
# GPU box: scan parity tests, then per-row phases with clocks up, then kernel times
# (with and without the profiler).
cd "$(dirname "$0")/.."
timeout -k 10 600 python3 -m pytest tests/test_gpu_scan.py -x -q 2>&1 | tail -3
timeout -k 10 120 python3 tools/scan_rows_debug.py bench 2>&1 | tail -6
timeout -k 10 120 python3 tools/scan_bench.py bench 2>&1 | tail -1
tools/scan_variants.sh "$@"
