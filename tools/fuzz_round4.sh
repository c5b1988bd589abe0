#!/bin/bash
# GPU box: end-to-end fuzz batches of round 4's build (scan conversions on their own stream,
# frames of one reservation, fills and the line-break arrow on the device): reference frames
# vs the element's, byte for byte.  New seeds.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
export FUZZ_E2E=1 FLUTL_CHECK_PATHS=1
timeout -k 10 400 python3 tools/fuzz_ttml.py 50000 220 > gpurun_out/fuzz_r4_plain.txt 2>&1; tail -1 gpurun_out/fuzz_r4_plain.txt
FUZZ_SETS=1 timeout -k 10 300 python3 tools/fuzz_ttml.py 51000 150 > gpurun_out/fuzz_r4_sets.txt 2>&1; tail -1 gpurun_out/fuzz_r4_sets.txt
FUZZ_SCALE=3 timeout -k 10 300 python3 tools/fuzz_ttml.py 52000 100 > gpurun_out/fuzz_r4_x3.txt 2>&1; tail -1 gpurun_out/fuzz_r4_x3.txt
FUZZ_UNKNOWN=1 timeout -k 10 200 python3 tools/fuzz_ttml.py 53000 80 > gpurun_out/fuzz_r4_unknown.txt 2>&1; tail -1 gpurun_out/fuzz_r4_unknown.txt
FUZZ_VIDEO=1 timeout -k 10 200 python3 tools/fuzz_ttml.py 54000 50 > gpurun_out/fuzz_r4_video.txt 2>&1; tail -1 gpurun_out/fuzz_r4_video.txt
grep -c ": ok" gpurun_out/fuzz_r4_*.txt
