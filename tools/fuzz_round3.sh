#!/bin/bash
# GPU box: end-to-end fuzz batches of round 3's final build (every composed outline path
# checked against Cairo's on the way): reference frames vs the element's, byte for byte.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
export FUZZ_E2E=1 FLUTL_CHECK_PATHS=1
timeout -k 10 400 python3 tools/fuzz_ttml.py 40000 220 > gpurun_out/fuzz_r3_plain.txt 2>&1; tail -1 gpurun_out/fuzz_r3_plain.txt
FUZZ_SETS=1 timeout -k 10 300 python3 tools/fuzz_ttml.py 41000 150 > gpurun_out/fuzz_r3_sets.txt 2>&1; tail -1 gpurun_out/fuzz_r3_sets.txt
FUZZ_SCALE=3 timeout -k 10 300 python3 tools/fuzz_ttml.py 42000 100 > gpurun_out/fuzz_r3_x3.txt 2>&1; tail -1 gpurun_out/fuzz_r3_x3.txt
FUZZ_UNKNOWN=1 timeout -k 10 200 python3 tools/fuzz_ttml.py 43000 80 > gpurun_out/fuzz_r3_unknown.txt 2>&1; tail -1 gpurun_out/fuzz_r3_unknown.txt
FUZZ_VIDEO=1 timeout -k 10 200 python3 tools/fuzz_ttml.py 44000 50 > gpurun_out/fuzz_r3_video.txt 2>&1; tail -1 gpurun_out/fuzz_r3_video.txt
grep -c ": ok" gpurun_out/fuzz_r3_*.txt
