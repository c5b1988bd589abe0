#!/bin/bash
# GPU box: end-to-end fuzz on the round's last build with the second compositor lane in use
# from the first frame (FLUTL_LANES_AFTER=0; shipped: after 8 frames) and every frame's list
# read from its page-locked slot.  $1: seed base (default 80000).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
export FUZZ_E2E=1 FLUTL_CHECK_PATHS=1 FLUTL_LANES_AFTER=0
S=${1:-80000}
timeout -k 10 400 python3 tools/fuzz_ttml.py $S 220 > gpurun_out/fuzz_r4c_plain.txt 2>&1; tail -1 gpurun_out/fuzz_r4c_plain.txt
FUZZ_SETS=1 timeout -k 10 300 python3 tools/fuzz_ttml.py $((S + 1000)) 180 > gpurun_out/fuzz_r4c_sets.txt 2>&1; tail -1 gpurun_out/fuzz_r4c_sets.txt
FUZZ_SETS=1 FUZZ_SCALE=2 timeout -k 10 300 python3 tools/fuzz_ttml.py $((S + 2000)) 100 > gpurun_out/fuzz_r4c_x2.txt 2>&1; tail -1 gpurun_out/fuzz_r4c_x2.txt
FUZZ_UNKNOWN=1 timeout -k 10 150 python3 tools/fuzz_ttml.py $((S + 3000)) 60 > gpurun_out/fuzz_r4c_unknown.txt 2>&1; tail -1 gpurun_out/fuzz_r4c_unknown.txt
grep -c ": ok" gpurun_out/fuzz_r4c_*.txt
