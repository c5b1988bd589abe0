#!/bin/bash
# GPU box: more end-to-end fuzz on round 4's build, in the modes that move the atlas and the
# streams about: the atlas starting over every 2 MB, one stream, the frames' memory of their
# own, writable frames.  $1: seed base (default 60000).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
export FUZZ_E2E=1 FLUTL_CHECK_PATHS=1
S=${1:-60000}
FLUTL_ATLAS_LIMIT_MB=2 timeout -k 10 330 python3 tools/fuzz_ttml.py $S 200 > gpurun_out/fuzz_r4b_atlas.txt 2>&1; tail -1 gpurun_out/fuzz_r4b_atlas.txt
FLUTL_ONE_STREAM=1 FLUTL_NO_FRAME_ARENA=1 timeout -k 10 250 python3 tools/fuzz_ttml.py $((S + 1000)) 150 > gpurun_out/fuzz_r4b_one.txt 2>&1; tail -1 gpurun_out/fuzz_r4b_one.txt
FLUTL_WRITABLE=1 FUZZ_SETS=1 timeout -k 10 250 python3 tools/fuzz_ttml.py $((S + 2000)) 150 > gpurun_out/fuzz_r4b_writable.txt 2>&1; tail -1 gpurun_out/fuzz_r4b_writable.txt
FUZZ_SCALE=2 FLUTL_ATLAS_LIMIT_MB=8 timeout -k 10 250 python3 tools/fuzz_ttml.py $((S + 3000)) 100 > gpurun_out/fuzz_r4b_x2.txt 2>&1; tail -1 gpurun_out/fuzz_r4b_x2.txt
grep -c ": ok" gpurun_out/fuzz_r4b_*.txt
