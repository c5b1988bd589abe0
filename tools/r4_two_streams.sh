#!/bin/bash
# GPU box: the GPU tests of the library and the element, then the typical stream end to end
# with the scan conversions on their own stream (default) against FLUTL_ONE_STREAM=1, and the
# frames' memory in one reservation against FLUTL_NO_FRAME_ARENA=1.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.txt 2>&1; rc=$?
tail -5 gpurun_out/gpu_tests.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 bash tools/r4_ab_e2e.sh "" "FLUTL_ONE_STREAM=1" "FLUTL_NO_FRAME_ARENA=1" "FLUTL_ONE_STREAM=1 FLUTL_NO_FRAME_ARENA=1"
