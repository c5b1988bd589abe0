#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats and the two
# HBM-traffic PMC passes for bench.py, written under gpurun_out/prof_<tag>/.
# Summaries are then copied into profiles/ by tools/collect_profiles.py.
#   tools/profile_gpu.sh <tag> [bench args...]
set -eo pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
  python3 "$ROOT/bench.py" --no-cpu-baseline --no-e2e "$@" > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- \
  python3 "$ROOT/bench.py" --no-cpu-baseline --no-e2e --steps 40 --warmup 10 "$@" > /dev/null 2> "$OUT/pmc_write.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- \
  python3 "$ROOT/bench.py" --no-cpu-baseline --no-e2e --steps 40 --warmup 10 "$@" > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_sq" -- \
  python3 "$ROOT/bench.py" --no-cpu-baseline --no-e2e --steps 40 --warmup 10 "$@" > /dev/null 2> "$OUT/pmc_sq.err"
echo "profile $TAG done"
