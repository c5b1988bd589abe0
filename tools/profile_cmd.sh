#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 over one command -- kernel-trace stats, then the
# PMC passes, each in a run of its own (gfx950 has 4 TCC counter slots per pass: FETCH_SIZE takes
# 3 of them and WRITE_SIZE 2, so they never share a pass -- the request that asked for both plus
# the hit / miss counters in round 3 could not be scheduled and never came back).  The program
# comes directly after `--` (no env / bash -c hop: the profiler's library has initialised the GPU).
#   tools/profile_cmd.sh <tag> python3 <script> [args...]
# Output: gpurun_out/prof_<tag>/{trace,pmc_fetch,pmc_write,pmc_sq,pmc_sq2}/ ; condensed by
# tools/collect_profiles.py into profiles/.
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run () {   # name, then rocprofv3 arguments
  local name=$1; shift
  timeout -k 10 400 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- "${CMD[@]}" > "$OUT/$name.out" 2> "$OUT/$name.err" \
    || { echo "profile $TAG: pass $name failed or timed out (rc $?)"; tail -5 "$OUT/$name.err"; return 1; }
}
CMD=("$@")
run trace --kernel-trace --stats || exit 1
run pmc_fetch --kernel-trace --pmc FETCH_SIZE || exit 1
run pmc_write --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
run pmc_sq --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES || exit 1
run pmc_sq2 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SMEM || exit 1
cp "$OUT/trace.out" "$OUT/bench_trace.json" 2>/dev/null
echo "profile $TAG done"
