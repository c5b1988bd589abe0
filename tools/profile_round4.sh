#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): round 4's profiles -- every configuration bench.py's line
# quotes a kernel time for, each with rocprofv3 kernel-trace stats and the PMC passes
# (tools/profile_cmd.sh), condensed by tools/collect_profiles.py r04 ... into profiles/r04_*.
#   tools/profile_round4.sh [tags...]      (default: all)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
B="python3 $ROOT/bench.py --no-cpu-baseline --no-e2e --no-configs --no-outline"
TAGS=${@:-cfg2_b64 cfg1_b64 cfg4_b64 cfg2_b1 delta_stream scan}
for t in $TAGS; do
  case $t in
    cfg2_b64)     tools/profile_cmd.sh $t $B --steps 40 --warmup 10 ;;
    cfg1_b64)     tools/profile_cmd.sh $t $B --config cfg1_1080p_4regions --steps 40 --warmup 10 ;;
    cfg4_b64)     tools/profile_cmd.sh $t $B --config cfg4_4320p_16regions --steps 20 --warmup 4 --kernel-launches 40 ;;
    cfg2_b1)      tools/profile_cmd.sh $t $B --batch 1 --steps 2000 --warmup 200 ;;
    delta_stream) tools/profile_cmd.sh $t python3 $ROOT/tools/delta_profile.py ;;
    scan)         tools/profile_cmd.sh $t python3 $ROOT/tools/scan_bench.py bench ;;
  esac
done
python3 tools/collect_profiles.py r04 $TAGS
# (what gpurun brings back is gpurun_out/, at most 64 MiB: the summaries go there too, the
# raw per-dispatch traces stay behind)
mkdir -p gpurun_out/profiles_r04 && cp profiles/r04_* gpurun_out/profiles_r04/
rm -rf gpurun_out/prof_*
