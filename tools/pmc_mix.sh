#!/bin/bash
# GPU-box tool: dynamic instruction mix of the compositor for one ablation build
# (see tools/ablate.sh).   tools/pmc_mix.sh <exp-number> [bench args]
set -eo pipefail
n=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/mix_exp$n
mkdir -p "$OUT"
so=/tmp/libflutl_exp$n.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -DFLUTL_EXP=$n $FLUTL_CFLAGS \
  -shared -o $so $ROOT/flu-plugins-oss_amd/csrc/flutl_kernels.hip -x hip $ROOT/flu-plugins-oss_amd/csrc/flutl_api.cpp
export FLUTL_LIB=$so
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -- \
    python3 "$ROOT/bench.py" --no-cpu-baseline --steps 30 --warmup 5 "$@" > /dev/null 2> "$OUT/p$i.err"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "flutl_composite_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
w = sum(acc["SQ_WAVES"]) / max(1, len(acc["SQ_WAVES"]))
for k in sorted(acc):
    a = sum(acc[k]) / len(acc[k])
    print("%-24s %14.0f   per wave %10.1f" % (k, a, a / w))
PY
