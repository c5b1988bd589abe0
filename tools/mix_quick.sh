#!/bin/bash
# GPU-box tool: for each ablation build (tools/ablate.sh) the bench time and the
# VALU / SALU / LDS instruction counts per wave (one rocprofv3 --pmc pass).
#   [FLUTL_SRC=other_kernels.hip] [FLUTL_CFLAGS=-D...] tools/mix_quick.sh "0 1 6" [bench args]
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
exps=$1; shift
SRC=${FLUTL_SRC:-$ROOT/flu-plugins-oss_amd/csrc/flutl_kernels.hip}
cd /tmp && export TMPDIR=/tmp
for n in $exps; do
  so=/tmp/libflutl_exp$n.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -DFLUTL_EXP=$n $FLUTL_CFLAGS \
    -shared -o $so $SRC $ROOT/flu-plugins-oss_amd/csrc/flutl_scan.hip -x hip $ROOT/flu-plugins-oss_amd/csrc/flutl_api.cpp
  export FLUTL_LIB=$so
  OUT=$ROOT/gpurun_out/mixq_$n
  rm -rf "$OUT"; mkdir -p "$OUT"
  timeout -k 10 200 python3 "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH \
    --output-format csv -d "$OUT/p1" -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 10 --warmup 2 "$@" > /dev/null 2> "$OUT/p1.err"
  python3 - "$OUT" $n <<'PY'
import csv, glob, sys, collections, json
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p1/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "flutl_composite_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
w = sum(acc["SQ_WAVES"]) / max(1, len(acc["SQ_WAVES"]))
d = json.load(open(sys.argv[1] + "/bench.json"))
s = "EXP %s: %9.0f fps kernel_ms %.4f |" % (sys.argv[2], d["value"], d["roofline"]["kernel_ms"])
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_BRANCH"):
    s += " %s %.1f" % (k[9:], sum(acc[k]) / len(acc[k]) / w)
print(s)
PY
done
