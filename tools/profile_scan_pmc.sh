#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): PMC passes over the scan converter on bench.py's outline
# (tools/scan_profile.py bench), written under gpurun_out/prof_scan_pmc/ and condensed into
# gpurun_out/r03_scan_pmc.txt (per-launch averages for scan_rows_kernel).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_scan_pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU \
  --output-format csv -d "$OUT/a" -- python3 "$ROOT/tools/scan_profile.py" bench > "$OUT/a.out" 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --output-format csv -d "$OUT/b" -- python3 "$ROOT/tools/scan_profile.py" bench > "$OUT/b.out" 2>&1
python3 - "$OUT" > "$ROOT/gpurun_out/r03_scan_pmc.txt" <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "scan_rows_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("scan_rows_kernel, per launch (bench.py's 11 781-edge outline, 239 workgroups of 1024 threads):")
for k in sorted(acc):
    v = acc[k]
    print(f"  {k:24s} {sum(v) / len(v):16.1f}   ({len(v)} launches)")
PY
cat "$ROOT/gpurun_out/r03_scan_pmc.txt"
