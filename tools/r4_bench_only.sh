#!/bin/bash
# GPU box: bench.py as the driver runs it, and a digest of its line.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 800 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_steps20.json 2> gpurun_out/r04_bench_steps20.err; rc=$?
tail -3 gpurun_out/r04_bench_steps20.err
cp gpurun_out/e2e_bench.log gpurun_out/r04_e2e_bench.log 2>/dev/null
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04_bench_steps20.json") if l.startswith("{")][-1])
print("value", d["value"], "steady", d["value_steady"], "frac", d["roofline"]["frac"], "prewarm", d.get("prewarm_s"))
for k, v in d.get("configs", {}).items():
    print(" ", k, {x: v.get(x) for x in ("frames_per_s", "kernel_ms", "kernel_us", "frac", "error")})
print("outline", d.get("outline_scan"))
for k, v in d.get("e2e", {}).items():
    if isinstance(v, dict):
        print(" e2e", k, {x: v.get(x) for x in ("value", "device_us", "host_record_us", "vs_reference", "vs_reference_all_cores", "error")})
print("e2e errors", d.get("e2e", {}).get("errors"))
print("cpu", d.get("cpu_baseline"))
PY
exit $rc
