#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): round 3's records under gpurun_out/ --
#   r03_bench_steps20.json   the line the driver's run prints (python bench.py --steps 20 --warmup 5)
#   r03_e2e_bench.log        the element pipelines' own timing lines of that run
#   r03_e2e_strokes_ab.txt   the element end to end, outlines on the device vs by Cairo on the workers
#   prof_cfg2_b64/           tools/profile_gpu.sh cfg2_b64 (bench.py default shape: kernel stats + PMC)
#   prof_scan_bench/         rocprofv3 --kernel-trace --stats of the scan converter on bench.py's outline
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" && mkdir -p gpurun_out
echo "bench"; python3 bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_steps20.json 2> gpurun_out/r03_bench_steps20.err
cp gpurun_out/e2e_bench.log gpurun_out/r03_e2e_bench.log 2>/dev/null
echo "a/b"; python3 tools/e2e_strokes_ab.py > gpurun_out/r03_e2e_strokes_ab.txt 2>&1
echo "scan profile"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv \
   -d "$ROOT/gpurun_out/prof_scan_bench" -o p -- python3 "$ROOT/tools/scan_profile.py" bench \
   > "$ROOT/gpurun_out/prof_scan_bench.out" 2>&1)
echo "compositor profile"; tools/profile_gpu.sh cfg2_b64
echo "round 3 records done"
