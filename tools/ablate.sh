#!/bin/bash
# GPU-box tool: build ablation variants of the compositor (-DFLUTL_EXP=n, see
# csrc/flutl_kernels.hip) and time bench.py's default workload with each.
#   tools/ablate.sh "0 1 2 3" [extra bench args]
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/exp
for n in $1; do
  so=/tmp/libflutl_exp$n.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DFLUTL_EXP=$n $FLUTL_CFLAGS \
    -shared -o $so flu-plugins-oss_amd/csrc/flutl_kernels.hip -x hip flu-plugins-oss_amd/csrc/flutl_api.cpp
  FLUTL_LIB=$so timeout -k 10 300 python bench.py --no-cpu-baseline ${@:2} > gpurun_out/exp/exp$n.json 2> gpurun_out/exp/exp$n.err
  python - <<PY
import json
d = json.load(open("gpurun_out/exp/exp$n.json"))
print("EXP $n: %.1f fps  ms/step %.4f  kernel_ms %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"]))
PY
done
