#!/bin/bash
# GPU-box tool: alternate two builds of the compositor on bench.py's default workload.
set -e
cd "$(dirname "$0")/.."
for k in 4 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DFLUTL_KRG=$k \
    -shared -o /tmp/libflutl_k$k.so flu-plugins-oss_amd/csrc/flutl_kernels.hip -x hip flu-plugins-oss_amd/csrc/flutl_api.cpp 2>/dev/null
done
for i in 1 2 3 4; do for k in 4 6; do
  FLUTL_LIB=/tmp/libflutl_k$k.so python bench.py --no-cpu-baseline --no-e2e --steps 1500 --warmup 100 ${@} | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); r=d['roofline']
print('KRG $k: value %.0f steady %.0f kernel_ms mean %.4f median %.4f' % (d['value'], d['value_steady'], r['kernel_ms'], r['kernel_ms_median']))"
done; done
