import sys, os, json
sys.path.insert(0, '/root/repo/tools')
import e2e_timing as E
ttml = '/root/repo/bench_data/typical_4k.ttml'
E.measure(ttml, 3840, 2160)
for rep in range(2):
    E.measure(ttml, 3840, 2160, log_path='/root/repo/gpurun_out/e2e_detail.log')
print(open('/root/repo/gpurun_out/e2e_detail.log').read())
