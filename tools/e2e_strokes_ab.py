import sys, os, json
sys.path.insert(0, '/root/repo/tools')
import e2e_timing as E
for name in ("typical_4k", "cfg2"):
    ttml = os.path.join('/root/repo/bench_data', name + '.ttml')
    E.measure(ttml, 3840, 2160)
    for env in ({}, {"FLUTL_HOST_STROKES": "1"}):
        for rep in range(3):
            r = E.measure(ttml, 3840, 2160, extra_env=env)
            print(name, "host-strokes" if env else "device-strokes", json.dumps({k: r[k] for k in ("frames_per_s", "record_us_per_buffer", "device_us_per_buffer", "strokes", "stroke_us", "pure_wait_us", "polygons", "polygon_edges", "atlas_upload_us") if k in r}))
