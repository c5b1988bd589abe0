"""TEST/BENCH INFRASTRUCTURE: the `cpu_baseline` leg of bench.py.

kind "reference": the reference's own ttmlrender (oracle/_ref/libgstfluttml.so,
built from /root/reference by oracle/build_ref.sh; it travels to the GPU box as
a prebuilt .so) timed through gst-launch on the box's host cores -- only when
the GStreamer runtime of this image (/opt/conda) and the config's TTML document
(bench_data/<config>.ttml) are present.
kind "port": the scalar C restatement (oracle/flutl_oracle.c) over the same
display lists, 1 thread, on a bounded sample of the same states.
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def measure_port(W, H, states, atlas, budget_s=12.0):
    import oracle_lib as O
    n, t0 = 0, time.perf_counter()
    while True:
        R, P = states[n % len(states)]
        O.render(W, H, R, P, atlas)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 4 * len(states):
            break
    return {"value": round(n / el, 3), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n} frames of the same {W}x{H} display lists through "
                      f"oracle/flutl_oracle.c (compositing only, no layout)"}


def measure(config, W, H, states, atlas):
    try:
        import ref_runner
        r = ref_runner.measure_reference(config, W, H)
        if r is not None:
            r["port"] = measure_port(W, H, states, atlas, budget_s=6.0)
            try:    # one instance is the baseline; all cores is reported beside it
                r["all_cores"] = ref_runner.measure_reference_all_cores(config, W, H)
            except Exception as e:      # noqa: BLE001
                print(f"cpu_baseline: all-cores leg failed ({e!r})", file=sys.stderr)
            return r
    except Exception as e:  # the reference leg is optional; the port is not
        print(f"cpu_baseline: reference leg unavailable ({e!r}); using the port",
              file=sys.stderr)
    return measure_port(W, H, states, atlas)
