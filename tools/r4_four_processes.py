#!/usr/bin/env python3
"""GPU box: four element pipelines on ONE GPU as four PROCESSES (one element each) against four
elements in one process (tools/r4_x4_ab.sh): is what caps the four-in-one the device or the
process (the runtime's locks, the allocator)?  The 2400-state stream; frames/s over the span
from the earliest first frame to the latest last one (the elements' CLOCK_MONOTONIC stamps).
    python tools/r4_four_processes.py [n_processes]"""
import os
import sys
import tempfile
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import e2e_timing as E  # noqa: E402
import gen_config_ttml  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W, H = 3840, 2160
doc = os.path.join(tempfile.gettempdir(), "cfg2_2400_states.ttml")
open(doc, "w").write(gen_config_ttml.gen("cfg2", 2, n_states=2400))
E.measure(doc, W, H)
for rep in range(3):
    out = [None] * n

    def run(i):
        out[i] = E.measure(doc, W, H)
    th = [threading.Thread(target=run, args=(i,)) for i in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    t0 = min(r["t_first_us"] for r in out)
    t1 = max(r["t_end_us"] for r in out)
    frames = sum(r["buffers"] - 1 for r in out)
    print(f"{n} processes: {frames / ((t1 - t0) * 1e-6):.0f} frames/s together; each "
          f"{[round(r['frames_per_s']) for r in out]}", flush=True)
