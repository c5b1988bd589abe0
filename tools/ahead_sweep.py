#!/usr/bin/env python3
"""GPU box: how far ahead of the device the element should record (FLUTL_MAX_AHEAD) on a
stream with new outlined text in every cue: the short one of bench.py and a long one."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import e2e_timing as E  # noqa: E402
import sprof_stream  # noqa: E402

short = os.path.join(ROOT, "bench_data", "typical_4k.ttml")
long_ = os.path.join(tempfile.gettempdir(), "typical_long.ttml")
sprof_stream.document(3000, long_)
E.measure(short, 3840, 2160)
for ahead in sys.argv[1:] or ["3", "4", "6", "8", "12", "16", "default"]:
    for name, doc in (("short", short), ("long", long_)):
        rs = [E.measure(doc, 3840, 2160, extra_env={} if ahead == "default" else {"FLUTL_MAX_AHEAD": ahead}) for _ in range(3)]
        print(f"ahead {ahead:>7} {name:5}:", " ".join(f"{r['frames_per_s']:8.0f}" for r in rs),
              " push_block", " ".join(f"{r.get('push_block_us', 0):5.1f}" for r in rs),
              " pure_wait", " ".join(f"{r.get('pure_wait_us', 0):5.1f}" for r in rs), flush=True)
