#!/usr/bin/env python3
"""GPU box, under rocprofv3: the launch the ELEMENT makes per buffer (the DELTA instantiation
of the compositor) in bench.py's `cfg2_delta_stream` shape -- cfg2's states into three twin /
page-locked host pairs -- so that its kernel time and counters get a summary of their own
(tools/profile_round4.sh).  Prints bench.py's block for it."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "flu-plugins-oss_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402

print(json.dumps({"cfg2_delta_stream": bench.delta_stream_block()}))
