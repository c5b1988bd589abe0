#!/usr/bin/env python3
"""Builder-container tool: render a TTML document with the REFERENCE plugin
(oracle/_ref) and record it with the fluttml-hip element (record-only mode),
composite the recorded display lists with the CPU oracle and compare.

    python tools/compare_ref.py doc.ttml 320 180
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tools", "tests", "flu-plugins-oss_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import gstenv          # noqa: E402
import oracle_lib as O  # noqa: E402
import pyflutl as F     # noqa: E402
import scene_pack       # noqa: E402


def compare(ttml, W, H, verbose=True):
    ref = np.frombuffer(gstenv.render(gstenv.REF_PLUGIN_DIR, ttml, W, H),
                        np.uint32).reshape(-1, H, W)
    with tempfile.TemporaryDirectory() as d:
        path = gstenv.dump_pack(ttml, W, H, os.path.join(d, "p.flpk"))
        if not os.path.exists(path) and len(ref) == 0:
            return True, ref, None      # a document neither element makes a buffer of
        pk = scene_pack.load(path)
    states = pk.states()
    ok = len(states) == len(ref)
    if verbose:
        print(f"{os.path.basename(ttml)}: {len(ref)} reference frames, "
              f"{len(states)} recorded states")
    total_bad = 0
    for i, (R, P) in enumerate(states[:len(ref)]):
        rc = F.validate(W, H, R, P, pk.atlas.size)
        got = O.render(W, H, R, P, pk.atlas)
        bad = got != ref[i]
        total_bad += int(bad.sum())
        if verbose:
            print(f"  state {i}: {len(R)} regions {len(P)} prims validate={rc} "
                  f"mismatching px={int(bad.sum())}")
            if bad.any():
                ys, xs = np.nonzero(bad)
                print("    bbox", xs.min(), ys.min(), xs.max(), ys.max(), "first:",
                      [(int(x), int(y), hex(got[y, x]), hex(ref[i][y, x]))
                       for y, x in list(zip(ys, xs))[:5]])
    return ok and total_bad == 0, ref, pk


if __name__ == "__main__":
    ok, _, _ = compare(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
    print("BIT-EXACT" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
