#!/usr/bin/env python3
"""Builder-container tool: scene packs for BASELINE config 3 -- cfg2 with
seeds 20..27, one independent stream per GPU.  Each TTML is made by
tools/gen_config_ttml.py, recorded by the fluttml-hip element in record-only mode
(no GPU) and cut to the first N_STATES scene states.

    python tools/make_stream_packs.py [first_seed last_seed]
"""
import os
import sys
import tempfile
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tools", "tests", "flu-plugins-oss_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import gen_config_ttml   # noqa: E402
import gstenv            # noqa: E402
import pyflutl as F      # noqa: E402
import scene_pack        # noqa: E402

N_STATES = 64


def main():
    lo = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    hi = int(sys.argv[2]) if len(sys.argv) > 2 else 27
    W, H = gen_config_ttml.SIZES["cfg2"]
    for seed in range(lo, hi + 1):
        with tempfile.TemporaryDirectory() as d:
            ttml = os.path.join(d, f"cfg2_seed{seed}.ttml")
            open(ttml, "w").write(gen_config_ttml.gen("cfg2", seed))
            pk = scene_pack.load(gstenv.dump_pack(ttml, W, H, os.path.join(d, "p.flpk"),
                                                  record_only=True))
        states = pk.states(N_STATES)
        for (R, P) in states:
            assert F.validate(W, H, R, P, pk.atlas.size) == 0
        buf = zlib.compress(scene_pack.dumps(W, H, states, pk.atlas), 9)
        out = os.path.join(ROOT, "bench_data", f"cfg2_seed{seed}.flpk.z")
        open(out, "wb").write(buf)
        print(f"seed {seed}: {len(states)} states, {sum(len(P) for _, P in states)} prims, "
              f"atlas {pk.atlas.size} B -> {len(buf)} B")


if __name__ == "__main__":
    main()
