#!/usr/bin/env python3
"""Builder-container tool: the streams of BASELINE config 3 -- cfg2 with seeds
20..27, one independent stream per GPU.  Each TTML is made by
tools/gen_config_ttml.py, rendered by the REFERENCE plugin (SHA-256 of every kept
frame -> bench_data/cfg2_seed<NN>.sha256.json) and recorded by the fluttml-hip
element in record-only mode (-> bench_data/cfg2_seed<NN>.flpk.z), cut to the first
N_STATES scene states; the TTML itself is reproducible from the seed and not kept.

    python tools/make_stream_packs.py [first_seed last_seed]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_goldens      # noqa: E402

N_STATES = 64


def main():
    lo = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    hi = int(sys.argv[2]) if len(sys.argv) > 2 else 27
    for seed in range(lo, hi + 1):
        make_goldens.config("cfg2", seed=seed, keep_states=N_STATES, keep_ttml=False)


if __name__ == "__main__":
    main()
