"""CPU checks of what bench.py feeds the GPU: every committed scene pack loads, validates
against its atlas and has the geometry of its BASELINE config; the profiled-traffic lookup
finds the committed rocprof summary of the default command."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "flu-plugins-oss_amd"))
import pyflutl as F  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(F.LIB_PATH),
                                reason="libflutl.so not built (run __graft_entry__.build())")


def _bench():
    import importlib
    return importlib.import_module("bench")


@pytest.mark.parametrize("cfg,seed", [("cfg1_1080p_4regions", None), ("cfg2_2160p_8regions_outline", None),
                                      ("cfg4_4320p_16regions", None)] +
                         [("cfg2_2160p_8regions_outline", s) for s in range(20, 28)])
def test_scene_packs_are_valid_display_lists(cfg, seed):
    b = _bench()
    W, H, states, atlas, source = b.load_states(cfg, 32, seed)
    assert (W, H) == b.CONFIGS[cfg][:2]
    assert source.endswith(".flpk.z"), source
    if seed is not None:
        assert f"seed{seed}" in source
    assert len(states) >= 16
    assert not atlas[:F.ATLAS_SLACK].any(), "the leading atlas slack must be zero"
    nreg = b.CONFIGS[cfg][2]
    for R, P in states:
        assert F.validate(W, H, R, P, atlas.size) == 0
        assert 1 <= len(R) <= nreg
        assert len(P) > 0


def test_profiled_traffic_of_the_default_command_is_committed():
    b = _bench()
    import inspect
    default_batch = inspect.signature(b.main).parameters  # noqa: F841  (main takes no args)
    src = open(os.path.join(ROOT, "bench.py")).read()
    import re
    batch = int(re.search(r'"--batch", type=int, default=(\d+)', src).group(1))
    t = b.profiled_traffic("cfg2_2160p_8regions_outline", batch)
    assert t is not None and t[0] > batch * 3840 * 2160 * 4 * 0.99
    assert t[1].startswith("profiles/") and os.path.exists(os.path.join(ROOT, t[1]))


def test_every_stream_of_config3_has_reference_hashes():
    """cfg3 = cfg2 with seeds 20..27, one stream per GPU: each pack comes with the
    SHA-256 of the frames the reference rendered (checked on the GPU by
    tests/test_gpu_scenes.py::test_full_size_config_matches_reference_sha256)."""
    import json
    for seed in range(20, 28):
        meta = json.load(open(os.path.join(ROOT, "bench_data", f"cfg2_seed{seed}.sha256.json")))
        assert meta["states"] == 64 and len(meta["sha256"]) == 64
        assert (meta["width"], meta["height"]) == (3840, 2160)


def test_end_to_end_documents_have_reference_hashes_of_every_frame():
    """The documents the element's end-to-end figures are measured on come with the
    SHA-256 of every frame the reference renders for them (tools/make_e2e_sha.py; checked
    on the GPU through the real pipeline by test_element_at_full_size_is_the_reference_s);
    cfg2's agree with the per-state hashes of its scene pack."""
    import json
    for doc, frames in (("cfg2", 241), ("typical_4k", 121)):
        meta = json.load(open(os.path.join(ROOT, "bench_data", doc + ".e2e_sha256.json")))
        assert os.path.exists(os.path.join(ROOT, "bench_data", meta["document"]))
        assert (meta["width"], meta["height"], meta["frames"]) == (3840, 2160, frames)
        assert len(meta["sha256"]) == frames and all(len(s) == 64 for s in meta["sha256"])
    pack = json.load(open(os.path.join(ROOT, "bench_data", "cfg2.sha256.json")))
    e2e = json.load(open(os.path.join(ROOT, "bench_data", "cfg2.e2e_sha256.json")))
    assert e2e["sha256"][:pack["states"]] == pack["sha256"]
