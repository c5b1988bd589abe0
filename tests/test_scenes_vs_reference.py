"""CPU: scene-level pinning of the oracle -- and of the host-side display-list
builder -- against frames rendered by the REFERENCE plugin itself.

tests/golden/scenes/<name>.npz (tools/make_goldens.py, builder container) holds
for each TTML document next to it: the frames the reference's ttmlrender pushed
(expected), and the display lists + atlas the fluttml-hip element recorded for
the same document.  Here: oracle(display list) == reference frame, bit for bit.
The same fixtures are demanded from the HIP kernel in test_gpu_scenes.py."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as O
import pyflutl as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "scenes", "*.npz")))


def load(path):
    z = np.load(path)
    states = [(z[f"regions{i}"], z[f"prims{i}"]) for i in range(int(z["n_states"]))]
    return int(z["width"]), int(z["height"]), states, z["atlas"], z["frames"]


def test_fixtures_exist():
    assert len(SCENES) >= 7
    for s in SCENES:
        assert os.path.exists(s[:-4] + ".ttml"), "fixture without its TTML input"


@pytest.mark.parametrize("path", SCENES, ids=[os.path.basename(s)[:-4] for s in SCENES])
def test_oracle_reproduces_reference_frames(path):
    W, H, states, atlas, frames = load(path)
    assert len(states) == len(frames)
    for i, (R, P) in enumerate(states):
        assert F.validate(W, H, R, P, atlas.size) == 0
        got = O.render(W, H, R, P, atlas)
        assert np.array_equal(got, frames[i]), \
            f"state {i}: {(got != frames[i]).sum()} px differ from the reference"


def test_fixtures_cover_every_primitive_and_rule():
    ops, rules, grouped = set(), set(), False
    for path in SCENES:
        _, _, states, _, _ = load(path)
        for R, P in states:
            ops |= set(P["op"].tolist())
            rules |= set(zip(P["op"].tolist(), P["rule"].tolist()))
    assert {F.OP_RECT, F.OP_MASK, F.OP_GROUP_BEGIN, F.OP_GROUP_END} <= ops
    assert (F.OP_MASK, F.RULE_LERP) in rules and (F.OP_MASK, F.RULE_PIXMAN) in rules
    assert (F.OP_RECT, F.RULE_LERP) in rules and (F.OP_RECT, F.RULE_PIXMAN) in rules
