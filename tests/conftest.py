import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flu-plugins-oss_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # A context takes to its second compositor lane after 8 frames without company on its
    # device (flutl_frame_stage); under test from the first frame on, so that every frame test
    # and every pipeline of a few buffers runs through the lanes' ordering.  The full-size
    # streams are also run as shipped (tests/test_gpu_scenes.py: FLUTL_LANES_AFTER=8).
    os.environ.setdefault("FLUTL_LANES_AFTER", "0")


@pytest.fixture(scope="session")
def kats():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "kats.npz"))
