"""CPU: the C-ABI library loads, exports every symbol include/flutl.h
declares, and its host-side logic (validation, error behaviour without a GPU)
works.  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

import pyflutl as F
import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "flutl.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(flutl_[a-z_]+)\s*\(", hdr)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(F.EXPORTS)


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(F.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(L, name), name
    assert L.flutl_abi_version() == 2


def test_struct_layout_matches_header():
    assert F.PRIM_DTYPE.itemsize == 32
    assert F.PRIM_DTYPE.fields["cx0"][1] == 4      # kernel reads bbox at +4
    assert F.PRIM_DTYPE.fields["color"][1] == 12
    assert F.PRIM_DTYPE.fields["a"][1] == 16
    assert F.REGION_DTYPE.itemsize == 16 and F.REGION_DTYPE.fields["first"][1] == 8


def test_validate_accepts_generated_scenes():
    for seed in range(8):
        R, P, A = scenes.random_scene(seed, 200, 90)
        assert F.validate(200, 90, R, P, A.size) == 0


def test_validate_rejects_malformed_lists():
    R, P, A = scenes.random_scene(3, 200, 90)
    assert F.validate(0, 90, R, P, A.size) == -22
    assert F.validate(40000, 90, R, P, A.size) == -22
    # prim outside its region
    bad = P.copy(); bad["cx1"][0] = 5000
    assert F.validate(200, 90, R, bad, A.size) == -22
    # unknown op
    bad = P.copy(); bad["op"][1] = 9
    assert F.validate(200, 90, R, bad, A.size) == -22
    # region pointing past the prim array
    badr = R.copy(); badr["count"][-1] += 5
    assert F.validate(200, 90, badr, P, A.size) == -22
    # mask read outside the atlas
    i = int(np.nonzero(P["op"] == F.OP_MASK)[0][0])
    bad = P.copy(); bad["a"][i] = A.size
    assert F.validate(200, 90, R, bad, A.size) == -34
    bad = P.copy(); bad["a"][i] = 0          # inside the leading slack
    bad["c"][i] = bad["cx0"][i]; bad["d"][i] = bad["cy0"][i]
    assert F.validate(200, 90, R, bad, A.size) == -34
    # unbalanced group
    g = np.nonzero(P["op"] == F.OP_GROUP_END)[0]
    if len(g):
        bad = P.copy(); bad["op"][g[0]] = F.OP_GROUP_BEGIN
        assert F.validate(200, 90, R, bad, A.size) == -22


def test_strerror():
    assert F.strerror(0) == "ok"
    assert "HIP" in F.strerror(-19)


def test_no_gpu_means_loud_failure():
    """Product path never falls back to a CPU renderer."""
    L = F.lib()
    if L.flutl_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(F.FlutlError) as e:
        F.Context(0)
    assert e.value.code == -19


def test_product_does_not_link_the_oracle():
    """The product sources never reference oracle/."""
    pkg = os.path.join(ROOT, "flu-plugins-oss_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".hip", ".cpp", ".c", ".h", ".py", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "flutl_oracle" not in txt and "oracle_lib" not in txt, f
