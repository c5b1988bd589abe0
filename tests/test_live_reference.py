"""Where the reference plugin (oracle/_ref, built from /root/reference) and the
GStreamer runtime of this image are present: re-render the fixtures LIVE and
check that the committed goldens still are what the reference produces here,
and that the fluttml-hip element still records the committed display lists.
Skipped elsewhere."""
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gstenv  # noqa: E402
import scene_pack  # noqa: E402

SCENES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "scenes", "*.ttml")))
ids = [os.path.basename(s)[:-5] for s in SCENES]


@pytest.mark.skipif(not gstenv.available(gstenv.REF_PLUGIN_DIR),
                    reason="reference plugin / GStreamer runtime not present")
@pytest.mark.parametrize("ttml", SCENES, ids=ids)
def test_reference_still_produces_the_goldens(ttml):
    z = np.load(ttml[:-5] + ".npz")
    W, H = int(z["width"]), int(z["height"])
    live = np.frombuffer(gstenv.render(gstenv.REF_PLUGIN_DIR, ttml, W, H),
                         np.uint32).reshape(-1, H, W)
    assert np.array_equal(live, z["frames"])


@pytest.mark.skipif(not gstenv.available(gstenv.HIP_PLUGIN_DIR),
                    reason="fluttml-hip plugin / GStreamer runtime not present")
@pytest.mark.parametrize("ttml", [s for s in SCENES if os.path.basename(s)[0] == "k"],
                         ids=[i for i in ids if i[0] == "k"])
def test_element_still_records_the_committed_lists(ttml, tmp_path):
    # (g* scenes hold device-produced atlas entries -- blurred outlines -- and
    # can only be recorded with a GPU: covered by test_gpu_scenes.py)
    z = np.load(ttml[:-5] + ".npz")
    W, H = int(z["width"]), int(z["height"])
    pk = scene_pack.load(gstenv.dump_pack(ttml, W, H, str(tmp_path / "p.flpk")))
    st = pk.states()
    assert len(st) == int(z["n_states"])
    for i, (R, P) in enumerate(st):
        assert np.array_equal(R, z[f"regions{i}"]) and np.array_equal(P, z[f"prims{i}"])
    assert np.array_equal(pk.atlas, z["atlas"])


@pytest.mark.skipif(not gstenv.available(gstenv.HIP_PLUGIN_DIR),
                    reason="fluttml-hip plugin / GStreamer runtime not present")
def test_embedded_images_record_like_file_images(tmp_path):
    """smpte:image data embedded in the document (tests/golden/variants/): the
    reference loses it to a buffer its base class keeps writing to
    (gstttmlstate.c:440-443 vs gstttmlbase.c:428-435; every decode fails here), so
    its frames cannot pin this branch.  The element keeps a copy of the data
    (host/flutl_state_fix.c) and must record exactly what it records for the same
    PNGs referenced as files (k08, which the reference does render) -- and exit
    cleanly (gstenv.render raises on the reference's teardown abort)."""
    d = os.path.join(ROOT, "tests", "golden")
    z = np.load(os.path.join(d, "scenes", "k08_images.npz"))
    W, H = int(z["width"]), int(z["height"])
    pk = scene_pack.load(gstenv.dump_pack(
        os.path.join(d, "variants", "k08_images_embedded.ttml"), W, H, str(tmp_path / "p.flpk")))
    st = pk.states()
    assert len(st) == int(z["n_states"])
    for i, (R, P) in enumerate(st):
        assert np.array_equal(R, z[f"regions{i}"]) and np.array_equal(P, z[f"prims{i}"])
    assert np.array_equal(pk.atlas, z["atlas"])


OUTLINED = [s for s in SCENES if os.path.basename(s)[:3] in
            ("k04", "k07", "k10", "k12", "k13", "k14", "k15", "k18", "k19")]


@pytest.mark.skipif(not gstenv.available(gstenv.HIP_PLUGIN_DIR),
                    reason="fluttml-hip plugin / GStreamer runtime not present")
@pytest.mark.parametrize("ttml", OUTLINED, ids=[os.path.basename(s)[:-5] for s in OUTLINED])
def test_outlines_stroked_on_worker_threads_are_the_reference_s(ttml, tmp_path):
    """The element rasterises outline masks on worker threads (deferred strokes:
    placeholder primitives resolved later).  Recorded that way (FLUTL_DUMP_DEFER=1)
    the lists must still composite -- through the oracle -- to the REFERENCE's
    frames, i.e. a stroke made on a worker's surface from the copied path is the
    stroke Cairo makes in place."""
    import oracle_lib as O
    z = np.load(ttml[:-5] + ".npz")
    W, H = int(z["width"]), int(z["height"])
    pk = scene_pack.load(gstenv.dump_pack(ttml, W, H, str(tmp_path / "p.flpk"),
                                          extra_env={"FLUTL_DUMP_DEFER": "1"}))
    st = pk.states()
    assert len(st) == int(z["n_states"])
    for i, (R, P) in enumerate(st):
        import pyflutl as F
        assert F.validate(W, H, R, P, pk.atlas.size) == 0
        assert np.array_equal(O.render(W, H, R, P, pk.atlas), z["frames"][i]), i


RECOLOURED = [s for s in SCENES if os.path.basename(s)[:3] in ("k06", "k17", "k18")]


@pytest.mark.skipif(not gstenv.available(gstenv.HIP_PLUGIN_DIR),
                    reason="fluttml-hip plugin / GStreamer runtime not present")
@pytest.mark.parametrize("ttml", RECOLOURED, ids=[os.path.basename(s)[:-5] for s in RECOLOURED])
def test_regions_played_back_in_other_colours_are_the_reference_s(ttml, tmp_path):
    """A region whose layouts differ from a cached one in colours only is played back
    from that one's primitives with other colour words (region_recolour).  Recorded
    with and without that (FLUTL_NO_RECOLOUR=1), both lists must composite -- through
    the oracle -- to the REFERENCE's frames; k18 is made of the cases that may and
    may not be exchanged, and must really exchange some."""
    import re
    import subprocess
    import oracle_lib as O
    import pyflutl as F
    z = np.load(ttml[:-5] + ".npz")
    W, H = int(z["width"]), int(z["height"])
    for extra in ({}, {"FLUTL_NO_RECOLOUR": "1"}):
        pk = scene_pack.load(gstenv.dump_pack(ttml, W, H, str(tmp_path / "p.flpk"),
                                              extra_env=extra))
        st = pk.states()
        assert len(st) == int(z["n_states"])
        for i, (R, P) in enumerate(st):
            assert F.validate(W, H, R, P, pk.atlas.size) == 0
            assert np.array_equal(O.render(W, H, R, P, pk.atlas), z["frames"][i]), (extra, i)
    if os.path.basename(ttml).startswith("k18"):
        counts = {}
        for extra in ({}, {"FLUTL_NO_RECOLOUR": "1"}):
            env = gstenv.env_for(gstenv.HIP_PLUGIN_DIR, dict(
                extra, FLUTL_TIMING="1", FLUTL_DUMP_ONLY="1", FLUTL_DUMP_PACK=os.devnull))
            r = subprocess.run([os.path.join(gstenv.CONDA, "bin", "gst-launch-1.0"), "-q",
                                "filesrc", f"location={ttml}", "!", "ttmlrender", "!",
                                f"video/x-raw,width={W},height={H}", "!", "fakesink"],
                               env=env, capture_output=True, text=True, cwd=str(tmp_path))
            assert r.returncode == 0, r.stderr[-800:]
            counts[bool(extra)] = int(re.search(r"recoloured=(\d+)", r.stderr).group(1))
        assert counts[False] >= 3 and counts[True] == 0, counts


CAIRO_DEV = os.path.exists(os.path.join(gstenv.CONDA, "include", "cairo", "cairo.h"))


@pytest.mark.skipif(not CAIRO_DEV, reason="cairo development files not present")
@pytest.mark.parametrize("mode,what", [(0, "random paths"), (1, "random paths cut by a clip box"),
                                       (2, "lines of text"), (3, "text cut by a clip box"),
                                       (4, "random paths, filled"), (5, "filled, cut by a clip box"),
                                       (6, "lines of text, filled"), (7, "filled text cut by a clip box")])
def test_stroker_and_scan_restatement_are_cairo_stroke(mode, what):
    """The product's host-side stroker (host/flutl_stroker.c: path -> the polygon Cairo
    makes of a stroke; modes 4-7: of a FILL, flutl_fill_to_polygon) and the scan converter's
    restatement (oracle/flutl_tor_oracle.c) against the REAL cairo_stroke / cairo_fill of this
    image, byte for byte: oracle/stroke_check.c (560 000 strokes and 172 000 fills when they
    were written; a few hundred per mode here)."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_build/stroke_check"],
                          stdout=subprocess.DEVNULL)
    n = 1500 if (mode & 3) < 2 else 250
    r = subprocess.run([os.path.join(ROOT, "oracle", "_build", "stroke_check"), str(n),
                        str(4242 + mode), str(mode)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and f"{n} paths, 0 differ" in r.stdout, r.stdout[-1500:]


@pytest.mark.skipif(not CAIRO_DEV, reason="cairo development files not present")
def test_scan_restatement_is_cairo_fill():
    """oracle/tor_check.c: random polygons (coarse grids: equal sort keys everywhere)
    through the real cairo_fill and through the restatement."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_build/tor_check"],
                          stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(ROOT, "oracle", "_build", "tor_check"), "3000", "99"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "3000 polygons, 0 differ" in r.stdout, r.stdout[-1500:]
