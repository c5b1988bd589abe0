"""The drop-in boundary of the element: factory name, rank, klass,
pads/caps and properties with their defaults must be what the reference's
ttmlrender exposes.  Compared LIVE against the reference plugin where both
plugins and the GStreamer tools of this image are present; skipped elsewhere."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gstenv  # noqa: E402

need = pytest.mark.skipif(
    not (gstenv.available(gstenv.REF_PLUGIN_DIR) and gstenv.available(gstenv.HIP_PLUGIN_DIR)),
    reason="needs both plugins + gst-inspect of this image")


def inspect(plugin_dir, what="ttmlrender"):
    r = subprocess.run([os.path.join(gstenv.CONDA, "bin", "gst-inspect-1.0"), what],
                       env=gstenv.env_for(plugin_dir), capture_output=True, text=True,
                       cwd="/tmp")
    assert r.returncode == 0, r.stderr
    return r.stdout


def section(text, start, end=None):
    i = text.index(start)
    j = text.index(end, i) if end else len(text)
    return text[i:j]


@need
def test_factory_rank_klass_and_hierarchy():
    ref, hip = inspect(gstenv.REF_PLUGIN_DIR), inspect(gstenv.HIP_PLUGIN_DIR)
    for key in ("Rank", "Klass", "Long-name"):
        get = lambda t: re.search(rf"{key}\s+(.*)", t).group(1).strip()
        assert get(ref) == get(hip), key
    assert "marginal" in re.search(r"Rank\s+(.*)", hip).group(1)
    assert section(ref, "GObject", "Pad Templates") == section(hip, "GObject", "Pad Templates")


@need
def test_pad_templates_and_caps_are_identical():
    ref, hip = inspect(gstenv.REF_PLUGIN_DIR), inspect(gstenv.HIP_PLUGIN_DIR)
    assert section(ref, "Pad Templates:", "Element has") == \
        section(hip, "Pad Templates:", "Element has")
    assert "application/ttml+xml" in hip and "format: BGRA" in hip


@need
def test_properties_and_defaults_are_identical():
    ref, hip = inspect(gstenv.REF_PLUGIN_DIR), inspect(gstenv.HIP_PLUGIN_DIR)
    pr, ph = section(ref, "Element Properties:"), section(hip, "Element Properties:")
    assert pr == ph
    for must in ('default-font-family', 'Default: "Serif"', 'window-width',
                 'Default: 720', 'Default: 576', 'assume-ordered-spans'):
        assert must in ph, must


@need
def test_plugin_registers_all_three_elements():
    out = inspect(gstenv.HIP_PLUGIN_DIR, "fluttml")
    for el in ("ttmlparse", "ttmlsegmentedparse", "ttmlrender"):
        assert el in out


@need
def test_video_pad_knob_adds_one_request_template_and_nothing_else():
    """FLUTL_VIDEO_PAD=1 (fused over-video) adds the request sink pad
    'video_sink'; properties, the always pads and everything else stay the
    reference's.  Without the knob the pad templates are identical (test above)."""
    r = subprocess.run([os.path.join(gstenv.CONDA, "bin", "gst-inspect-1.0"), "ttmlrender"],
                       env=gstenv.env_for(gstenv.HIP_PLUGIN_DIR, {"FLUTL_VIDEO_PAD": "1"}),
                       capture_output=True, text=True, cwd="/tmp")
    assert r.returncode == 0, r.stderr
    hip, ref = r.stdout, inspect(gstenv.REF_PLUGIN_DIR)
    assert "SINK template: 'video_sink'" in hip and "On request" in hip
    assert section(ref, "Element Properties:") == section(hip, "Element Properties:")
    assert "SINK template: 'sink'" in hip and "SRC template: 'src'" in hip


@pytest.mark.skipif(not gstenv.available(gstenv.HIP_PLUGIN_DIR),
                    reason="fluttml-hip plugin / GStreamer runtime not present")
def test_video_pad_dataflow_without_a_device():
    """The plumbing of the fused mode on CPU (record-only: cues are made, frames pass
    through): every video frame comes out once, in order, and the pipeline ends."""
    ttml = os.path.join(ROOT, "tests", "golden", "scenes", "k06_animation.ttml")
    v, o = gstenv.render_over_video(ttml, 320, 180, 9, 3, extra_env={
        "FLUTL_DUMP_ONLY": "1", "FLUTL_DUMP_PACK": os.devnull})
    assert len(v) == 9 * 320 * 180 * 4 and v == o


@pytest.mark.skipif(not gstenv.available(gstenv.HIP_PLUGIN_DIR),
                    reason="fluttml-hip plugin / GStreamer runtime not present")
def test_six_elements_in_one_process_record_side_by_side():
    """Several ttmlrender in one gst-launch (one streaming thread each).  The reference
    creates its Pango context from the DEFAULT font map of the thread that creates the
    element (gstttmlrender.c:1727): all elements of a pipeline share one font map, which
    Pango does not allow to be used from several threads -- four elements crashed or hung
    one run in three, the reference's own included.  This element has a font map of its
    own; here six of them record the same outlined, animated 4K document at once, three
    times over (record-only: no device needed), and every one must report all buffers."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import e2e_timing
    ttml = os.path.join(ROOT, "bench_data", "cfg2.ttml")
    for _ in range(3):
        r = e2e_timing.measure_streams(ttml, 3840, 2160, 6, extra_env={
            "FLUTL_DUMP_ONLY": "1", "MALLOC_CHECK_": "3"})
        assert r["streams"] == 6 and len(r["per_stream"]) == 6


@pytest.mark.skipif(not gstenv.available(gstenv.HIP_PLUGIN_DIR),
                    reason="fluttml-hip plugin / GStreamer runtime not present")
def test_video_mode_makes_the_cues_the_stand_alone_element_makes_buffers_of(tmp_path):
    """The fused over-video mode pushes no TTML buffers, so the base class never
    advances last_out_time -- which is what clamps the intervals of events that arrive
    out of order (<set> inside spans).  The element keeps it; the intervals it records
    for such a document (record-only, dumped as a scene pack) are the stand-alone
    element's, list by list -- not 16 times as many."""
    sys.path.insert(0, os.path.join(ROOT, "flu-plugins-oss_amd"))
    import numpy as np
    import scene_pack
    ttml = os.path.join(ROOT, "tests", "golden", "variants", "v01_sets_out_of_order.ttml")
    packs = {}
    for video in (False, True):
        pack = str(tmp_path / ("video.flpk" if video else "plain.flpk"))
        extra = {"FLUTL_DUMP_ONLY": "1", "FLUTL_DUMP_PACK": pack}
        gst = os.path.join(gstenv.CONDA, "bin", "gst-launch-1.0")
        if video:
            extra["FLUTL_VIDEO_PAD"] = "1"
            cmd = [gst, "-q", "videotestsrc", "num-buffers=6", "!",
                   "video/x-raw,format=BGRA,width=320,height=180,framerate=2/1", "!", "r.video_sink",
                   "filesrc", "location=" + ttml, "!", "ttmlrender", "name=r", "!", "fakesink"]
        else:
            cmd = [gst, "-q", "filesrc", "location=" + ttml, "!", "ttmlrender", "!",
                   "video/x-raw,width=320,height=180", "!", "fakesink"]
        r = subprocess.run(cmd, env=gstenv.env_for(gstenv.HIP_PLUGIN_DIR, extra),
                           capture_output=True, text=True, cwd=str(tmp_path), timeout=120)
        assert r.returncode == 0, r.stderr[-1000:]
        packs[video] = scene_pack.load(pack).states()
    assert len(packs[False]) == 18 and len(packs[True]) == 18, (len(packs[False]), len(packs[True]))
    for (Ra, Pa), (Rb, Pb) in zip(packs[False], packs[True]):
        assert np.array_equal(Ra, Rb) and np.array_equal(Pa, Pb)


@pytest.mark.skipif(not gstenv.available(gstenv.HIP_PLUGIN_DIR),
                    reason="fluttml-hip plugin / GStreamer runtime not present")
def test_video_mode_leaves_the_src_caps_to_the_video(tmp_path):
    """In the fused mode the src pad carries the VIDEO's caps.  The base class
    negotiates the src pad itself on the TTML thread when it finds it without caps
    right before gen_buffer (gstttmlbase.c:115, :922-975: the overlay template, BGRA at
    the window size), which would race with the video's stream-start and caps.  The TTML
    side is held back until the video's caps are on the pad.  Downstream here converts
    and constrains (videoconvert + capsfilter: wrong geometry or a second caps event
    cannot pass unnoticed as it does into filesink), the video is NOT the window size
    and the TTML thread is ready long before the video thread (delayed by a queue
    behind a slow start): every run must deliver every frame, converted, in order."""
    ttml = os.path.join(ROOT, "tests", "golden", "scenes", "k06_animation.ttml")
    gst = os.path.join(gstenv.CONDA, "bin", "gst-launch-1.0")
    W, H, N = 352, 200, 7        # (not the 720x576 window the overlay template fixates to)
    vcaps = f"video/x-raw,format=BGRA,width={W},height={H},framerate=3/1"
    plain = str(tmp_path / "plain.raw")
    env = gstenv.env_for(gstenv.HIP_PLUGIN_DIR, {
        "FLUTL_VIDEO_PAD": "1", "FLUTL_DUMP_ONLY": "1", "FLUTL_DUMP_PACK": os.devnull})
    r = subprocess.run([gst, "-q", "videotestsrc", f"num-buffers={N}", "!", vcaps, "!",
                        "videoconvert", "!", f"video/x-raw,format=RGB,width={W},height={H}", "!",
                        "filesink", "location=" + plain],
                       env=env, capture_output=True, text=True, cwd=str(tmp_path), timeout=120)
    assert r.returncode == 0, r.stderr[-1000:]
    want = open(plain, "rb").read()
    assert len(want) == N * W * H * 3
    for run in range(4):
        out = str(tmp_path / f"subs{run}.raw")
        cmd = [gst, "-v", "videotestsrc", f"num-buffers={N}", "!", vcaps, "!",
               "queue", "!", "identity", "sleep-time=20000", "!", "r.video_sink",
               "filesrc", "location=" + ttml, "!", "ttmlrender", "name=r", "!",
               "videoconvert", "!", f"video/x-raw,format=RGB,width={W},height={H}", "!",
               "filesink", "location=" + out]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=str(tmp_path),
                           timeout=120)
        assert r.returncode == 0, r.stderr[-1000:]
        assert "not-negotiated" not in r.stderr and "CRITICAL" not in r.stderr, r.stderr[-1000:]
        # the base class's own negotiation used to show as caps ahead of the video's
        # stream-start ("Sticky event misordering") and as a second set of caps
        assert "misordering" not in r.stderr, r.stderr[-1000:]
        src_caps = [l for l in r.stdout.splitlines() if "GstTTMLRender:r.GstPad:src: caps" in l]
        assert len(src_caps) == 1 and "framerate=(fraction)3/1" in src_caps[0], src_caps
        # (record-only: cues are made, the frames pass through unchanged)
        assert open(out, "rb").read() == want, f"run {run}"


@pytest.mark.skipif(not gstenv.available(gstenv.HIP_PLUGIN_DIR),
                    reason="fluttml-hip plugin / GStreamer runtime not present")
def test_composed_outline_paths_are_the_paths_cairo_builds():
    """An outline's path is composed from glyph outlines taken from Cairo once (the
    recorder's "composed outline paths") instead of built on a context and copied out of
    it for every cue.  FLUTL_CHECK_PATHS=1 builds every such path both ways inside the
    element and aborts on any difference, element by element and in the path's extents:
    a stream with new outlined text in every cue (120 cues, 96 px DejaVu Sans at 4K) and
    the animated 8-region document, record-only (no device needed), strokes deferred to
    the workers as in the real pipeline."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import e2e_timing
    for doc, strokes in (("typical_4k", 120), ("cfg2", 8)):
        r = e2e_timing.measure(os.path.join(ROOT, "bench_data", doc + ".ttml"), 3840, 2160, extra_env={
            "FLUTL_DUMP_ONLY": "1", "FLUTL_DUMP_PACK": os.devnull, "FLUTL_DUMP_DEFER": "1",
            "FLUTL_CHECK_PATHS": "1"})
        assert r["strokes"] == strokes
        assert r["composed_paths"] == strokes and r["composed_checked"] == strokes
