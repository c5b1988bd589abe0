"""GPU (MI355X): the HIP compositor reproduces the REFERENCE plugin's frames,
bit for bit, from the display lists the element recorded (small fixtures), and
the SHA-256 of the reference's full-size frames for the BASELINE configs.
Where the GStreamer runtime is present on the box, the whole element is also
run end to end (TTML in, BGRA frames out of the HIP kernel)."""
import glob
import hashlib
import json
import os
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import pyflutl as F
import scene_pack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gstenv  # noqa: E402

SCENES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "scenes", "*.npz")))
# cfg2_seed20..27: the eight streams of BASELINE config 3, one per GPU at N = 8
CONFIGS = [c for c in ["cfg0", "cfg1", "cfg2", "cfg4"] + [f"cfg2_seed{s}" for s in range(20, 28)]
           if os.path.exists(os.path.join(ROOT, "bench_data", c + ".flpk.z"))]


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available()
    import gpu_util
    return gpu_util


@pytest.mark.parametrize("path", SCENES, ids=[os.path.basename(s)[:-4] for s in SCENES])
def test_kernel_reproduces_reference_frames(gpu, path):
    z = np.load(path)
    W, H = int(z["width"]), int(z["height"])
    for i in range(int(z["n_states"])):
        got = gpu.hip_render(W, H, z[f"regions{i}"], z[f"prims{i}"], z["atlas"])
        assert np.array_equal(got, z["frames"][i]), \
            f"state {i}: {(got != z['frames'][i]).sum()} px differ from the reference"


@pytest.mark.parametrize("cfg", CONFIGS)
def test_full_size_config_matches_reference_sha256(gpu, cfg):
    """BASELINE.json configs at full size: every frame's SHA-256 equals that of
    the frame the reference plugin rendered from the same TTML."""
    pk = scene_pack.load(os.path.join(ROOT, "bench_data", cfg + ".flpk.z"))
    meta = json.load(open(os.path.join(ROOT, "bench_data", cfg + ".sha256.json")))
    states = pk.states()
    assert len(states) == meta["states"]
    step = max(1, len(states) // 12)          # a spread of states, incl. last
    for i in sorted(set(list(range(0, len(states), step)) + [len(states) - 1])):
        R, P = states[i]
        got = gpu.hip_render(pk.width, pk.height, R, P, pk.atlas)
        assert int((got != 0).sum()) == meta["nonzero_pixels"][i]
        assert hashlib.sha256(got.tobytes()).hexdigest() == meta["sha256"][i], (cfg, i)


def run_pipeline(cmd, env, cwd, limit=90):
    """One gst-launch, ONE attempt.  Several elements in one process hung twice in some 700
    runs on the GPU box in round 3 -- libxml2 initialised lazily by racing streaming threads,
    fixed by xmlInitParser in class_init.  A pipeline still running after
    `limit` seconds has its threads' backtraces written to gpurun_out/hang_backtraces.txt
    (rocgdb, if it may attach), is killed, and FAILS the test: a recurrence must show in the
    record, and its cause is then read from those backtraces."""
    import subprocess
    # (a debugger that is not the pipeline's parent may attach only if the pipeline says so)
    so = os.path.join(tempfile.gettempdir(), "flutl_ptracer_any.so")
    if not os.path.exists(so):
        src = so[:-3] + ".c"
        with open(src, "w") as f:
            f.write("#include <sys/prctl.h>\n__attribute__((constructor)) static void any(void)"
                    "{ prctl(PR_SET_PTRACER, PR_SET_PTRACER_ANY, 0, 0, 0); }\n")
        try:
            subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", src, "-o", so])
        except Exception:       # noqa: BLE001
            so = None
    if so:
        env = dict(env, LD_PRELOAD=(env.get("LD_PRELOAD", "") + ":" + so).strip(":"))
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=cwd)
    try:
        out, err = p.communicate(timeout=limit)
        return subprocess.CompletedProcess(cmd, p.returncode, out, err)
    except subprocess.TimeoutExpired:
        bt = ""
        if os.path.exists("/opt/rocm/bin/rocgdb"):
            try:
                bt = subprocess.run(["/opt/rocm/bin/rocgdb", "-q", "-batch", "-p", str(p.pid), "-ex",
                                     "set pagination off", "-ex", "thread apply all bt 30"],
                                    capture_output=True, text=True, timeout=120).stdout
            except Exception as e:      # noqa: BLE001
                bt = repr(e)
        p.kill()
        p.communicate()
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "hang_backtraces.txt"), "a") as f:
            f.write("$ " + " ".join(cmd) + "\n" + bt[-80000:] + "\n")
        raise AssertionError(f"pipeline still running after {limit} s (killed; backtraces in "
                             f"gpurun_out/hang_backtraces.txt): {' '.join(cmd[:12])} ...")


@pytest.mark.skipif(not gstenv.available(gstenv.HIP_PLUGIN_DIR),
                    reason="fluttml-hip plugin / GStreamer runtime not present")
@pytest.mark.parametrize("path", SCENES, ids=[os.path.basename(s)[:-4] for s in SCENES])
def test_element_end_to_end_in_a_gstreamer_pipeline(gpu, path):
    """filesrc ! ttmlrender(HIP) ! capsfilter ! filesink == reference frames."""
    z = np.load(path)
    W, H = int(z["width"]), int(z["height"])
    raw = gstenv.render(gstenv.HIP_PLUGIN_DIR, path[:-4] + ".ttml", W, H)
    got = np.frombuffer(raw, np.uint32).reshape(-1, H, W)
    assert np.array_equal(got, z["frames"])


@pytest.mark.skipif(not gstenv.available(gstenv.HIP_PLUGIN_DIR),
                    reason="fluttml-hip plugin / GStreamer runtime not present")
def test_element_overlay_composition_knob_keeps_the_frames(gpu):
    """FLUTL_OVERLAY_COMPOSITION=1 attaches a GstVideoOverlayCompositionMeta (an
    extension; the reference has none): the pushed frames must not change."""
    path = SCENES[0]
    z = np.load(path)
    W, H = int(z["width"]), int(z["height"])
    raw = gstenv.render(gstenv.HIP_PLUGIN_DIR, path[:-4] + ".ttml", W, H,
                        extra_env={"FLUTL_OVERLAY_COMPOSITION": "1"})
    got = np.frombuffer(raw, np.uint32).reshape(-1, H, W)
    assert np.array_equal(got, z["frames"])


@pytest.mark.skipif(not gstenv.available(gstenv.HIP_PLUGIN_DIR),
                    reason="fluttml-hip plugin / GStreamer runtime not present")
@pytest.mark.parametrize("name", ["k06_animation", "k08_images", "g02_blur_fractional"])
def test_element_survives_dropping_its_mask_caches(gpu, name):
    """A live stream is one endless document: the element bounds its atlas by dropping
    the mask caches at a frame boundary.  Forced here before every frame."""
    path = [s for s in SCENES if name in s][0]
    z = np.load(path)
    W, H = int(z["width"]), int(z["height"])
    raw = gstenv.render(gstenv.HIP_PLUGIN_DIR, path[:-4] + ".ttml", W, H,
                        extra_env={"FLUTL_ATLAS_LIMIT_BYTES": "64"})
    got = np.frombuffer(raw, np.uint32).reshape(-1, H, W)
    assert np.array_equal(got, z["frames"])


@pytest.mark.skipif(not gstenv.available(gstenv.HIP_PLUGIN_DIR),
                    reason="fluttml-hip plugin / GStreamer runtime not present")
def test_element_renders_embedded_images_and_exits_cleanly(gpu):
    """Embedded smpte:image data (the reference loses it to a memory bug of its base
    class and aborts at teardown): same frames as the same PNGs referenced as
    files, which the reference renders (k08), and exit code 0."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "scenes", "k08_images.npz"))
    W, H = int(z["width"]), int(z["height"])
    raw = gstenv.render(gstenv.HIP_PLUGIN_DIR, os.path.join(
        ROOT, "tests", "golden", "variants", "k08_images_embedded.ttml"), W, H)
    got = np.frombuffer(raw, np.uint32).reshape(-1, H, W)
    assert np.array_equal(got, z["frames"])


@pytest.mark.skipif(not gstenv.available(gstenv.HIP_PLUGIN_DIR),
                    reason="fluttml-hip plugin / GStreamer runtime not present")
@pytest.mark.parametrize("name", ["k06_animation", "k02_opacity_zorder", "k08_images",
                                  "g01_blur_outline"])
def test_element_fused_over_video_through_its_request_pad(gpu, name):
    """videotestsrc ! r.video_sink  filesrc ! ttmlrender name=r ! filesink
    (FLUTL_VIDEO_PAD=1).  Every output frame must be the REFERENCE's overlay frame of
    that time composited OVER the video frame with pixman's rule -- what the
    reference's README pipeline gets from a compositor after ttmlrender
    (plugins/ttml/README.md:47-50)."""
    import oracle_lib as O
    path = [s for s in SCENES if name in s][0]
    z = np.load(path)
    W, H = int(z["width"]), int(z["height"])
    ttml = path[:-4] + ".ttml"
    fps, n = 4, 15
    vraw, oraw = gstenv.render_over_video(ttml, W, H, n, fps, pattern="smpte")
    video = np.frombuffer(vraw, np.uint32).reshape(n, H, W)
    out = np.frombuffer(oraw, np.uint32).reshape(n, H, W)
    assert (video >> 24 == 255).all()
    times = gstenv.buffer_times(gstenv.HIP_PLUGIN_DIR, ttml, W, H)
    assert len(times) == int(z["n_states"])
    for k in range(n):
        t = k * 10**9 // fps
        state = [i for i, (p, d) in enumerate(times) if p <= t and (d is None or t < p + d)]
        if state:
            exp = O.over(z["frames"][state[0]].reshape(-1), video[k].reshape(-1)).reshape(H, W)
        else:
            exp = video[k]
        assert np.array_equal(out[k], exp), (k, state, int((out[k] != exp).sum()))


@pytest.mark.skipif(not gstenv.available(gstenv.HIP_PLUGIN_DIR),
                    reason="fluttml-hip plugin / GStreamer runtime not present")
@pytest.mark.parametrize("names", [["k04_outline", "k12_moving_outline"],
                                   ["k16_unknown_glyphs", "k18_recolour", "k13_vertical_group",
                                    "k20_hexbox_vertical", "k06_animation"]],
                         ids=["two", "five"])
def test_two_elements_in_one_process_and_a_thread_boundary_downstream(gpu, tmp_path, names):
    """Several ttmlrender instances in ONE pipeline process (each with its own context,
    finisher thread, frame pool and -- unlike the reference's -- font map; the stroke
    workers are shared), each followed by a queue: buffers pile up downstream and are
    mapped on another thread than the one that made them.  All outputs must be the
    reference's frames."""
    import subprocess
    outs = [str(tmp_path / (n + ".raw")) for n in names]
    cmd = [os.path.join(gstenv.CONDA, "bin", "gst-launch-1.0"), "-q", "--no-fault"]
    for n, o in zip(names, outs):
        z = np.load(os.path.join(ROOT, "tests", "golden", "scenes", n + ".npz"))
        cmd += ["filesrc", "location=" + os.path.join(ROOT, "tests", "golden", "scenes", n + ".ttml"),
                "!", "ttmlrender", "!", f"video/x-raw,width={int(z['width'])},height={int(z['height'])}",
                "!", "queue", "max-size-buffers=64", "!", "filesink", "location=" + o]
    r = run_pipeline(cmd, gstenv.env_for(gstenv.HIP_PLUGIN_DIR), str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    for n, o in zip(names, outs):
        z = np.load(os.path.join(ROOT, "tests", "golden", "scenes", n + ".npz"))
        got = np.fromfile(o, np.uint32).reshape(-1, int(z["height"]), int(z["width"]))
        assert np.array_equal(got, z["frames"]), n


@pytest.mark.skipif(not gstenv.available(gstenv.HIP_PLUGIN_DIR),
                    reason="fluttml-hip plugin / GStreamer runtime not present")
@pytest.mark.parametrize("knobs", [{}, {"FLUTL_DEVICE_MAP": "0,0,0"}, {"FLUTL_DEVICE": "0"}],
                         ids=["round-robin", "map", "pinned"])
def test_instances_of_one_process_take_the_nodes_devices_in_turn(gpu, tmp_path, knobs):
    """Each ttmlrender picks its HIP device when it makes its context: the k-th instance of
    the process takes device k mod the device count (FLUTL_DEVICE_MAP: the k-th entry of the
    map; FLUTL_DEVICE: that one for all).  Three instances in one pipeline process: their
    frames are the reference's, and their timing lines name the devices -- 0, 1, 2 where the
    box has that many, all 0 on the one-GPU box (and under the map / the pinned device)."""
    import re
    names = ["k04_outline", "k06_animation", "k16_unknown_glyphs"]
    outs = [str(tmp_path / (n + ".raw")) for n in names]
    cmd = [os.path.join(gstenv.CONDA, "bin", "gst-launch-1.0"), "-q", "--no-fault"]
    for n, o in zip(names, outs):
        z = np.load(os.path.join(ROOT, "tests", "golden", "scenes", n + ".npz"))
        cmd += ["filesrc", "location=" + os.path.join(ROOT, "tests", "golden", "scenes", n + ".ttml"),
                "!", "ttmlrender", "!", f"video/x-raw,width={int(z['width'])},height={int(z['height'])}",
                "!", "filesink", "location=" + o]
    r = run_pipeline(cmd, gstenv.env_for(gstenv.HIP_PLUGIN_DIR, dict(knobs, FLUTL_TIMING="1")), str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    for n, o in zip(names, outs):
        z = np.load(os.path.join(ROOT, "tests", "golden", "scenes", n + ".npz"))
        got = np.fromfile(o, np.uint32).reshape(-1, int(z["height"]), int(z["width"]))
        assert np.array_equal(got, z["frames"]), n
    devs = sorted(int(d) for d in re.findall(r"ttmlrender-hip timing: device=(\d+) buffers=[1-9]", r.stderr))
    assert len(devs) == 3, r.stderr[-1500:]
    n_dev = F.lib().flutl_device_count()
    want = sorted(k % n_dev for k in range(3)) if not knobs else [0, 0, 0]
    assert devs == want, (devs, want, n_dev)


@pytest.mark.skipif(not gstenv.available(gstenv.HIP_PLUGIN_DIR),
                    reason="fluttml-hip plugin / GStreamer runtime not present")
@pytest.mark.parametrize("knobs", [{"FLUTL_SYNC": "1"}, {"FLUTL_NO_REGION_CACHE": "1"},
                                   {"FLUTL_MAX_AHEAD": "1", "FLUTL_STROKE_THREADS": "1"},
                                   {"FLUTL_FULL_D2H": "1"}, {"FLUTL_POOL_PREFILL": "0"},
                                   {"FLUTL_NO_SPAN_PLANS": "1", "FLUTL_NO_RUN_MERGE": "1"},
                                   {"FLUTL_QUEUE_ALWAYS": "0", "FLUTL_NO_LAYOUT_CACHE": "1"},
                                   {"FLUTL_NO_RECOLOUR": "1"}, {"FLUTL_HOST_STROKES": "1"},
                                   {"FLUTL_NO_COMPOSED_PATHS": "1"},
                                   {"FLUTL_CHECK_PATHS": "1", "FLUTL_MAX_AHEAD": "16"},
                                   {"FLUTL_WRITABLE": "1"}, {"FLUTL_SCAN_FUSED": "1"},
                                   {"FLUTL_NO_EDGE_BLOCKS": "1"}, {"FLUTL_ONE_STREAM": "1"},
                                   {"FLUTL_NO_FRAME_ARENA": "1"}, {"FLUTL_LIST_COPY": "1"},
                                   {"FLUTL_ONE_LANE": "1"}],
                         ids=["sync", "no-region-cache", "ahead1", "full-d2h", "no-prefill",
                              "no-plans-no-merge", "direct-when-idle", "no-recolour", "host-strokes",
                              "paths-by-cairo", "paths-checked-ahead16", "writable", "scan-fused",
                              "no-edge-blocks", "one-stream", "no-frame-arena", "list-copied", "one-lane"])
def test_element_modes_all_reproduce_the_reference(gpu, knobs):
    """The element's pipeline knobs change how a buffer is produced, never what it
    holds: an outlined, animated scene and a moving outline under each of them."""
    for name in ("k06_animation", "k12_moving_outline", "k13_vertical_group", "k17_glyph_runs",
                 "k18_recolour"):
        path = [s for s in SCENES if name in s][0]
        z = np.load(path)
        W, H = int(z["width"]), int(z["height"])
        raw = gstenv.render(gstenv.HIP_PLUGIN_DIR, path[:-4] + ".ttml", W, H, extra_env=knobs)
        got = np.frombuffer(raw, np.uint32).reshape(-1, H, W)
        assert np.array_equal(got, z["frames"]), (name, knobs)


@pytest.mark.skipif(not gstenv.available(gstenv.HIP_PLUGIN_DIR),
                    reason="fluttml-hip plugin / GStreamer runtime not present")
def test_promise_buffers_through_a_converting_element(gpu, tmp_path):
    """The element's buffers are promises in its own GstMemory (map waits).  A stock
    transform element downstream must get the right pixels without knowing:
    ttmlrender ! videoconvert ! RGBA ! filesink == the reference's frames with B and R
    swapped."""
    import subprocess
    name = "k06_animation"
    z = np.load(os.path.join(ROOT, "tests", "golden", "scenes", name + ".npz"))
    W, H = int(z["width"]), int(z["height"])
    out = str(tmp_path / "rgba.raw")
    cmd = [os.path.join(gstenv.CONDA, "bin", "gst-launch-1.0"), "-q", "--no-fault", "filesrc",
           "location=" + os.path.join(ROOT, "tests", "golden", "scenes", name + ".ttml"), "!",
           "ttmlrender", "!", f"video/x-raw,width={W},height={H}", "!", "videoconvert", "!",
           "video/x-raw,format=RGBA", "!", "filesink", "location=" + out]
    r = run_pipeline(cmd, gstenv.env_for(gstenv.HIP_PLUGIN_DIR), str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.fromfile(out, np.uint8).reshape(-1, H, W, 4)
    exp = z["frames"].view(np.uint8).reshape(-1, H, W, 4)[..., [2, 1, 0, 3]]
    assert np.array_equal(got, exp)


@pytest.mark.skipif(not gstenv.available(gstenv.HIP_PLUGIN_DIR),
                    reason="fluttml-hip plugin / GStreamer runtime not present")
@pytest.mark.parametrize("doc,knobs", [("cfg2", {}), ("typical_4k", {}),
                                       ("cfg2", {"FLUTL_QUEUE_ALWAYS": "0"}),
                                       ("typical_4k", {"FLUTL_SYNC": "1"}),
                                       ("wide_8k", {}), ("wide_8k", {"FLUTL_SYNC": "1"}),
                                       ("typical_4k", {"FLUTL_ATLAS_LIMIT_MB": "4"}),
                                       ("typical_4k", {"FLUTL_ONE_STREAM": "1"}),
                                       ("cfg2", {"FLUTL_LANES_AFTER": "8"}),
                                       ("cfg2", {"FLUTL_ONE_LANE": "1"})],
                         ids=["cfg2", "typical_4k", "cfg2-direct-when-idle", "typical_4k-sync", "wide_8k",
                              "wide_8k-sync", "typical_4k-atlas-starting-over", "typical_4k-one-stream",
                              "cfg2-lanes-as-shipped", "cfg2-one-lane"])
def test_element_at_full_size_is_the_reference_s(gpu, doc, knobs):
    """The documents bench.py's end-to-end figures are measured on, through the real
    pipeline at 3840x2160 (`filesrc ! ttmlrender ! caps ! fdsink`, frames read from the
    pipe): SHA-256 of EVERY frame equals that of the frame the reference plugin rendered
    from the same TTML (bench_data/<doc>.e2e_sha256.json, tools/make_e2e_sha.py) -- 241
    frames of cfg2 (regions played back in other colours, delta read-back into pooled
    frames) and 121 of typical_4k (an outline stroked on a worker thread per cue); and
    wide_8k at 7680x4320: lines of 180 px text whose outlines are 6 000 pixels wide, beyond
    the scan converter's 4 096-pixel boxes (converted in column windows).  typical_4k also
    with the atlas starting over every few cues (4 MB: its bytes are used again while
    conversions and compositor launches are in flight on their two streams)."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_e2e_sha
    meta = json.load(open(os.path.join(ROOT, "bench_data", doc + ".e2e_sha256.json")))
    shas, rc, err = make_e2e_sha.stream_sha(
        gstenv.HIP_PLUGIN_DIR, os.path.join(ROOT, "bench_data", doc + ".ttml"),
        meta["width"], meta["height"], extra_env=knobs)
    assert rc == 0, err[-1500:]
    assert len(shas) == meta["frames"]
    bad = [i for i, (a, b) in enumerate(zip(shas, meta["sha256"])) if a != b]
    assert not bad, (doc, knobs, bad[:10])


@pytest.mark.skipif(not gstenv.available(gstenv.HIP_PLUGIN_DIR),
                    reason="fluttml-hip plugin / GStreamer runtime not present")
@pytest.mark.parametrize("cfg,n", [("cfg0", None), ("cfg1", None), ("cfg4", 24)])
def test_element_at_8k_is_the_reference_s(gpu, cfg, n):
    """... and the first 24 buffers of cfg4.ttml at 7680x4320 (133 MB frames: 16 overlapping
    regions, translucent groups parking pixels in the scratch frame, the pool bounded by
    size), all states of cfg0 (720p) and cfg1 (1080p), against the per-state hashes of the
    reference's frames (bench_data/<cfg>.sha256.json)."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_e2e_sha
    meta = json.load(open(os.path.join(ROOT, "bench_data", cfg + ".sha256.json")))
    n = n or meta["states"]
    shas, rc, err = make_e2e_sha.stream_sha(
        gstenv.HIP_PLUGIN_DIR, os.path.join(ROOT, "bench_data", cfg + ".ttml"),
        meta["width"], meta["height"], max_frames=n, stop_after=True)
    assert rc == 0 and len(shas) == n, err[-1500:]
    assert shas == meta["sha256"][:n], [i for i in range(n) if shas[i] != meta["sha256"][i]]


@pytest.mark.skipif(not gstenv.available(gstenv.HIP_PLUGIN_DIR),
                    reason="fluttml-hip plugin / GStreamer runtime not present")
def test_element_fused_over_video_at_full_size(gpu):
    """The request pad at 3840x2160: 12 frames of videotestsrc (2 per second) with cfg2.ttml
    composited into them by the fused kernel.  Expected, as SHA-256 per frame
    (bench_data/cfg2.video_sha256.json, tools/make_e2e_sha.py video): the reference's overlay
    frame of that time OVER the video frame with pixman's rule."""
    import hashlib
    import json
    meta = json.load(open(os.path.join(ROOT, "bench_data", "cfg2.video_sha256.json")))
    W, H, n = meta["width"], meta["height"], meta["frames"]
    log = []
    vraw, oraw = gstenv.render_over_video(os.path.join(ROOT, "bench_data", meta["document"]),
                                          W, H, n, meta["fps"], pattern=meta["pattern"],
                                          extra_env={"FLUTL_TIMING": "1"}, log=log)
    size = W * H * 4
    assert len(oraw) == n * size and len(vraw) == n * size
    got = [hashlib.sha256(oraw[k * size:(k + 1) * size]).hexdigest() for k in range(n)]
    assert got == meta["sha256"], [k for k in range(n) if got[k] != meta["sha256"][k]]
    # the cues' outlines came from the workers as polygons for the device: none by Cairo
    import re
    m = re.search(r"polygons=(\d+).* masks_by_cairo=(\d+)", log[0])
    assert m and int(m.group(1)) > 0 and int(m.group(2)) == 0, log[0][-1500:]


@pytest.mark.skipif(not gstenv.available(gstenv.HIP_PLUGIN_DIR),
                    reason="fluttml-hip plugin / GStreamer runtime not present")
@pytest.mark.parametrize("mode,knobs", [("none", {}), ("read", {}), ("write", {}),
                                        ("write", {"FLUTL_WRITABLE": "1"})],
                         ids=["dropped", "read", "write-readonly", "write-writable"])
def test_a_consumer_that_maps_the_buffers(gpu, tmp_path, mode, knobs):
    """ttmlrender ! probe ! fakesink, the probe (tools/map_cost.c: a GstBaseTransform) passing on
    / reading / working in place in every buffer.  READONLY memory (the default): GStreamer
    copies each frame an in-place element touches; FLUTL_WRITABLE=1: it works in the frame's
    own block, and the element puts block and twin together again when the frame is next
    used.  Every buffer of the document arrives either way."""
    import re
    import map_cost
    import gen_config_ttml
    exe = map_cost.build()
    doc = str(tmp_path / "cfg2_300.ttml")
    with open(doc, "w") as f:
        f.write(gen_config_ttml.gen("cfg2", 2, n_states=300))
    r = run_pipeline([exe, doc, "3840", "2160", mode], gstenv.env_for(gstenv.HIP_PLUGIN_DIR, knobs),
                     str(tmp_path), limit=60)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    m = re.search(r"buffers=(\d+) frames_per_s=([0-9.]+) memories_replaced_by_copies=(\d+)", r.stdout)
    assert m and int(m.group(1)) == 601, r.stdout
    if mode == "write":     # READONLY memory: GStreamer copies each frame; writable: none
        assert (int(m.group(3)) == 0) == bool(knobs), r.stdout
