"""Shared helpers to run GStreamer pipelines of this image's /opt/conda stack
with either the reference plugin (oracle/_ref) or the fluttml-hip plugin."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONDA = os.environ.get("CONDA", "/opt/conda")
REF_PLUGIN_DIR = os.path.join(ROOT, "oracle", "_ref")
HIP_PLUGIN_DIR = os.path.join(ROOT, "flu-plugins-oss_amd", "lib", "gstreamer-1.0")


def available(plugin_dir):
    return (os.path.exists(os.path.join(CONDA, "bin", "gst-launch-1.0")) and
            os.path.exists(os.path.join(plugin_dir, "libgstfluttml.so")))


def env_for(plugin_dir, extra=None):
    e = dict(os.environ)
    so = os.path.join(plugin_dir, "libgstfluttml.so")
    stamp = int(os.path.getmtime(so)) if os.path.exists(so) else 0
    # one registry per plugin build: a stale registry would keep a plugin that
    # once failed to load blacklisted
    # (FLUTL_VIDEO_PAD changes the element's pad templates, which the registry caches)
    vp = "v" if (e.get("FLUTL_VIDEO_PAD") or (extra or {}).get("FLUTL_VIDEO_PAD")) else ""
    reg = os.path.join(tempfile.gettempdir(), "flutl-gst-registry-%s-%d-%d%s.bin" % (
        os.path.basename(os.path.dirname(plugin_dir + "/")), os.getuid(), stamp, vp))
    e.update({
        "GST_PLUGIN_SYSTEM_PATH_1_0": os.path.join(CONDA, "lib", "gstreamer-1.0"),
        "GST_PLUGIN_PATH_1_0": plugin_dir,
        "GST_REGISTRY": reg,
        "GST_PLUGIN_SCANNER": os.path.join(CONDA, "libexec", "gstreamer-1.0",
                                           "gst-plugin-scanner"),
        "LD_LIBRARY_PATH": os.path.join(CONDA, "lib") + ":" +
                           os.path.join(ROOT, "flu-plugins-oss_amd", "lib") + ":" +
                           e.get("LD_LIBRARY_PATH", ""),
    })
    # libamdhip64 needs a newer libstdc++ than /opt/conda ships; the system one
    # is backward compatible, so it is preloaded for the whole pipeline process.
    sys_stdcpp = "/usr/lib/x86_64-linux-gnu/libstdc++.so.6"
    if plugin_dir == HIP_PLUGIN_DIR and os.path.exists(sys_stdcpp):
        e["LD_PRELOAD"] = sys_stdcpp + (":" + e["LD_PRELOAD"] if e.get("LD_PRELOAD") else "")
    if extra:
        e.update(extra)
    return e


def render(plugin_dir, ttml_path, width, height, out_path=None, extra_env=None,
           sink="filesink"):
    """filesrc ! ttmlrender ! caps ! filesink|fakesink.  Returns raw bytes of all
    pushed frames (filesink) or b''."""
    tmp = None
    if sink == "filesink" and out_path is None:
        tmp = tempfile.NamedTemporaryFile(suffix=".raw", delete=False)
        tmp.close()
        out_path = tmp.name
    sink_desc = f"filesink location={out_path}" if sink == "filesink" else "fakesink"
    cmd = [os.path.join(CONDA, "bin", "gst-launch-1.0"), "-q", "--no-fault",
           "filesrc", f"location={os.path.abspath(ttml_path)}", "!", "ttmlrender", "!",
           f"video/x-raw,width={width},height={height}", "!"] + sink_desc.split()
    # cwd: the reference's base class can append an XML dump to ./ttml.xml
    r = subprocess.run(cmd, env=env_for(plugin_dir, extra_env), capture_output=True,
                       text=True, cwd=tempfile.gettempdir())
    # Known bug of the reference's base class: gst_ttml_state_save_data
    # (gstttmlstate.c:440-443) reallocs the XML text buffer that gstttmlbase.c
    # still owns, so documents with embedded smpte:image data abort with "double
    # free" at teardown -- after all buffers have been pushed.  Tolerated for the
    # REFERENCE plugin only; the product plugin carries a copying save_data
    # (host/flutl_state_fix.c) and must exit cleanly.
    teardown_bug = (plugin_dir == REF_PLUGIN_DIR and r.returncode == -6 and
                    "double free" in r.stderr)
    if r.returncode != 0 and not teardown_bug:
        raise RuntimeError(f"gst-launch failed (rc {r.returncode}): {r.stdout}\n{r.stderr}")
    data = b""
    if sink == "filesink" and tmp:      # a caller-provided path is left alone
        data = open(out_path, "rb").read()
        os.unlink(out_path)
    return data


def dump_pack(ttml_path, width, height, pack_path, record_only=True, extra_env=None):
    """Run the fluttml-hip element and return the path of the scene pack it
    wrote.  record_only (default) works without a GPU; with record_only=False
    the element also renders, and device-produced atlas entries (blurred
    outlines) are part of the pack."""
    if os.path.exists(pack_path):
        os.unlink(pack_path)
    env = {"FLUTL_DUMP_PACK": pack_path}
    if record_only:
        env["FLUTL_DUMP_ONLY"] = "1"
    env.update(extra_env or {})
    render(HIP_PLUGIN_DIR, ttml_path, width, height, sink="fakesink", extra_env=env)
    return pack_path


def render_over_video(ttml_path, width, height, n_frames, fps, pattern="smpte",
                      extra_env=None, log=None):
    """The element's fused over-video mode (FLUTL_VIDEO_PAD=1, request pad video_sink):
        videotestsrc ! BGRA caps ! r.video_sink   filesrc ! ttmlrender name=r ! filesink
    Returns (video frames as produced by videotestsrc alone, output frames), raw bytes."""
    outs = []
    vcaps = (f"video/x-raw,format=BGRA,width={width},height={height},framerate={fps}/1")
    for with_subs in (False, True):
        tmp = tempfile.NamedTemporaryFile(suffix=".raw", delete=False)
        tmp.close()
        src = ["videotestsrc", f"num-buffers={n_frames}", f"pattern={pattern}", "!", vcaps, "!"]
        if with_subs:
            cmd = src + ["r.video_sink", "filesrc", f"location={os.path.abspath(ttml_path)}", "!",
                         "ttmlrender", "name=r", "!", "filesink", f"location={tmp.name}"]
        else:
            cmd = src + ["filesink", f"location={tmp.name}"]
        env = dict(extra_env or {})
        env["FLUTL_VIDEO_PAD"] = "1"
        r = subprocess.run([os.path.join(CONDA, "bin", "gst-launch-1.0"), "-q", "--no-fault"] + cmd,
                           env=env_for(HIP_PLUGIN_DIR, env), capture_output=True, text=True,
                           cwd=tempfile.gettempdir(), timeout=300)
        if r.returncode != 0:
            raise RuntimeError(f"gst-launch failed (rc {r.returncode}): {r.stdout}\n{r.stderr}")
        if log is not None and with_subs:
            log.append(r.stderr)
        outs.append(open(tmp.name, "rb").read())
        os.unlink(tmp.name)
    return outs[0], outs[1]


def buffer_times(plugin_dir, ttml_path, width, height):
    """[(pts_ns, duration_ns)] of the buffers ttmlrender pushes for a document
    (fakesink silent=false -v); an open-ended last buffer has duration None."""
    import re
    cmd = [os.path.join(CONDA, "bin", "gst-launch-1.0"), "-v", "filesrc",
           f"location={os.path.abspath(ttml_path)}", "!", "ttmlrender", "!",
           f"video/x-raw,width={width},height={height}", "!", "fakesink", "silent=false"]
    r = subprocess.run(cmd, env=env_for(plugin_dir, {"FLUTL_DUMP_ONLY": "1",
                                                     "FLUTL_DUMP_PACK": os.devnull}),
                       capture_output=True, text=True, cwd=tempfile.gettempdir())

    def ns(t):
        h, m, sec = t.split(":")
        return (int(h) * 3600 + int(m) * 60) * 10**9 + int(round(float(sec) * 1e9))
    out = []
    for m in re.finditer(r"chain .*?pts: ([0-9:.]+), duration: ([0-9:.]+|none)", r.stdout):
        d = None if m.group(2) == "none" else ns(m.group(2))
        out.append((ns(m.group(1)), d if d is not None and d < 10**15 else None))
    return out
