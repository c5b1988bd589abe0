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
    reg = os.path.join(tempfile.gettempdir(), "flutl-gst-registry-%s-%d-%d.bin" % (
        os.path.basename(os.path.dirname(plugin_dir + "/")), os.getuid(), stamp))
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
