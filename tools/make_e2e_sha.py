#!/usr/bin/env python3
"""Builder-container tool (needs oracle/_ref = the reference plugin built from
/root/reference): SHA-256 of every frame the REFERENCE renders for a document of
bench_data/, streamed (frames are 33 MB at 4K), into bench_data/<doc>.e2e_sha256.json --
what tests/test_gpu_scenes.py::test_element_at_full_size_is_the_reference_s checks the
element's own frames against, through a real pipeline at full size.

    python tools/make_e2e_sha.py typical_4k 3840 2160 [max_frames]"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gstenv  # noqa: E402


def stream_sha(plugin_dir, ttml, W, H, max_frames=None, extra_env=None, stop_after=False):
    """SHA-256 per frame of `filesrc ! ttmlrender ! caps ! fdsink`, read from the pipe"""
    cmd = [os.path.join(gstenv.CONDA, "bin", "gst-launch-1.0"), "-q", "filesrc",
           f"location={os.path.abspath(ttml)}", "!", "ttmlrender", "!",
           f"video/x-raw,width={W},height={H}", "!", "fdsink", "fd=1"]
    p = subprocess.Popen(cmd, env=gstenv.env_for(plugin_dir, extra_env), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, cwd="/tmp")
    size, shas = W * H * 4, []
    while True:
        buf = p.stdout.read(size)
        if len(buf) < size:
            break
        if max_frames is None or len(shas) < max_frames:
            shas.append(hashlib.sha256(buf).hexdigest())
        if stop_after and max_frames is not None and len(shas) >= max_frames:
            p.kill()            # (our own child; 8K documents are 64 GB of frames)
            p.wait()
            return shas, 0, ""
    err = p.stderr.read().decode(errors="replace")
    rc = p.wait()
    return shas, rc, err


if __name__ == "__main__":
    doc, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    n = int(sys.argv[4]) if len(sys.argv) > 4 else None
    ttml = os.path.join(ROOT, "bench_data", doc + ".ttml")
    shas, rc, err = stream_sha(gstenv.REF_PLUGIN_DIR, ttml, W, H, n)
    assert shas and rc in (0, -6), (rc, err[-500:])     # (the reference may abort at teardown)
    out = os.path.join(ROOT, "bench_data", doc + ".e2e_sha256.json")
    json.dump({"document": doc + ".ttml", "width": W, "height": H, "frames": len(shas),
               "stack": "reference ttmlrender (oracle/_ref): gstreamer 1.14.0, cairo 1.16.0, "
                        "pixman 0.40.0, pango 1.45.3", "sha256": shas}, open(out, "w"), indent=0)
    print(out, len(shas), "frames")
