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


def video_expectations(doc, W, H, fps, n, pattern="smpte"):
    """SHA-256 of what the element's request pad must put out for n frames of
    videotestsrc: the reference's overlay frame of each frame's time OVER the video
    frame, pixman's rule (oracle_lib.over).  -> bench_data/<doc>.video_sha256.json"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    ttml = os.path.join(ROOT, "bench_data", doc + ".ttml")
    times = gstenv.buffer_times(gstenv.REF_PLUGIN_DIR, ttml, W, H)
    want = {}
    for k in range(n):
        t = k * 10**9 // fps
        st = [i for i, (p, d) in enumerate(times) if p <= t and (d is None or t < p + d)]
        want[k] = st[0] if st else None
    need = {i for i in want.values() if i is not None}
    # the reference's frames, streamed; only the needed ones are kept
    cmd = [os.path.join(gstenv.CONDA, "bin", "gst-launch-1.0"), "-q", "filesrc",
           f"location={ttml}", "!", "ttmlrender", "!", f"video/x-raw,width={W},height={H}",
           "!", "fdsink", "fd=1"]
    p = subprocess.Popen(cmd, env=gstenv.env_for(gstenv.REF_PLUGIN_DIR), stdout=subprocess.PIPE,
                         stderr=subprocess.DEVNULL, cwd="/tmp")
    size, frames, i = W * H * 4, {}, 0
    while True:
        buf = p.stdout.read(size)
        if len(buf) < size:
            break
        if i in need:
            frames[i] = np.frombuffer(buf, np.uint32).copy()
        i += 1
        if i > max(need):
            p.kill()
            break
    p.wait()
    vcaps = f"video/x-raw,format=BGRA,width={W},height={H},framerate={fps}/1"
    v = subprocess.run([os.path.join(gstenv.CONDA, "bin", "gst-launch-1.0"), "-q", "videotestsrc",
                        f"num-buffers={n}", f"pattern={pattern}", "!", vcaps, "!", "fdsink", "fd=1"],
                       env=gstenv.env_for(gstenv.REF_PLUGIN_DIR), capture_output=True, cwd="/tmp")
    video = np.frombuffer(v.stdout, np.uint32).reshape(n, H * W)
    shas = []
    for k in range(n):
        exp = video[k] if want[k] is None else O.over(frames[want[k]], video[k].copy())
        shas.append(hashlib.sha256(np.ascontiguousarray(exp).tobytes()).hexdigest())
    out = os.path.join(ROOT, "bench_data", doc + ".video_sha256.json")
    json.dump({"document": doc + ".ttml", "width": W, "height": H, "fps": fps, "frames": n,
               "pattern": pattern, "states": [want[k] for k in range(n)], "sha256": shas},
              open(out, "w"), indent=0)
    print(out, n, "frames; states", [want[k] for k in range(n)])


if __name__ == "__main__":
    if sys.argv[1] == "video":      # python tools/make_e2e_sha.py video cfg2 3840 2160 2 12
        video_expectations(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]),
                           int(sys.argv[6]))
        sys.exit(0)
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

