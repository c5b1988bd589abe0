#!/usr/bin/env python3
"""Builder-container tool (needs /root/reference built into oracle/_ref and the
/opt/conda GStreamer stack): produces the committed fixtures.

  tests/golden/scenes/<name>.npz   for every tests/golden/scenes/<name>.ttml:
        frames rendered by the REFERENCE plugin (expected output), and the
        display lists + atlas recorded by the fluttml-hip element for the same
        document (input of the oracle and of the HIP kernel)
  bench_data/<cfg>.ttml, bench_data/<cfg>.flpk.z, bench_data/<cfg>.sha256.json
        the BASELINE.json configs at full size: generated TTML, recorded scene
        pack, and SHA-256 of every reference frame (frames are 8-133 MB each and
        are not committed)

    python tools/make_goldens.py [scenes] [cfg0 cfg1 cfg2 cfg4]
"""
import hashlib
import re
import json
import os
import subprocess
import sys
import tempfile
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tools", "tests", "flu-plugins-oss_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import gstenv            # noqa: E402
import gen_config_ttml   # noqa: E402
import oracle_lib as O   # noqa: E402
import pyflutl as F      # noqa: E402
import scene_pack        # noqa: E402

STACK = "gstreamer 1.14.0, cairo 1.16.0, pixman 0.40.0, pango 1.45.3 (/opt/conda)"
SCENE_SIZE = (320, 180)
MAX_STATES = {"cfg0": 8, "cfg1": 64, "cfg2": 120, "cfg4": 64}


def record(ttml, W, H, gpu=False):
    with tempfile.TemporaryDirectory() as d:
        return scene_pack.load(gstenv.dump_pack(ttml, W, H, os.path.join(d, "p.flpk"),
                                                record_only=not gpu))


def scenes(gpu=False, only=None):
    """k*.ttml: recorded without a device.  g*.ttml (gpu=True, run on an MI355X
    box through gpurun): scenes whose display lists contain device-produced
    atlas entries (blurred outlines); fixtures go to gpurun_out/goldens/."""
    W, H = SCENE_SIZE
    d = os.path.join(ROOT, "tests", "golden", "scenes")
    outd = d
    if gpu:
        outd = os.path.join(ROOT, "gpurun_out", "goldens")
        os.makedirs(outd, exist_ok=True)
    for name in sorted(f[:-5] for f in os.listdir(d)
                       if f.endswith(".ttml") and f.startswith("g" if gpu else "k")):
        if only and not name.startswith(only):
            continue
        ttml = os.path.join(d, name + ".ttml")
        m = re.search(r"frame:\s*(\d+)x(\d+)", open(ttml).read())     # <!-- frame: WxH -->
        W, H = (int(m.group(1)), int(m.group(2))) if m else SCENE_SIZE
        ref = np.frombuffer(gstenv.render(gstenv.REF_PLUGIN_DIR, ttml, W, H),
                            np.uint32).reshape(-1, H, W)
        pk = record(ttml, W, H, gpu)
        st = pk.states()
        assert len(st) == len(ref), (name, len(st), len(ref))
        out = {"width": W, "height": H, "n_states": len(st), "atlas": pk.atlas,
               "frames": ref, "stack": np.array(STACK)}
        bad = 0
        for i, (R, P) in enumerate(st):
            out[f"regions{i}"] = R
            out[f"prims{i}"] = P
            bad += int((O.render(W, H, R, P, pk.atlas) != ref[i]).sum())
        np.savez_compressed(os.path.join(outd, name + ".npz"), **out)
        print(f"{name}: {len(st)} states, oracle-vs-reference mismatching px: {bad}")


def config(cfg, seed=None, keep_states=None, keep_ttml=True):
    """cfg with its canonical seed -> bench_data/<cfg>.*; with another seed (the
    streams of BASELINE config 3) -> bench_data/<cfg>_seed<seed>.* (no .ttml kept)"""
    kind = cfg
    W, H = gen_config_ttml.SIZES[kind]
    d = os.path.join(ROOT, "bench_data")
    os.makedirs(d, exist_ok=True)
    if seed is None:
        seed = {"cfg0": 0, "cfg1": 1, "cfg2": 2, "cfg4": 4}[kind]
    else:
        cfg = f"{kind}_seed{seed}"
    tmpd = None
    if keep_ttml:
        ttml = os.path.join(d, cfg + ".ttml")
    else:
        tmpd = tempfile.mkdtemp()
        ttml = os.path.join(tmpd, cfg + ".ttml")
    open(ttml, "w").write(gen_config_ttml.gen(kind, seed))
    pk = record(ttml, W, H)
    st = pk.states()
    # reference frames -> SHA-256, streamed through a FIFO (cfg4 is 64 GB raw)
    import threading
    fb = W * H * 4
    keep = min(len(st), keep_states or MAX_STATES[kind])
    shas, nz, counts = [], [], {"n": 0, "bad": 0}
    fifo = os.path.join(tempfile.mkdtemp(), "frames.fifo")
    os.mkfifo(fifo)

    def reader():
        with open(fifo, "rb") as f:
            while True:
                buf = f.read(fb)
                while 0 < len(buf) < fb:
                    more = f.read(fb - len(buf))
                    if not more:
                        break
                    buf += more
                if len(buf) < fb:
                    break
                i = counts["n"]
                counts["n"] += 1
                if i < keep:
                    shas.append(hashlib.sha256(buf).hexdigest())
                    fr = np.frombuffer(buf, np.uint32).reshape(H, W)
                    nz.append(int((fr != 0).sum()))
                    if i < 3 or i == keep - 1:   # spot-check the oracle at full size
                        counts["bad"] += int((O.render(W, H, st[i][0], st[i][1], pk.atlas) != fr).sum())

    th = threading.Thread(target=reader)
    th.start()
    try:
        gstenv.render(gstenv.REF_PLUGIN_DIR, ttml, W, H, out_path=fifo)
    finally:
        th.join()
        os.unlink(fifo)
    n, bad = counts["n"], counts["bad"]
    assert n == len(st), (cfg, n, len(st))
    states = st[:keep]
    blob = zlib.compress(scene_pack.dumps(W, H, states, pk.atlas), 9)
    open(os.path.join(d, cfg + ".flpk.z"), "wb").write(blob)
    json.dump({"config": cfg, "width": W, "height": H, "states": keep,
               "reference_frames_total": n, "stack": STACK,
               "sha256": shas, "nonzero_pixels": nz},
              open(os.path.join(d, cfg + ".sha256.json"), "w"), indent=0)
    print(f"{cfg}: {n} reference frames, kept {keep}; pack {len(blob)/1e6:.2f} MB "
          f"(atlas {pk.atlas.size/1e6:.2f} MB raw); oracle spot-check mismatching px: {bad}")
    if tmpd:
        import shutil
        shutil.rmtree(tmpd)


def repack(cfg, seed=None):
    """Re-record bench_data/<cfg>[_seed<N>].flpk.z with the element as it is now,
    WITHOUT running the reference again: the kept states must hash (through the
    oracle, spot checks) to the SHA-256 the reference's frames gave.  Also stores
    the algorithmic bytes per state of the UNMERGED list (one mask per glyph
    instance, SURVEY 8d's definition) in the json: bench.py prices the roofline
    with those, not with the larger masks of merged glyph runs."""
    kind = cfg
    W, H = gen_config_ttml.SIZES[kind]
    d = os.path.join(ROOT, "bench_data")
    name = cfg if seed is None else f"{kind}_seed{seed}"
    if seed is None:
        seed = {"cfg0": 0, "cfg1": 1, "cfg2": 2, "cfg4": 4}[kind]
    meta_path = os.path.join(d, name + ".sha256.json")
    meta = json.load(open(meta_path))
    keep = meta["states"]
    import scenes as S
    with tempfile.TemporaryDirectory() as td:
        ttml = os.path.join(td, name + ".ttml")
        open(ttml, "w").write(gen_config_ttml.gen(kind, seed))
        pk = scene_pack.load(gstenv.dump_pack(ttml, W, H, os.path.join(td, "a.flpk")))
        pk0 = scene_pack.load(gstenv.dump_pack(ttml, W, H, os.path.join(td, "b.flpk"),
                                               extra_env={"FLUTL_NO_RUN_MERGE": "1"}))
    st, st0 = pk.states()[:keep], pk0.states()[:keep]
    assert len(st) == keep == len(st0)
    for i in sorted({0, 1, 2, keep // 2, keep - 1}):
        got = O.render(W, H, st[i][0], st[i][1], pk.atlas)
        assert hashlib.sha256(got.tobytes()).hexdigest() == meta["sha256"][i], (name, i)
    meta["algorithmic_bytes"] = [int(S.algorithmic_bytes(W, H, R, P)) for (R, P) in st0]
    meta["prims_per_state"] = [int(len(P)) for (_, P) in st]
    json.dump(meta, open(meta_path, "w"), indent=0)
    blob = zlib.compress(scene_pack.dumps(W, H, st, pk.atlas), 9)
    open(os.path.join(d, name + ".flpk.z"), "wb").write(blob)
    print(f"{name}: repacked {keep} states, {sum(len(P) for _, P in st)} prims "
          f"(unmerged {sum(len(P) for _, P in st0)}), pack {len(blob)/1e6:.2f} MB, "
          f"atlas {pk.atlas.size/1e6:.2f} MB")


if __name__ == "__main__":
    what = sys.argv[1:] or ["scenes", "cfg0", "cfg1", "cfg2"]
    for w in what:
        if w.startswith("scenes"):           # scenes, or scenes:<name prefix>
            scenes(only=w.partition(":")[2] or None)
        elif w.startswith("gpu_scenes"):
            scenes(gpu=True, only=w.partition(":")[2] or None)
        elif w.startswith("repack:"):         # repack:cfg2 or repack:cfg2:23
            parts = w.split(":")
            repack(parts[1], int(parts[2]) if len(parts) > 2 else None)
        else:
            config(w)
