"""Scene packs: display lists + glyph atlas for a sequence of scene states of
one TTML stream, as produced by the host side of the element
(host/gstttmlrender_hip.c, FLUTL_DUMP_DIR) from a TTML document.

File format (.flpk, little endian), written by host/flutl_pack.c:
    magic  "FLPK1\\0\\0\\0"
    u32 width, height, n_states, atlas_bytes
    per state: u32 n_regions, n_prims; regions[n_regions]; prims[n_prims]
    atlas bytes (incl. the 16-byte leading/trailing slack)
Packs under bench_data/ are zlib-compressed (.flpk.z).
"""
import os
import zlib

import numpy as np

import pyflutl as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"FLPK1\0\0\0"


class ScenePack:
    def __init__(self, width, height, state_list, atlas, source):
        self.width, self.height = width, height
        self._states = state_list
        self.atlas = atlas
        self.source = source

    def states(self, n=None):
        s = self._states
        if n is None or n >= len(s):
            return list(s)
        return list(s[:n])


def loads(buf, source="pack"):
    assert buf[:8] == MAGIC, "not a scene pack"
    w, h, n, ab = np.frombuffer(buf, "<u4", 4, 8)
    off = 24
    states = []
    for _ in range(int(n)):
        nr, npr = np.frombuffer(buf, "<u4", 2, off)
        off += 8
        R = np.frombuffer(buf, F.REGION_DTYPE, int(nr), off).copy()
        off += int(nr) * 16
        P = np.frombuffer(buf, F.PRIM_DTYPE, int(npr), off).copy()
        off += int(npr) * 32
        states.append((R, P))
    atlas = np.frombuffer(buf, np.uint8, int(ab), off).copy()
    return ScenePack(int(w), int(h), states, atlas, source)


def dumps(width, height, states, atlas):
    out = [MAGIC, np.array([width, height, len(states), atlas.size], "<u4").tobytes()]
    for (R, P) in states:
        out.append(np.array([len(R), len(P)], "<u4").tobytes())
        out.append(np.ascontiguousarray(R, F.REGION_DTYPE).tobytes())
        out.append(np.ascontiguousarray(P, F.PRIM_DTYPE).tobytes())
    out.append(np.ascontiguousarray(atlas, np.uint8).tobytes())
    return b"".join(out)


def load(path):
    buf = open(path, "rb").read()
    if path.endswith(".z"):
        buf = zlib.decompress(buf)
    return loads(buf, source=os.path.relpath(path, ROOT))


def find_pack(config, stream_seed):
    """bench_data/<config>_seed<seed>.flpk.z, else the seed-less pack, else None."""
    d = os.path.join(ROOT, "bench_data")
    names = [f"{config}.flpk.z"]
    if stream_seed is not None:
        names.insert(0, f"{config}_seed{stream_seed}.flpk.z")
    for name in names:
        p = os.path.join(d, name)
        if os.path.exists(p):
            return load(p)
    return None
