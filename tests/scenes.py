"""Deterministic synthetic display lists (include/flutl.h IR) for parity tests.

These are NOT TTML scenes: they are adversarial mixes of every primitive, rule
and edge case the IR allows (overlapping regions, nested clipping, unaligned
boxes, masks at odd byte offsets, groups, ragged frame sizes), used to hold the
HIP kernel to the oracle on the same seeded input.  Real TTML scenes rendered
by the reference plugin live under tests/golden/scenes/.
"""
import numpy as np

import pyflutl as F


class AtlasBuilder:
    def __init__(self):
        self.buf = bytearray(F.ATLAS_SLACK)

    def put(self, arr, lead=0):
        """append arr 16-byte aligned (+ `lead` bytes to force odd alignment)"""
        arr = np.ascontiguousarray(arr)
        pad = (-len(self.buf)) % 16
        self.buf += bytes(pad + lead)
        off = len(self.buf)
        self.buf += arr.tobytes()
        return off

    def finish(self):
        pad = (-len(self.buf)) % 16
        return np.frombuffer(bytes(self.buf) + bytes(pad + F.ATLAS_SLACK), np.uint8).copy()


def _premul(rng, opaque=None):
    if opaque is None and rng.random() < 0.08:
        return 0xff000000           # opaque black: the usual outline, a variant of its own
    a = 255 if opaque else int(rng.integers(1, 256))
    if opaque is None and rng.random() < 0.3:
        a = 255
    c = [int(rng.integers(0, 256)) * a // 255 for _ in range(3)]
    return (a << 24) | (c[0] << 16) | (c[1] << 8) | c[2]


def _blob_mask(rng, w, h):
    """glyph-like A8 mask: hard interior, soft edge, empty exterior."""
    yy, xx = np.mgrid[0:h, 0:w]
    cx, cy = rng.uniform(0.3, 0.7) * w, rng.uniform(0.3, 0.7) * h
    rx, ry = rng.uniform(0.2, 0.5) * w + 1, rng.uniform(0.2, 0.5) * h + 1
    dist = np.sqrt(((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2)
    m = np.clip((1.0 - dist) * rng.uniform(2, 12), 0, 1)
    if rng.random() < 0.5:   # ring (outline-like)
        m = np.clip(m - np.clip((0.6 - dist) * 8, 0, 1), 0, 1)
    m = (m * 255 + 0.5).astype(np.uint8)
    if rng.random() < 0.3:
        noise = rng.integers(0, 32, m.shape, dtype=np.uint8)
        m = np.where(m > 0, m ^ noise, m).astype(np.uint8)
    return m


def _clip(x0, y0, x1, y1, bx):
    return (max(x0, bx[0]), max(y0, bx[1]), min(x1, bx[2]), min(y1, bx[3]))


def random_scene(seed, width, height, n_regions=6, prims_per_region=24,
                 allow_groups=True, fixed_count=False):
    """Returns (regions, prims, atlas) numpy arrays."""
    rng = np.random.default_rng(seed)
    atlas = AtlasBuilder()
    prims, regions = [], []

    def add(op, rule, bbox, color=0, a=0, b=0, c=0, d=0, param=0):
        x0, y0, x1, y1 = bbox
        if x1 <= x0 or y1 <= y0:
            return False
        prims.append((op, rule, param, x0, y0, x1, y1, color & 0xffffffff, a, b, c, d))
        return True

    for _ in range(n_regions):
        rw = int(rng.integers(8, max(9, width)))
        rh = int(rng.integers(4, max(5, height // 2 + 1)))
        rx = int(rng.integers(-rw // 4, max(1, width - rw // 2)))
        ry = int(rng.integers(-rh // 4, max(1, height - rh // 2)))
        box = _clip(rx, ry, rx + rw, ry + rh, (0, 0, width, height))
        if box[2] <= box[0] or box[3] <= box[1]:
            continue
        first = len(prims)
        grouped = allow_groups and rng.random() < 0.4
        if grouped:
            add(F.OP_GROUP_BEGIN, 0, box)
        # background box (aligned)
        if rng.random() < 0.8:
            col = _premul(rng)
            add(F.OP_RECT, F.RULE_PIXMAN, box, col, box[0] * 256, box[1] * 256,
                box[2] * 256, box[3] * 256)
        # inner clip
        pad = int(rng.integers(0, 4))
        clip = _clip(box[0] + pad, box[1] + pad, box[2] - pad, box[3] - pad, box)
        for _ in range(prims_per_region if fixed_count else
                       int(rng.integers(1, prims_per_region + 1))):
            kind = rng.random()
            if clip[2] <= clip[0] or clip[3] <= clip[1]:
                break
            if kind < 0.25:   # unaligned rect
                fx1 = int(rng.integers(clip[0] * 256 - 300, clip[2] * 256))
                fy1 = int(rng.integers(clip[1] * 256 - 300, clip[3] * 256))
                fx2 = fx1 + int(rng.integers(1, 40 * 256))
                fy2 = fy1 + int(rng.integers(1, 6 * 256))
                if rng.random() < 0.25:     # a box with fractional edges over many tiles
                    fx2 = fx1 + int(rng.integers(1, max(2, (clip[2] - clip[0]) * 256)))
                    fy2 = fy1 + int(rng.integers(1, max(2, (clip[3] - clip[1]) * 256)))
                if rng.random() < 0.3:
                    fx1 &= ~255; fx2 &= ~255
                col = _premul(rng)
                aligned = ((fx1 | fy1 | fx2 | fy2) & 255) == 0
                rule = F.RULE_LERP if (col >> 24) == 255 and not aligned else F.RULE_PIXMAN
                if not aligned and rng.random() < 0.15:
                    rule = F.RULE_LERP      # cairo: any colour onto a surface that is still clear
                bb = _clip(fx1 >> 8, fy1 >> 8, (fx2 + 255) >> 8, (fy2 + 255) >> 8, clip)
                add(F.OP_RECT, rule, bb, col, fx1, fy1, fx2, fy2)
            elif kind < 0.85:  # glyph / outline mask
                big = rng.random() < 0.15
                mw = int(rng.integers(3, 200 if big else 40))
                mh = int(rng.integers(3, 60 if big else 40))
                m = _blob_mask(rng, mw, mh)
                # atlas contract: >= 3 zero bytes before the first row and after
                # every row, unless the prim carries MASK_XCLIP
                padded = rng.random() < 0.7
                stride = mw + (int(rng.integers(3, 7)) if padded else int(rng.integers(0, 3)))
                buf = np.zeros((mh, stride), np.uint8)
                buf[:, :mw] = m
                off = atlas.put(buf, lead=int(rng.integers(3, 7)) if padded else int(rng.integers(0, 3)))
                ox = int(rng.integers(clip[0] - mw // 2, clip[2]))
                oy = int(rng.integers(clip[1] - mh // 2, clip[3]))
                bb = _clip(ox, oy, ox + mw, oy + mh, clip)
                col = _premul(rng)
                rule = F.RULE_LERP if (col >> 24) == 255 and rng.random() < 0.5 else F.RULE_PIXMAN
                xclip = (not padded) or bb[0] > ox or bb[2] < ox + mw or rng.random() < 0.2
                add(F.OP_MASK, rule, bb, col, off, stride, ox, oy, param=1 if xclip else 0)
            else:              # ARGB image
                iw, ih = int(rng.integers(2, 50)), int(rng.integers(2, 20))
                img = np.array([[_premul(rng) if rng.random() < 0.8 else 0
                                 for _ in range(iw)] for _ in range(ih)], np.uint32)
                off = atlas.put(img)
                ox = int(rng.integers(clip[0] - iw // 2, clip[2]))
                oy = int(rng.integers(clip[1] - ih // 2, clip[3]))
                bb = _clip(ox, oy, ox + iw, oy + ih, clip)
                add(F.OP_IMAGE, F.RULE_PIXMAN, bb, 0, off, iw * 4, ox, oy)
        if grouped:
            add(F.OP_GROUP_END, 0, box, param=int(rng.integers(0, 256)))
        regions.append((box[0], box[1], box[2], box[3], first, len(prims) - first))

    R = np.array(regions, F.REGION_DTYPE) if regions else np.zeros(0, F.REGION_DTYPE)
    P = np.array(prims, F.PRIM_DTYPE) if prims else np.zeros(0, F.PRIM_DTYPE)
    return R, P, atlas.finish()


def subtitle_like_scene(seed, width, height, n_regions=8, font_px=None,
                        outline_px=3, lines=2, opacity_every=4):
    """Synthetic stand-in for BASELINE config 2/4 shaped scenes when no
    reference-derived scene pack is available: n_regions stacked region boxes
    (80% x ~10% of the frame) with a translucent background, `lines` lines of
    glyph-sized A8 masks and, if outline_px, one outline mask per line (LERP
    rule, opaque black), every `opacity_every`-th region in an opacity group."""
    rng = np.random.default_rng(seed)
    font_px = font_px or max(8, height // 22)
    atlas = AtlasBuilder()
    # a small alphabet of glyph masks shared by all instances
    gw = max(3, int(font_px * 0.6))
    alphabet = []
    for _ in range(40):
        w = int(rng.integers(max(2, gw // 2), gw + 1))
        m = np.zeros((font_px, w + 4), np.uint8)
        m[:, :w] = _blob_mask(rng, w, font_px)
        alphabet.append((atlas.put(m, lead=4), w))
    prims, regions = [], []
    rw = int(width * 0.8) // 4 * 4
    rh = min(int(lines * font_px * 1.15) + 8, height // max(1, n_regions))
    for r in range(n_regions):
        x0 = (width - rw) // 2
        y0 = int((r + 0.5) * height / n_regions - rh / 2)
        box = _clip(x0, y0, x0 + rw, y0 + rh, (0, 0, width, height))
        first = len(prims)
        grouped = opacity_every and (r % opacity_every == opacity_every - 1)
        if grouped:
            prims.append((F.OP_GROUP_BEGIN, 0, 0, *box, 0, 0, 0, 0, 0))
        a = int(rng.integers(0x99, 0xE1))
        col = (a << 24) | (int(rng.integers(0, 80)) * a // 255 << 16) | \
            (int(rng.integers(0, 80)) * a // 255 << 8) | (int(rng.integers(0, 120)) * a // 255)
        prims.append((F.OP_RECT, F.RULE_PIXMAN, 0, *box, col, box[0] * 256,
                      box[1] * 256, box[2] * 256, box[3] * 256))
        text_col = 0xff000000 | int(rng.integers(0x808080, 0xffffff))
        for ln in range(lines):
            ly = box[1] + 4 + int(ln * font_px * 1.15)
            n_gl = int(rng.integers(24, 49))
            xs, x = [], box[0] + int(rng.integers(4, 40))
            for _ in range(n_gl):
                off, w = alphabet[int(rng.integers(0, len(alphabet)))]
                if x + w + 2 >= box[2]:
                    break
                xs.append((x, off, w))
                x += w + int(rng.integers(1, 4))
            if outline_px and xs:
                ox0, ox1 = xs[0][0] - outline_px, xs[-1][0] + xs[-1][2] + outline_px
                ow, oh = ox1 - ox0, font_px + 2 * outline_px
                om = np.zeros((oh, ow), np.uint8)
                for (gx, off, w) in xs:
                    g = np.frombuffer(bytes(atlas.buf[off:off + (w + 4) * font_px]),
                                      np.uint8).reshape(font_px, w + 4)[:, :w]
                    for dy in range(0, 2 * outline_px + 1, max(1, outline_px)):
                        for dx in range(0, 2 * outline_px + 1, max(1, outline_px)):
                            sub = om[dy:dy + font_px, gx - ox0 - outline_px + dx:
                                     gx - ox0 - outline_px + dx + w]
                            np.maximum(sub, g[:sub.shape[0], :sub.shape[1]], out=sub)
                omp = np.zeros((oh, ow + 4), np.uint8)
                omp[:, :ow] = om
                ooff = atlas.put(omp, lead=4)
                bb = _clip(ox0, ly - outline_px, ox1, ly - outline_px + oh, box)
                if bb[2] > bb[0] and bb[3] > bb[1]:
                    xc = 1 if (bb[0] > ox0 or bb[2] < ox1) else 0
                    prims.append((F.OP_MASK, F.RULE_LERP, xc, *bb, 0xff000000, ooff, ow + 4,
                                  ox0, ly - outline_px))
            for (gx, off, w) in xs:
                bb = _clip(gx, ly, gx + w, ly + font_px, box)
                if bb[2] > bb[0] and bb[3] > bb[1]:
                    xc = 1 if (bb[0] > gx or bb[2] < gx + w) else 0
                    prims.append((F.OP_MASK, F.RULE_PIXMAN, xc, *bb, text_col, off, w + 4, gx, ly))
        if grouped:
            prims.append((F.OP_GROUP_END, 0, int(rng.integers(100, 200)), *box, 0, 0, 0, 0, 0))
        regions.append((*box, first, len(prims) - first))
    return (np.array(regions, F.REGION_DTYPE), np.array(prims, F.PRIM_DTYPE),
            atlas.finish())


def algorithmic_bytes(width, height, regions, prims):
    """4*W*H + sum of A8 mask bytes sampled + 4 * image bytes
    + display-list bytes.  Per frame."""
    m = prims[prims["op"] == F.OP_MASK]
    im = prims[prims["op"] == F.OP_IMAGE]
    area = lambda p: int(((p["cx1"].astype(np.int64) - p["cx0"]) *
                          (p["cy1"].astype(np.int64) - p["cy0"])).sum())
    return (4 * width * height + area(m) + 4 * area(im) +
            prims.nbytes + regions.nbytes)
