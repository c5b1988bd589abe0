"""Turns the real-cairo/pixman known-answer tables (tests/golden/kats.npz,
made by oracle/kat_gen.c) into display lists, so that the SAME answers can be
demanded from the CPU oracle's compositor and from the HIP kernel.

The destination values of a KAT record are first established with an IMAGE
primitive (x OVER transparent == x exactly), then the primitive under test is
drawn on top."""
import numpy as np

import oracle_lib as O
import pyflutl as F
from scenes import AtlasBuilder


def _finish(regions, prims, atlas):
    return (np.array(regions, F.REGION_DTYPE), np.array(prims, F.PRIM_DTYPE),
            atlas.finish())


def mask_scene(kats, max_rows=None):
    """KAT 'mask': rows of 64 px, one colour per row."""
    t = kats["mask"].reshape(-1, 64, 4)
    if max_rows:
        t = t[:max_rows]
    n = len(t)
    W, H = 64, n
    atlas = AtlasBuilder()
    dst_off = atlas.put(np.ascontiguousarray(t[:, :, 2]))
    m_off = atlas.put(t[:, :, 1].astype(np.uint8))
    prims = [(F.OP_IMAGE, 0, 0, 0, 0, W, H, 0, dst_off, W * 4, 0, 0)]
    cols = O.color_rgba8888(t[:, 0, 0])
    for r in range(n):
        prims.append((F.OP_MASK, F.RULE_PIXMAN, 0, 0, r, W, r + 1, int(cols[r]),
                      m_off + r * 64, 64, 0, r))
    regions = [(0, 0, W, H, 0, len(prims))]
    return W, H, _finish(regions, prims, atlas), np.ascontiguousarray(t[:, :, 3])


def fill_scene(kats):
    """KAT 'fill': 1x1 aligned solid fills; laid out 256 per row."""
    t = kats["fill"]
    n = (len(t) // 256) * 256
    t = t[:n]
    W, H = 256, n // 256
    atlas = AtlasBuilder()
    dst_off = atlas.put(np.ascontiguousarray(t[:, 1]))
    prims = [(F.OP_IMAGE, 0, 0, 0, 0, W, H, 0, dst_off, W * 4, 0, 0)]
    cols = O.color_rgba8888(t[:, 0])
    for i in range(n):
        x, y = i % W, i // W
        prims.append((F.OP_RECT, F.RULE_PIXMAN, 0, x, y, x + 1, y + 1, int(cols[i]),
                      x * 256, y * 256, (x + 1) * 256, (y + 1) * 256))
    # many prims: split into regions of one row each to exercise region culling
    regions = [(0, 0, W, H, 0, 1)]
    for y in range(H):
        regions.append((0, y, W, y + 1, 1 + y * W, W))
    return W, H, _finish(regions, prims, atlas), t[:, 2].reshape(H, W)


def group_scene(kats, max_rows=None):
    """KAT 'group': rows of 64 px painted through a region opacity group."""
    g = kats["group"].reshape(-1, 64, 3)
    op = kats["group_opacity"].reshape(-1, 64)[:, 0]
    if max_rows:
        g, op = g[:max_rows], op[:max_rows]
    n = len(g)
    W, H = 64, n
    atlas = AtlasBuilder()
    dst_off = atlas.put(np.ascontiguousarray(g[:, :, 1]))
    grp_off = atlas.put(np.ascontiguousarray(g[:, :, 0]))
    prims = [(F.OP_IMAGE, 0, 0, 0, 0, W, H, 0, dst_off, W * 4, 0, 0)]
    regions = [(0, 0, W, H, 0, 1)]
    for r in range(n):
        alpha = O.lib().flutl_oracle_opacity_to_alpha(float(op[r]))
        first = len(prims)
        box = (0, r, W, r + 1)
        prims.append((F.OP_GROUP_BEGIN, 0, 0, *box, 0, 0, 0, 0, 0))
        prims.append((F.OP_IMAGE, 0, 0, *box, 0, grp_off + r * W * 4, W * 4, 0, r))
        prims.append((F.OP_GROUP_END, 0, alpha, *box, 0, 0, 0, 0, 0))
        regions.append((*box, first, 3))
    return W, H, _finish(regions, prims, atlas), np.ascontiguousarray(g[:, :, 2])


def rect_scene(kats):
    """KAT 'rect': 12x12 patches with one (possibly unaligned) box each,
    tiled 16 per row."""
    P, C = kats["rect_params"], kats["rect_color"]
    D, E = kats["rect_dst"], kats["rect_out"]
    n = len(P)
    cols_per_row = 16
    rows = (n + cols_per_row - 1) // cols_per_row
    W, H = 12 * cols_per_row, 12 * rows
    dst = np.zeros((H, W), np.uint32)
    exp = np.zeros((H, W), np.uint32)
    atlas = AtlasBuilder()
    prims = [None]
    regions = [(0, 0, W, H, 0, 1)]
    fx = O.lib().flutl_oracle_fixed_from_double
    for i in range(n):
        px, py = (i % cols_per_row) * 12, (i // cols_per_row) * 12
        dst[py:py + 12, px:px + 12] = D[i]
        exp[py:py + 12, px:px + 12] = E[i]
        tx, ty, x, y, w, h = P[i]
        # cairo path construction: move_to -> fixed(device x); rel_line_to adds
        # fixed(w) to the fixed current point
        x1 = fx(tx + x); y1 = fx(ty + y); x2 = x1 + fx(w); y2 = y1 + fx(h)
        col = int(O.lib().flutl_oracle_color_rgba8888(int(C[i])))
        aligned = ((x1 | y1 | x2 | y2) & 255) == 0
        rule = F.RULE_LERP if (col >> 24) == 255 and not aligned else F.RULE_PIXMAN
        X1, Y1, X2, Y2 = x1 + px * 256, y1 + py * 256, x2 + px * 256, y2 + py * 256
        bb = (max(px, X1 >> 8), max(py, Y1 >> 8), min(px + 12, (X2 + 255) >> 8),
              min(py + 12, (Y2 + 255) >> 8))
        if bb[2] <= bb[0] or bb[3] <= bb[1] or (col >> 24) == 0:
            continue
        prims.append((F.OP_RECT, rule, 0, *bb, col, X1, Y1, X2, Y2))
        regions.append((*bb, len(prims) - 1, 1))
    dst_off = atlas.put(dst)
    prims[0] = (F.OP_IMAGE, 0, 0, 0, 0, W, H, 0, dst_off, W * 4, 0, 0)
    return W, H, _finish(regions, prims, atlas), exp


def stroke_scene(kats):
    """KAT 'stroke': the coverage mask cairo produced for a stroked path,
    composited with the outline colour (LERP rule when opaque)."""
    C, M = kats["stroke_color"], kats["stroke_mask"]
    D, E = kats["stroke_dst"], kats["stroke_out"]
    n, mh, mw = M.shape
    W, H = mw, mh * n
    atlas = AtlasBuilder()
    dst_off = atlas.put(np.ascontiguousarray(D.reshape(H, W)))
    m_off = atlas.put(np.ascontiguousarray(M.reshape(H, W)))
    prims = [(F.OP_IMAGE, 0, 0, 0, 0, W, H, 0, dst_off, W * 4, 0, 0)]
    for i in range(n):
        col = int(O.lib().flutl_oracle_color_rgba8888(int(C[i])))
        rule = F.RULE_LERP if (col >> 24) == 255 else F.RULE_PIXMAN
        prims.append((F.OP_MASK, rule, 0, 0, i * mh, W, (i + 1) * mh, col,
                      m_off + i * mh * mw, mw, 0, i * mh))
    regions = [(0, 0, W, H, 0, len(prims))]
    return W, H, _finish(regions, prims, atlas), E.reshape(H, W)


def image_scene(kats):
    t = kats["image"].reshape(-1, 64, 3)
    n = len(t)
    W, H = 64, n
    atlas = AtlasBuilder()
    dst_off = atlas.put(np.ascontiguousarray(t[:, :, 1]))
    src_off = atlas.put(np.ascontiguousarray(t[:, :, 0]))
    prims = [(F.OP_IMAGE, 0, 0, 0, 0, W, H, 0, dst_off, W * 4, 0, 0),
             (F.OP_IMAGE, 0, 0, 0, 0, W, H, 0, src_off, W * 4, 0, 0)]
    regions = [(0, 0, W, H, 0, 2)]
    return W, H, _finish(regions, prims, atlas), np.ascontiguousarray(t[:, :, 2])


ALL = {"mask": mask_scene, "fill": fill_scene, "group": group_scene,
       "rect": rect_scene, "stroke": stroke_scene, "image": image_scene}
