"""CPU: holds oracle/flutl_oracle.c to the known answers of the REAL
cairo 1.16.0 / pixman 0.40.0 (tests/golden/kats.npz from oracle/kat_gen.c).
This is what pins the oracle's arithmetic; scene-level pinning against the
reference plugin's own frames is in test_oracle_scenes.py."""
import os

import numpy as np
import pytest

import kat_scenes
import oracle_lib as O
import pyflutl as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_versions(kats):
    assert str(kats["versions"]) == "cairo 1.16.0 pixman 0.40.0"


def test_color_quantisation(kats):
    t = kats["color"]
    assert np.array_equal(O.color_rgba8888(t[:, 0]), t[:, 1])


def test_solid_over(kats):
    t = kats["fill"]
    assert np.array_equal(O.over(O.color_rgba8888(t[:, 0]), t[:, 1]), t[:, 2])


def test_solid_in_mask_over(kats):
    t = kats["mask"]
    got = O.blend(F.RULE_PIXMAN, O.color_rgba8888(t[:, 0]), t[:, 1], t[:, 2])
    assert np.array_equal(got, t[:, 3])


def test_group_paint_with_alpha(kats):
    t, op = kats["group"], kats["group_opacity"]
    al = np.array([O.lib().flutl_oracle_opacity_to_alpha(float(x)) for x in op],
                  np.uint32)
    assert np.array_equal(O.group(t[:, 0], al, t[:, 1]), t[:, 2])


def test_image_over(kats):
    t = kats["image"]
    assert np.array_equal(O.over(t[:, 0], t[:, 1]), t[:, 2])


def test_blur_kernel_and_convolution(kats):
    for i in range(int(kats["blur_count"])):
        r, sigma = kats[f"blur{i}_radius_sigma"]
        r = int(r)
        k = kats[f"blur{i}_kernel"]
        assert np.array_equal(O.blur_kernel(r, sigma), k)
        assert np.array_equal(F.blur_kernel(r, sigma), k)   # product host helper
        assert np.array_equal(O.blur(kats[f"blur{i}_src"], k, r), kats[f"blur{i}_out"])


@pytest.mark.parametrize("name", sorted(kat_scenes.ALL))
def test_compositor_reproduces_kat(kats, name):
    """The oracle's display-list compositor (not just its scalar helpers)
    reproduces real cairo's output for every table, incl. the unaligned-box
    coverage rule and the opaque-stroke lerp rule."""
    W, H, (R, P, A), exp = kat_scenes.ALL[name](kats)
    assert F.validate(W, H, R, P, A.size) == 0
    got = O.render(W, H, R, P, A)
    assert np.array_equal(got, exp), f"{name}: {(got != exp).sum()} px differ"


def test_fractional_blit_is_resample_then_over(kats):
    """KAT table 9 (real cairo: surface painted / filled through its extent under a
    fractional translation) == oracle resample (pixman bilinear with constant weights,
    IN the box's edge coverage) + integer blit with OVER."""
    T, Of = kats["fracblit_t"], kats["fracblit_o"]
    for i in range(len(T)):
        src, dst, exp = kats["fracblit_src"][i], kats["fracblit_dst"][i], kats["fracblit_out"][i]
        X0, Y0 = T[i][0] + Of[i][0], T[i][1] + Of[i][1]
        bx, dx = O.frac_origin(X0)
        by, dy = O.frac_origin(Y0)
        box = None
        if i % 2 == 0:      # even records: cairo_rectangle + cairo_fill (the reference's way)
            sh, sw = src.shape
            fx1, fy1 = O.fixed_24_8(X0) - bx * 256, O.fixed_24_8(Y0) - by * 256
            box = [fx1, fy1, fx1 + sw * 256, fy1 + sh * 256]
        R = O.resample(src, dx, dy, box)
        H, W = dst.shape
        layer = np.zeros_like(dst)
        for j in range(R.shape[0]):
            for k in range(R.shape[1]):
                y, x = by + j, bx + k
                if 0 <= y < H and 0 <= x < W:
                    layer[y, x] = R[j, k]
        got = O.over(layer.reshape(-1), dst.reshape(-1)).reshape(H, W)
        assert np.array_equal(got, exp), i


def _polygons():
    z = np.load(os.path.join(ROOT, "tests", "golden", "polygons.npz"))
    for i, (kind, x0, y0, x1, y1, eo) in enumerate(z["meta"]):
        yield i, int(kind), (int(x0), int(y0), int(x1), int(y1)), bool(eo), z[f"edges{i}"], z[f"cov{i}"]


def test_tor_restatement_reproduces_cairos_coverage():
    """oracle/flutl_tor_oracle.c (Cairo 1.16's polygon scan converter restated) against
    the real libcairo: 321 committed polygons -- random fills on coarse grids (equal sort
    keys all over), random strokes and strokes of DejaVu text as the product's stroker
    makes them, inside the surface and cut by a clip box (oracle/polygon_kats.c)."""
    n = 0
    for i, kind, box, eo, edges, cov in _polygons():
        got = O.scan_polygon(edges, box, eo)
        assert np.array_equal(got, cov), f"polygon {i} (kind {kind}): {(got != cov).sum()} px differ"
        n += 1
    assert n >= 300
