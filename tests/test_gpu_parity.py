"""GPU (MI355X): bit-exact parity of the HIP compositor, called through the
C-ABI, against (a) the known answers of real cairo/pixman, (b) the CPU oracle
on seeded adversarial display lists, (c) size-independent properties at the
BASELINE frame sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import kat_scenes
import oracle_lib as O
import pyflutl as F
import scenes


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    assert F.lib().flutl_device_count() >= 1
    import gpu_util
    return gpu_util


@pytest.mark.parametrize("name", sorted(kat_scenes.ALL))
def test_kernel_reproduces_real_cairo_kats(gpu, kats, name):
    W, H, (R, P, A), exp = kat_scenes.ALL[name](kats)
    got = gpu.hip_render(W, H, R, P, A)
    assert np.array_equal(got, exp), f"{name}: {(got != exp).sum()} px differ"


SIZES = [(320, 180), (1, 1), (3, 2), (64, 8), (65, 9), (63, 7), (130, 50),
         (257, 33), (1280, 720)]


@pytest.mark.parametrize("size", SIZES)
@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_kernel_matches_oracle_on_random_lists(gpu, size, seed):
    W, H = size
    R, P, A = scenes.random_scene(seed * 100 + W, W, H, n_regions=8)
    exp = O.render(W, H, R, P, A)
    got = gpu.hip_render(W, H, R, P, A)
    assert np.array_equal(got, exp), f"{(got != exp).sum()} px differ"


def test_many_prims_per_region_spans_binning_chunks(gpu):
    """> 128 prims in one region: the kernel bins in several rounds."""
    W, H = 320, 96
    R, P, A = scenes.random_scene(77, W, H, n_regions=3, prims_per_region=600,
                                  fixed_count=True)
    assert max(R["count"]) > 256
    assert np.array_equal(gpu.hip_render(W, H, R, P, A), O.render(W, H, R, P, A))


def test_group_repaints_a_strip_whose_frame_pixels_are_parked(gpu):
    """Frame pixels (a mask of a lower region), then an overlapping opacity group
    that first puts coverage on the same strips and then covers them with an
    opaque box: the strips are one colour again, but their LDS slots no longer
    hold the frame's pixels (found by the random lists once they had big boxes)."""
    W, H = 192, 48
    atlas = scenes.AtlasBuilder()
    rng = np.random.default_rng(5)
    m = (rng.integers(0, 256, (40, 150)) * (rng.random((40, 150)) < 0.7)).astype(np.uint8)
    buf = np.zeros((40, 154), np.uint8); buf[:, :150] = m
    off = atlas.put(buf, lead=4)
    prims = [
        # region 0 (no group): background box + a big mask -> frame strips in LDS
        (F.OP_RECT, F.RULE_PIXMAN, 0, 0, 0, W, H, 0x80402010, 0, 0, W * 256, H * 256),
        (F.OP_MASK, F.RULE_PIXMAN, 0, 10, 4, 160, 44, 0xc0a05030, off, 154, 10, 4),
        # region 1: a group over it
        (F.OP_GROUP_BEGIN, 0, 0, 0, 0, W, H, 0, 0, 0, 0, 0),
        (F.OP_RECT, F.RULE_PIXMAN, 0, 0, 0, W, H, 0x60102030, 0, 0, W * 256, H * 256),
        (F.OP_MASK, F.RULE_PIXMAN, 0, 20, 2, 170, 42, 0xff204060, off, 154, 20, 2),
        (F.OP_RECT, F.RULE_PIXMAN, 0, 0, 8, W, 40, 0xff807060, 0, 8 * 256, W * 256, 40 * 256),
        (F.OP_MASK, F.RULE_LERP, 0, 30, 20, 180, 48, 0xff000000, off, 154, 30, 20),
        (F.OP_GROUP_END, 0, 150, 0, 0, W, H, 0, 0, 0, 0, 0),
    ]
    P = np.array(prims, F.PRIM_DTYPE)
    R = np.array([(0, 0, W, H, 0, 2), (0, 0, W, H, 2, 6)], F.REGION_DTYPE)
    A = atlas.finish()
    assert np.array_equal(gpu.hip_render(W, H, R, P, A), O.render(W, H, R, P, A))


def test_empty_list_clears_the_frame(gpu):
    R = np.zeros(0, F.REGION_DTYPE); P = np.zeros(0, F.PRIM_DTYPE)
    A = np.zeros(64, np.uint8)
    got = gpu.hip_render(200, 40, R, P, A)
    assert (got == 0).all()


def test_row_padding_is_not_touched(gpu):
    W, H = 100, 20
    R, P, A = scenes.random_scene(5, W, H)
    got = gpu.hip_render(W, H, R, P, A, stride=W * 4 + 64)
    assert np.array_equal(got, O.render(W, H, R, P, A))


def test_context_api_render_to_host(gpu):
    W, H = 320, 180
    R, P, A = scenes.random_scene(11, W, H)
    exp = O.render(W, H, R, P, A)
    with F.Context(0) as ctx:
        # upload the atlas entry by entry is the element's job; here one blob
        off = ctx.atlas_put(A[F.ATLAS_SLACK:])
        assert off == F.ATLAS_SLACK          # same offsets as the numpy atlas
        lst = ctx.list_create(W, H, R, P)
        got = ctx.render_to_host(lst, W, H)
        got2 = ctx.render_to_host(lst, W, H)
        ctx.list_destroy(lst)
    assert np.array_equal(got, exp) and np.array_equal(got2, exp)


def test_context_rejects_invalid_list(gpu):
    R, P, A = scenes.random_scene(11, 320, 180)
    with F.Context(0) as ctx:
        with pytest.raises(F.FlutlError) as e:   # atlas empty -> reads out of range
            ctx.list_create(320, 180, R, P)
        assert e.value.code == -34


def _concat_lists(lists, W, H):
    """Several display lists -> one region / prim array + atlas + frame descriptors."""
    regs, prims, frames = [], [], []
    atlas_all = np.zeros(F.ATLAS_SLACK, np.uint8)
    for i, (R, P, A) in enumerate(lists):
        R, P = R.copy(), P.copy()
        shift = atlas_all.size - F.ATLAS_SLACK
        isref = (P["op"] == F.OP_MASK) | (P["op"] == F.OP_IMAGE)
        P["a"][isref] += shift
        R["first"] += sum(len(p) for p in prims)
        frames.append((sum(len(r) for r in regs), len(R), i * W * H * 4))
        regs.append(R); prims.append(P)
        atlas_all = np.concatenate([atlas_all, A[F.ATLAS_SLACK:]])
    return (np.concatenate(regs), np.concatenate(prims), atlas_all,
            np.array(frames, F.FRAME_DTYPE))


# 5 frames: rows interleaved over the XCDs; 8: whole frames per XCD; 19: both launches
@pytest.mark.parametrize("n_frames", [5, 8, 19])
def test_batch_launch_equals_single_launches(gpu, n_frames):
    import torch
    W, H = 256, 64
    lists = [scenes.random_scene(s, W, H, n_regions=4) for s in range(n_frames)]
    R, P, atlas_all, Fd = _concat_lists(lists, W, H)
    dst = torch.full((n_frames * W * H * 4,), 0x5A, dtype=torch.uint8, device="cuda")
    F.launch_batch(torch.cuda.current_stream().cuda_stream, W, H, dst,
                   gpu.to_dev(Fd), n_frames, gpu.to_dev(R), gpu.to_dev(P),
                   gpu.to_dev(atlas_all))
    torch.cuda.synchronize()
    out = dst.cpu().numpy().view(np.uint32).reshape(n_frames, H, W)
    for i, (Ri, Pi, Ai) in enumerate(lists):
        assert np.array_equal(out[i], O.render(W, H, Ri, Pi, Ai)), i


@pytest.mark.parametrize("size,n_frames,seed0", [((200, 72), 40, 1000), ((97, 41), 27, 2000),
                                                 ((512, 96), 12, 3000)])
def test_fuzz_batches_against_oracle(gpu, size, n_frames, seed0):
    """Many adversarial lists (overlapping regions, opacity groups over painted
    pixels -> the park path, images, clipped masks) in one launch, every frame
    against the oracle."""
    import torch
    W, H = size
    lists = [scenes.random_scene(seed0 + s, W, H, n_regions=3 + s % 10,
                                 prims_per_region=8 + (s * 7) % 40)
             for s in range(n_frames)]
    R, P, atlas_all, Fd = _concat_lists(lists, W, H)
    assert F.validate(W, H, R, P, atlas_all.size) == 0
    dst = torch.full((n_frames * W * H * 4,), 0xC3, dtype=torch.uint8, device="cuda")
    F.launch_batch(torch.cuda.current_stream().cuda_stream, W, H, dst,
                   gpu.to_dev(Fd), n_frames, gpu.to_dev(R), gpu.to_dev(P),
                   gpu.to_dev(atlas_all))
    torch.cuda.synchronize()
    out = dst.cpu().numpy().view(np.uint32).reshape(n_frames, H, W)
    for i, (Ri, Pi, Ai) in enumerate(lists):
        exp = O.render(W, H, Ri, Pi, Ai)
        assert np.array_equal(out[i], exp), (i, int((out[i] != exp).sum()))


@pytest.mark.parametrize("size,n_frames", [((256, 64), 3), ((130, 50), 9), ((63, 7), 2),
                                           ((1280, 720), 2)])
def test_fused_over_video_equals_overlay_then_pixman_over(gpu, size, n_frames):
    """flutl_launch_batch_over == (overlay from the oracle) OVER video, pixman rule
    (the OVER itself is pinned by kats.image)."""
    import torch
    W, H = size
    lists = [scenes.random_scene(900 + s, W, H, n_regions=5) for s in range(n_frames)]
    R, P, atlas_all, Fd = _concat_lists(lists, W, H)
    rng = np.random.default_rng(W * 7 + n_frames)
    video = rng.integers(0, 1 << 32, (n_frames, H, W), dtype=np.uint64).astype(np.uint32)
    video[0] |= 0xFF000000                       # frame 0: opaque video, the usual case
    vd = gpu.to_dev(video)
    dst = torch.full((n_frames * W * H * 4,), 0x5A, dtype=torch.uint8, device="cuda")
    F.launch_batch_over(torch.cuda.current_stream().cuda_stream, W, H, dst, vd,
                        gpu.to_dev(Fd), n_frames, gpu.to_dev(R), gpu.to_dev(P),
                        gpu.to_dev(atlas_all))
    torch.cuda.synchronize()
    out = dst.cpu().numpy().view(np.uint32).reshape(n_frames, H, W)
    for i, (Ri, Pi, Ai) in enumerate(lists):
        overlay = O.render(W, H, Ri, Pi, Ai)
        exp = O.over(overlay.reshape(-1), video[i].reshape(-1)).reshape(H, W)
        assert np.array_equal(out[i], exp), (i, int((out[i] != exp).sum()))
    assert np.array_equal(vd.cpu().numpy().view(np.uint32).reshape(video.shape), video), \
        "the video frames were written to"


def test_blur_matches_pixman_kats(gpu, kats):
    import torch
    for i in range(int(kats["blur_count"])):
        r, sigma = kats[f"blur{i}_radius_sigma"]; r = int(r)
        src = kats[f"blur{i}_src"]; h, w = src.shape
        k = F.blur_kernel(r, sigma)
        d = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
        F.launch_blur(torch.cuda.current_stream().cuda_stream, gpu.to_dev(src), d,
                      w, h, gpu.to_dev(k), r)
        torch.cuda.synchronize()
        got = d.cpu().numpy().view(np.uint32).reshape(h, w)
        exp = kats[f"blur{i}_out"]
        bad = np.argwhere(got != exp)
        assert len(bad) == 0, (i, r, len(bad), [(int(y), int(x), hex(got[y, x]), hex(exp[y, x]))
                                             for y, x in bad[:6]])


@pytest.mark.parametrize("size", [(10, 7), (64, 4), (65, 5), (301, 57)])
def test_resample_matches_oracle_and_cairo_kats(gpu, kats, size):
    """flutl_launch_resample == oracle (pinned by the real-cairo fractional blits)."""
    import torch
    w, h = size
    rng = np.random.default_rng(w * 31 + h)
    for distx, disty in [(64, 0), (0, 32), (1, 127), (96, 17)]:
        if (w, h) == (10, 7):       # the KAT images themselves
            src = kats["fracblit_src"][(distx + disty) % len(kats["fracblit_src"])]
        else:
            src = rng.integers(0, 1 << 32, (h, w), dtype=np.uint64).astype(np.uint32)
        sd = gpu.to_dev(src)
        dd = torch.full(((h + 1) * (w + 1) * 4,), 0x77, dtype=torch.uint8, device="cuda")
        for box in (None, [256 - 2 * distx, 256 - 2 * disty,
                           256 - 2 * distx + w * 256, 256 - 2 * disty + h * 256]):
            F.launch_resample(torch.cuda.current_stream().cuda_stream, sd, dd, w, h,
                              distx, disty, box)
            torch.cuda.synchronize()
            got = dd.cpu().numpy().view(np.uint32).reshape(h + 1, w + 1)
            assert np.array_equal(got, O.resample(src, distx, disty, box)), (distx, disty, box)


def test_blur_matches_oracle_large(gpu):
    import torch
    rng = np.random.default_rng(3)
    h, w = 150, 333
    a = rng.integers(0, 256, (h, w)).astype(np.uint32)
    src = (a << 24) | ((a * rng.integers(0, 256, (h, w)) // 255) << 16) | (a >> 1)
    src = src.astype(np.uint32)
    # radius 16 is the last LDS-tiled one; 17, 40 take the wide kernel
    for r, sigma in [(2, 1.0), (6, 3.0), (16, 8.0), (17, 8.5), (40, 20.0)]:
        k = F.blur_kernel(r, sigma)
        assert np.array_equal(k, O.blur_kernel(r, sigma))
        d = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
        F.launch_blur(torch.cuda.current_stream().cuda_stream, gpu.to_dev(src), d,
                      w, h, gpu.to_dev(k), r)
        torch.cuda.synchronize()
        got = d.cpu().numpy().view(np.uint32).reshape(h, w)
        assert np.array_equal(got, O.blur(src, k, r))


def test_blur_in_bands_of_rows_is_the_blur(gpu, monkeypatch):
    """A blur is queued as launches over bands of rows of at most 2e10 taps each
    (flutl_launch_blur), so that no surface is too large and no kernel runs for seconds.
    FLUTL_BLUR_BAND_TAPS makes the bands small: the same surface in 2 to 40 bands, through
    either kernel, must be the oracle's blur -- incl. bands that end inside a tile row."""
    import torch
    rng = np.random.default_rng(11)
    h, w = 157, 203
    src = rng.integers(0, 1 << 32, (h, w), dtype=np.uint64).astype(np.uint32)
    for r, sigma in [(3, 1.5), (16, 8.0), (21, 10.5)]:
        k = F.blur_kernel(r, sigma)
        want = O.blur(src, k, r)
        size = 2 * r + 1
        for bands in (2, 7, 40):
            monkeypatch.setenv("FLUTL_BLUR_BAND_TAPS", str(w * h * size * size // bands))
            d = torch.full((h * w * 4,), 0x5A, dtype=torch.uint8, device="cuda")
            F.launch_blur(torch.cuda.current_stream().cuda_stream, gpu.to_dev(src), d,
                          w, h, gpu.to_dev(k), r)
            torch.cuda.synchronize()
            got = d.cpu().numpy().view(np.uint32).reshape(h, w)
            assert np.array_equal(got, want), (r, bands)
    monkeypatch.delenv("FLUTL_BLUR_BAND_TAPS")


def test_a_blur_beyond_the_former_limit_of_taps(gpu):
    """Rounds 1-3 refused w * h * (2r + 1)^2 > 4e11 taps.  A context's blurred atlas entry of
    4.6e11 taps (2048 x 1400 under a radius of 200): accepted, made in two dozen launches,
    and right where it can be told without convolving on the CPU -- the surface is one opaque
    rectangle, so a pixel of the result is, per channel, (c * S + 0x8000) >> 16 with S the sum
    of the kernel's weights whose taps fall inside the rectangle (pixman's weights are cut to
    16.16 one by one: at sigma 100 their sum is well short of 1.0, and the reference's blur is
    that much darker)."""
    W, H, r = 2048, 1400, 200
    x0, y0, x1, y1 = 100, 100, 1948, 1300
    colour = 0xFF336699
    R = np.zeros(1, F.REGION_DTYPE)
    R["x0"], R["y0"], R["x1"], R["y1"], R["first"], R["count"] = 0, 0, W, H, 0, 1
    P = np.zeros(1, F.PRIM_DTYPE)
    P["op"], P["rule"] = F.OP_RECT, F.RULE_PIXMAN
    P["cx0"], P["cy0"], P["cx1"], P["cy1"] = x0, y0, x1, y1
    P["a"], P["b"], P["c"], P["d"] = x0 * 256, y0 * 256, x1 * 256, y1 * 256
    P["color"] = colour
    assert W * H * (2 * r + 1) ** 2 > 4e11
    k = F.blur_kernel(r, r / 2.0).reshape(2 * r + 1, 2 * r + 1).astype(np.int64)
    assert np.array_equal(k.ravel(), O.blur_kernel(r, r / 2.0))
    with F.Context(0) as ctx:
        off, host = ctx.atlas_put_blurred(W, H, R, P, r, r / 2.0)
        assert off > 0

    def expect(x, y):
        # taps (j, i) read the surface at (x - r + i, y - r + j)
        i0, i1 = max(0, x0 - (x - r)), min(2 * r + 1, x1 - (x - r))
        j0, j1 = max(0, y0 - (y - r)), min(2 * r + 1, y1 - (y - r))
        S = int(k[j0:j1, i0:i1].sum()) if i1 > i0 and j1 > j0 else 0
        out = 0
        for sh in (24, 16, 8, 0):
            c = (colour >> sh) & 0xFF
            out |= min(255, max(0, (c * S + 0x8000) >> 16)) << sh
        return out

    rng = np.random.default_rng(5)
    points = [(0, 0), (W - 1, H - 1), (x0, y0), (x0 + r, y0 + r), (1024, 700), (x1 - 1, 700),
              (0, 700), (1024, H - 1)] + \
        [(int(rng.integers(0, W)), int(rng.integers(0, H))) for _ in range(200)]
    for (x, y) in points:
        assert int(host[y, x]) == expect(x, y), (x, y, hex(int(host[y, x])), hex(expect(x, y)))
    assert expect(1024, 700) != 0 and expect(1024, 700) != colour


@pytest.mark.parametrize("size", [(1920, 1080), (3840, 2160), (7680, 4320)])
def test_full_size_properties(gpu, size):
    """At BASELINE sizes (oracle too slow to run densely): size-independent
    properties.  (1) a list rendered at full size equals the oracle on a band
    of it; (2) idempotence: rendering twice gives identical frames;
    (3) pixels outside every region bbox are exactly transparent."""
    W, H = size
    R, P, A = scenes.random_scene(W, W, H, n_regions=12, prims_per_region=40)
    got = gpu.hip_render(W, H, R, P, A)
    again = gpu.hip_render(W, H, R, P, A)
    assert np.array_equal(got, again)
    cover = np.zeros((H, W), bool)
    for r in R:
        cover[r["y0"]:r["y1"], r["x0"]:r["x1"]] = True
    assert (got[~cover] == 0).all()
    if W <= 3840:
        assert np.array_equal(got, O.render(W, H, R, P, A))


def test_context_render_frame_to_host_pinned_and_pageable(gpu):
    """the element's per-buffer entry point, into page-locked and plain memory"""
    W, H = 320, 180
    R, P, A = scenes.random_scene(21, W, H)
    exp = O.render(W, H, R, P, A)
    with F.Context(0) as ctx:
        ctx.atlas_put(A[F.ATLAS_SLACK:])
        for pinned in (True, False, True):
            got = ctx.render_frame_to_host(W, H, R, P, pinned=pinned)
            assert np.array_equal(got, exp)
        # a bigger list than the context's list buffer held so far
        R2, P2, A2 = scenes.random_scene(77, W, H, n_regions=3, prims_per_region=3000,
                                         fixed_count=True)
        ctx.atlas_reset()
        ctx.atlas_put(A2[F.ATLAS_SLACK:])
        assert np.array_equal(ctx.render_frame_to_host(W, H, R2, P2), O.render(W, H, R2, P2, A2))
        with pytest.raises(F.FlutlError):
            bad = P2.copy(); bad["op"][0] = 99
            ctx.render_frame_to_host(W, H, R2, bad)


def test_context_row_band_copy_leaves_other_rows_alone(gpu):
    """flutl_render_frame_rows_to_host: only the requested bands of the host frame are
    written, and they hold the frame's rows."""
    W, H = 320, 180
    R, P, A = scenes.random_scene(5, W, H, n_regions=5)
    exp = O.render(W, H, R, P, A)
    with F.Context(0) as ctx:
        ctx.atlas_put(A[F.ATLAS_SLACK:])
        out = np.full((H, W), 0xDEADBEEF, np.uint32)
        bands = [(3, 40), (41, 42), (100, 180)]
        ctx.render_frame_rows_into(W, H, R, P, out, bands)
        inside = np.zeros(H, bool)
        for y0, y1 in bands:
            inside[y0:y1] = True
        assert np.array_equal(out[inside], exp[inside])
        assert (out[~inside] == 0xDEADBEEF).all()
        ctx.render_frame_rows_into(W, H, R, P, out, [])       # no bands: whole frame
        assert np.array_equal(out, exp)
        # ... and a column window of those bands
        out[:] = 0xDEADBEEF
        ctx.render_frame_rows_into(W, H, R, P, out, bands, cols=(17, 250))
        win = np.zeros((H, W), bool)
        win[inside, 17:250] = True
        assert np.array_equal(out[win], exp[win])
        assert (out[~win] == 0xDEADBEEF).all()
        with pytest.raises(F.FlutlError):
            ctx.render_frame_rows_into(W, H, R, P, out, bands, cols=(10, W + 1))


def test_contexts_on_concurrent_threads(gpu):
    """One context per element instance, several instances in a process, each on its own
    streaming thread (SURVEY 8b threading): results must not interfere."""
    import threading
    W, H = 320, 180
    jobs = [scenes.random_scene(4000 + i, W, H, n_regions=6) for i in range(4)]
    exp = [O.render(W, H, R, P, A) for (R, P, A) in jobs]
    errors = []

    def worker(i):
        try:
            R, P, A = jobs[i]
            with F.Context(0) as ctx:
                ctx.atlas_put(A[F.ATLAS_SLACK:])
                for _ in range(20):
                    got = ctx.render_frame_to_host(W, H, R, P, pinned=(i % 2 == 0))
                    if not np.array_equal(got, exp[i]):
                        errors.append((i, int((got != exp[i]).sum())))
                        return
        except Exception as e:      # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_context_blurred_atlas_entries(gpu):
    """flutl_atlas_put_blurred: sub-list composited on the device, pixman blur, then --
    by the blit position -- as is, bilinear-resampled with edge coverage, or turned a
    quarter clockwise (vertical writing modes); each against the oracle chain."""
    W, H = 96, 40
    R, P, A = scenes.random_scene(31, W, H, n_regions=3, prims_per_region=10)
    surf = O.render(W, H, R, P, A)
    radius, sigma = 3, 1.5
    blurred = O.blur(surf, O.blur_kernel(radius, sigma), radius)
    with F.Context(0) as ctx:
        ctx.atlas_put(A[F.ATLAS_SLACK:])
        off0, plain = ctx.atlas_put_blurred(W, H, R, P, radius, sigma)
        assert np.array_equal(plain, blurred)
        box = [128, 64, 128 + W * 256, 64 + H * 256]
        off1, res = ctx.atlas_put_blurred(W, H, R, P, radius, sigma, distx=64, disty=96, box=box)
        assert np.array_equal(res, O.resample(blurred, 64, 96, box))
        off2, rot = ctx.atlas_put_blurred(W, H, R, P, radius, sigma, rotate_cw=True)
        assert rot.shape == (W, H) and np.array_equal(rot, np.rot90(blurred, -1))
        assert off0 < off1 < off2
        # resample, turn, then the box's edge coverage in the turned axes
        tbox = [100, 300, 100 + (H + 1) * 256 - 150, 300 + (W + 1) * 256 - 500]
        _, both = ctx.atlas_put_blurred(W, H, R, P, radius, sigma, distx=17, disty=120,
                                        box=tbox, rotate_cw=True)
        turned = np.rot90(O.resample(blurred, 17, 120), -1)
        ys, xs = np.mgrid[0:turned.shape[0], 0:turned.shape[1]]
        ov = lambda lo, hi, p: np.clip(np.minimum(hi, p * 256 + 256) - np.maximum(lo, p * 256),
                                       0, None)
        cov = np.minimum((ov(tbox[0], tbox[2], xs) * ov(tbox[1], tbox[3], ys)) >> 8, 255)
        exp = O.group(np.ascontiguousarray(turned).reshape(-1), cov.astype(np.uint32).reshape(-1),
                      np.zeros(turned.size, np.uint32)).reshape(turned.shape)
        assert both.shape == (W + 1, H + 1) and np.array_equal(both, exp)


def _twin_of(frame):
    """the HBM twin of a frame, brought out with the library's own stateless delta
    launch: against an all-zero "twin" every non-zero 256-byte segment is copied"""
    import torch
    n = frame.width * frame.height * 4
    a = torch.zeros(n, dtype=torch.uint8, device="cuda")
    b = torch.zeros(n, dtype=torch.uint8, device="cuda")
    frame.ctx.sync()
    F.launch_delta(torch.cuda.current_stream().cuda_stream, frame.device_ptr(), a, b,
                   frame.width, frame.height, 0, frame.height)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    return b.cpu().numpy().view(np.uint32).reshape(frame.height, frame.width)


@pytest.mark.parametrize("size", [(320, 180), (333, 77), (1920, 1080)])
def test_frame_submit_delta_readback_tracks_every_state(gpu, size):
    """The element's asynchronous per-buffer path (flutl_frame_submit): a sequence of
    different lists into ONE pooled frame -- host block and HBM twin must equal the
    oracle's frame after each, whatever the frame held before (delta read-back),
    including an empty list in between and widths whose rows are not 16-byte
    multiples."""
    W, H = size
    lists = [scenes.random_scene(100 + i, W, H, n_regions=1 + i % 5) for i in range(6)]
    Rm, Pm, A, _ = _concat_lists(lists, W, H)
    with F.Context(0) as ctx:
        assert ctx.atlas_put(A[F.ATLAS_SLACK:]) == F.ATLAS_SLACK
        fr = ctx.frame(W, H)
        try:
            assert not fr.host().any()
            first = 0
            order = [0, 1, 1, None, 2, 3, 0, 4, None, None, 5]
            for k in order:
                if k is None:
                    R, P = Rm[:0], Pm[:0]
                else:
                    nr = len(lists[k][0])
                    lo = sum(len(l[0]) for l in lists[:k])
                    R = Rm[lo:lo + nr].copy()
                    P = Pm
                fr.submit(R, P)
                fr.wait()
                exp = O.render(W, H, R, P, A) if k is not None else np.zeros((H, W), np.uint32)
                assert np.array_equal(fr.host(), exp), f"host block differs after state {k}"
                assert np.array_equal(_twin_of(fr), exp), f"twin differs after state {k}"
        finally:
            fr.close()


@pytest.mark.parametrize("size", [(320, 180), (333, 77)])
def test_frames_of_a_reservation_behave_like_frames_of_their_own(gpu, size):
    """flutl_ctx_reserve_frames: the frames created afterwards take their host block and
    twin from one registered mapping.  They start transparent, track every state like a
    frame of its own, a block comes back (cleared) when its frame is destroyed, a frame
    beyond the reservation is made the old way, and a frame may outlive its context."""
    W, H = size
    lists = [scenes.random_scene(300 + i, W, H, n_regions=2 + i % 3) for i in range(4)]
    Rm, Pm, A, _ = _concat_lists(lists, W, H)
    ctx = F.Context(0)
    ctx.atlas_put(A[F.ATLAS_SLACK:])
    ctx.reserve_frames(W, H, 3)
    ctx.reserve_frames(W, H, 2)          # (already there: nothing new)
    frames = [ctx.frame(W, H) for _ in range(4)]     # the fourth: beyond the reservation
    hosts = [f.host().ctypes.data for f in frames]
    block = (W * H * 4 + 4095) & ~4095
    assert abs(hosts[1] - hosts[0]) == block
    assert abs(hosts[2] - hosts[1]) == block
    def state(k):
        lo = sum(len(l[0]) for l in lists[:k])
        return Rm[lo:lo + len(lists[k][0])].copy()
    try:
        for f in frames:
            assert not f.host().any()
        for step in range(6):
            for i, f in enumerate(frames):
                f.submit(state((step + i) % 4), Pm)
            for i, f in enumerate(frames):
                f.wait()
                exp = O.render(W, H, state((step + i) % 4), Pm, A)
                assert np.array_equal(f.host(), exp), (step, i)
                assert np.array_equal(_twin_of(f), exp), (step, i)
        # a block given back is taken again, and is transparent again in both halves
        frames[1].close()
        again = ctx.frame(W, H)
        assert again.host().ctypes.data == hosts[1]
        again.wait()
        assert not again.host().any() and not _twin_of(again).any()
        again.submit(state(2), Pm)
        again.wait()
        assert np.array_equal(again.host(), O.render(W, H, state(2), Pm, A))
        frames[1] = again
    finally:
        keep = frames[0]
        for f in frames[1:]:
            f.close()
        ctx.close()
    # the context is gone: the frame's memory is still there
    assert np.array_equal(keep.host(), O.render(W, H, state(5 % 4), Pm, A))
    keep.close()


def test_frame_submit_pipelines_without_waiting_in_between(gpu):
    """Submissions are queued, not waited for: many frames in flight through the
    list-slot ring (deeper than the ring), several pooled frames in rotation, one
    wait at the end of each -- every frame must hold ITS state; and the bytes
    moved to the host are far fewer than whole frames when little changes."""
    W, H = 640, 360
    R, P, A = scenes.subtitle_like_scene(7, W, H, n_regions=4, font_px=24, outline_px=2, lines=1)
    glyph = np.nonzero((P["op"] == F.OP_MASK) & (P["rule"] == F.RULE_PIXMAN))[0]
    states = []
    for s in range(12):            # recolour a handful of glyphs: a <set> on tts:color
        Ps = P.copy()
        Ps["color"][glyph[s % len(glyph)::7]] = 0xff000000 | (0x123457 * (s + 1) & 0xffffff)
        states.append(Ps)
    with F.Context(0) as ctx:
        ctx.atlas_put(A[F.ATLAS_SLACK:])
        frames = [ctx.frame(W, H) for _ in range(3)]
        try:
            for s, Ps in enumerate(states):
                frames[s % 3].submit(R, Ps)
            for k in (0, 1, 2):      # the last state each frame received
                last = max(s for s in range(len(states)) if s % 3 == k)
                frames[k].wait()
                assert np.array_equal(frames[k].host(), O.render(W, H, R, states[last], A))
            moved = ctx.delta_bytes()
            assert 0 < moved < len(states) * W * H * 4 // 2
        finally:
            for f in frames:
                f.close()


def test_frames_change_lanes_and_masks_arrive_just_in_time(gpu, monkeypatch):
    """Submissions go to two compositor streams in turn while the atlas is quiet.  Five frames
    in rotation change lanes; every third submission brings a mask (flutl_atlas_put, queued on
    lane 0) that it and later ones -- on either lane -- composite over what an earlier
    submission left in the same frame; a group parks frame pixels in the lane's scratch frame.
    After 60 submissions without a wait in between every frame must hold the oracle's
    frame of ITS last list; and again after waits at odd places."""
    W, H = 640, 200
    rng = np.random.default_rng(17)
    monkeypatch.setenv("FLUTL_LANES_AFTER", "0")      # (a context takes to its second lane after 8 frames)
    with F.Context(0) as ctx:
        frames = [ctx.frame(W, H) for _ in range(5)]
        atlas = [np.zeros(F.ATLAS_SLACK, np.uint8)]
        last = {}
        try:
            masks = []
            for i in range(60):
                # a new mask every third submission (a submission that brings one stays on lane
                # 0; the ones after it alternate and read masks lane 0 uploaded)
                if i % 3 == 0:
                    mw, mh = int(rng.integers(40, 200)), int(rng.integers(10, 60))
                    m = np.zeros((mh, mw + 4), np.uint8)
                    m[:, :mw] = rng.integers(0, 256, (mh, mw))
                    blob = np.concatenate([np.zeros(4, np.uint8), m.ravel()])
                    off = ctx.atlas_put(blob)
                    pad = (-len(np.concatenate(atlas))) % 16
                    assert off == len(np.concatenate(atlas)) + pad
                    atlas += [np.zeros(pad, np.uint8), blob]
                    masks.append((off, mw, mh))
                off, mw, mh = masks[-1] if i % 3 != 2 else masks[int(rng.integers(0, len(masks)))]
                x0, y0 = int(rng.integers(0, W - mw)), int(rng.integers(20, H - mh - 20))
                R = np.zeros(1, F.REGION_DTYPE)
                R["x0"], R["y0"], R["x1"], R["y1"], R["first"], R["count"] = 0, y0 - 20, W, y0 + mh + 20, 0, 4
                P = np.zeros(4, F.PRIM_DTYPE)
                P["cx0"], P["cy0"], P["cx1"], P["cy1"] = 0, y0 - 20, W, y0 + mh + 20
                P["op"] = [F.OP_GROUP_BEGIN, F.OP_RECT, F.OP_MASK, F.OP_GROUP_END]
                P["rule"] = [0, F.RULE_PIXMAN, F.RULE_LERP, 0]
                P["color"][1] = 0x80402010
                P["a"][1], P["b"][1], P["c"][1], P["d"][1] = 0, (y0 - 20) * 256, W * 256, (y0 + mh + 20) * 256
                P["color"][2] = 0xFF000000 | int(rng.integers(0, 1 << 24))
                P["cx0"][2], P["cy0"][2], P["cx1"][2], P["cy1"][2] = x0, y0, x0 + mw, y0 + mh
                P["a"][2], P["b"][2], P["c"][2], P["d"][2] = off + 4, mw + 4, x0, y0
                f = i % 5
                frames[f].submit(R, _with_opacity(P, int(rng.integers(80, 256))))
                last[f] = (R, _with_opacity.last)
                if i in (13, 31, 32, 47):
                    frames[f].wait()
                    A = np.concatenate(atlas + [np.zeros(F.ATLAS_SLACK, np.uint8)])
                    assert np.array_equal(frames[f].host(), O.render(W, H, R, _with_opacity.last, A)), i
            A = np.concatenate(atlas + [np.zeros(F.ATLAS_SLACK, np.uint8)])
            for f, fr in enumerate(frames):
                fr.wait()
                R, P = last[f]
                exp = O.render(W, H, R, P, A)
                assert np.array_equal(fr.host(), exp), f
                assert np.array_equal(_twin_of(fr), exp), f
        finally:
            for fr in frames:
                fr.close()


def _with_opacity(P, alpha):
    """GROUP_END carries the group's 8-bit alpha in `param` (include/flutl.h)"""
    Q = P.copy()
    Q["param"][3] = alpha
    _with_opacity.last = Q
    return Q


@pytest.mark.parametrize("size,rows", [((320, 180), (37, 101)), ((333, 77), (0, 5)),
                                       ((640, 360), (350, 360)), ((640, 360), (16, 32))])
def test_row_window_launch_writes_only_its_supertile_rows(gpu, size, rows):
    """flutl_launch_frame_rows: the supertile rows (16 px) that [row0, row1) touches
    hold the frame's pixels; nothing else of the destination is written."""
    import torch
    W, H = size
    r0, r1 = rows
    R, P, A = scenes.random_scene(31, W, H, n_regions=6)
    exp = O.render(W, H, R, P, A)
    rd, pd, ad = gpu.to_dev(R), gpu.to_dev(P), gpu.to_dev(A)
    dst = torch.full((H * W * 4,), 0xAB, dtype=torch.uint8, device="cuda")
    F.launch_frame_rows(torch.cuda.current_stream().cuda_stream, W, H, dst, rd, len(R), pd, ad,
                        r0, r1)
    torch.cuda.synchronize()
    got = dst.cpu().numpy().view(np.uint32).reshape(H, W)
    y0, y1 = r0 // 16 * 16, min(H, (r1 + 15) // 16 * 16)
    assert np.array_equal(got[y0:y1], exp[y0:y1])
    assert (got[:y0] == 0xABABABAB).all() and (got[y1:] == 0xABABABAB).all()


@pytest.mark.parametrize("size", [(320, 180), (333, 77), (1280, 720)])
def test_blend_over_host_frame_in_place(gpu, size):
    """flutl_blend_over_host (the element's video_sink path): overlay OVER a video
    frame in host memory, in place == (oracle overlay) OVER video, pixman rule; rows
    outside the regions are not touched; an empty list leaves the frame alone."""
    W, H = size
    R, P, A = scenes.random_scene(41, W, H, n_regions=3)
    rng = np.random.default_rng(W)
    video = (rng.integers(0, 1 << 24, (H, W), dtype=np.uint64).astype(np.uint32)) | 0xFF000000
    exp = O.over(O.render(W, H, R, P, A).reshape(-1), video.reshape(-1)).reshape(H, W)
    with F.Context(0) as ctx:
        ctx.atlas_put(A[F.ATLAS_SLACK:])
        got = video.copy()
        ctx.blend_over_host(W, H, R, P, got)
        assert np.array_equal(got, exp), int((got != exp).sum())
        again = video.copy()
        ctx.blend_over_host(W, H, R[:0], P[:0], again)
        assert np.array_equal(again, video)


@pytest.mark.parametrize("size", [(320, 180), (333, 77), (1920, 1080)])
def test_ordered_launch_gives_the_same_frame(gpu, size):
    """flutl_launch_frame_ordered: supertile rows heaviest first (flutl_row_order, a
    permutation of the window's rows) -- speed only, the frame is the oracle's."""
    import torch
    W, H = size
    R, P, A = scenes.subtitle_like_scene(3, W, H, n_regions=5, lines=2)
    order = F.row_order(W, H, R, P)
    assert sorted(order.tolist()) == list(range((H + 15) // 16))
    rd, pd, ad, od = gpu.to_dev(R), gpu.to_dev(P), gpu.to_dev(A), gpu.to_dev(order)
    dst = torch.full((H * W * 4,), 0xAB, dtype=torch.uint8, device="cuda")
    F.launch_frame_ordered(torch.cuda.current_stream().cuda_stream, W, H, dst, rd, len(R), pd,
                           ad, 0, H, od)
    torch.cuda.synchronize()
    got = dst.cpu().numpy().view(np.uint32).reshape(H, W)
    assert np.array_equal(got, O.render(W, H, R, P, A))
    # ... and over a window of rows
    r0, r1 = H // 3, H // 3 * 2
    order = F.row_order(W, H, R, P, r0, r1)
    y0, y1 = r0 // 16 * 16, min(H, (r1 + 15) // 16 * 16)
    assert sorted(order.tolist()) == list(range((y1 - y0 + 15) // 16))
    dst.fill_(0xAB)
    F.launch_frame_ordered(torch.cuda.current_stream().cuda_stream, W, H, dst, rd, len(R), pd,
                           ad, r0, r1, gpu.to_dev(order))
    torch.cuda.synchronize()
    got = dst.cpu().numpy().view(np.uint32).reshape(H, W)
    assert np.array_equal(got[y0:y1], O.render(W, H, R, P, A)[y0:y1])
    assert (got[:y0] == 0xABABABAB).all() and (got[y1:] == 0xABABABAB).all()


@pytest.mark.parametrize("size", [(320, 180), (333, 77), (1280, 720)])
def test_fused_delta_launch_tracks_a_sequence_of_lists(gpu, size):
    """flutl_launch_frame_delta, stateless: a sequence of different lists (overlapping
    translucent groups among them: they park pixels in the scratch frame) into ONE
    twin / host pair, each launch over the rows the old and the new content span.
    After each: twin == host == the oracle's frame; the segment counter moves only
    when something changed; a second launch of the same list writes nothing."""
    import torch
    W, H = size
    st = torch.cuda.current_stream().cuda_stream
    twin = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
    host = torch.zeros_like(twin)
    park = torch.full_like(twin, 0x5A)
    changed = torch.zeros(1, dtype=torch.int64, device="cuda")
    seg = F.delta_segment_bytes()
    assert seg in (64, 128, 256)
    prev = (0, 0)
    for i in range(7):
        if i == 4:
            R, P, A = scenes.random_scene(900, W, H, n_regions=0)      # an empty frame in between
        else:
            R, P, A = scenes.random_scene(900 + i, W, H, n_regions=2 + i % 4)
        ys = [int(r["y0"]) for r in R if r["y1"] > r["y0"]] + [prev[0]] * (prev[1] > prev[0])
        ye = [int(r["y1"]) for r in R if r["y1"] > r["y0"]] + [prev[1]] * (prev[1] > prev[0])
        now = (min([int(r["y0"]) for r in R if r["y1"] > r["y0"]], default=0),
               max([int(r["y1"]) for r in R if r["y1"] > r["y0"]], default=0))
        if not ys:
            continue
        y0, y1 = min(ys), max(ye)
        order = F.row_order(W, H, R, P, y0, y1)
        rd, pd, ad, od = gpu.to_dev(R), gpu.to_dev(P), gpu.to_dev(A), gpu.to_dev(order)
        for again in (False, True):
            before = int(changed.item())
            F.launch_frame_delta(st, W, H, twin, host, park, rd, len(R), pd, ad, y0, y1, od, changed)
            torch.cuda.synchronize()
            exp = O.render(W, H, R, P, A)
            t = twin.cpu().numpy().view(np.uint32).reshape(H, W)
            h = host.cpu().numpy().view(np.uint32).reshape(H, W)
            assert np.array_equal(t, exp) and np.array_equal(h, exp), (i, again)
            if again:
                assert int(changed.item()) == before, "an unchanged frame wrote segments"
        prev = now
    assert 0 < int(changed.item()) * seg <= 7 * (W * 4 + seg) * H


def test_a_frame_written_into_by_its_consumer_is_put_right_by_the_next_submission(gpu):
    """FLUTL_WRITABLE: downstream may work in place in a frame's host block.  After
    flutl_frame_mark_written the next submission copies the block up to the twin and composites
    every row, so that block and twin both hold the new frame again -- wherever the consumer
    scribbled (inside and outside the regions)."""
    W, H = 640, 360
    R, P, A = scenes.random_scene(77, W, H, n_regions=5)
    R2, P2, A2 = scenes.random_scene(78, W, H, n_regions=3)
    atlas = np.concatenate([A, A2[F.ATLAS_SLACK:]])
    P2 = P2.copy()
    mask = (P2["op"] == F.OP_MASK) | (P2["op"] == F.OP_IMAGE)
    P2["a"][mask] += len(A) - F.ATLAS_SLACK
    with F.Context(0) as ctx:
        ctx.atlas_put(atlas[F.ATLAS_SLACK:])
        fr = ctx.frame(W, H)
        try:
            fr.submit(R, P)
            fr.wait()
            assert np.array_equal(fr.host(), O.render(W, H, R, P, atlas))
            host = fr.host()
            host[::7, ::5] ^= 0x00ff00ff            # (the consumer's work, all over the frame)
            host[H - 3:, :] = 0x12345678
            fr.mark_written()
            fr.submit(R2, P2)
            fr.wait()
            want = O.render(W, H, R2, P2, atlas)
            assert np.array_equal(fr.host(), want)
            # ... and the twin with it: a further submission of the same list writes nothing
            before = ctx.delta_bytes()
            fr.submit(R2, P2)
            fr.wait()
            assert ctx.delta_bytes() == before and np.array_equal(fr.host(), want)
        finally:
            fr.close()
