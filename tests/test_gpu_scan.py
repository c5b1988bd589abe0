"""-m gpu: the polygon scan converter (csrc/flutl_scan.hip) through the C-ABI against
the REAL libcairo's coverage of committed polygons (tests/golden/polygons.npz, made by
oracle/polygon_kats.c: cairo_fill / cairo_stroke on A8 surfaces) and against the CPU
restatement (oracle/flutl_tor_oracle.c) on fresh random polygons.  Bit exact."""
import os

import numpy as np
import pytest

import oracle_lib as O
import pyflutl as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "needs cuda:0"
    return torch


def _polygons():
    z = np.load(os.path.join(ROOT, "tests", "golden", "polygons.npz"))
    for i, (kind, x0, y0, x1, y1, eo) in enumerate(z["meta"]):
        yield i, int(kind), (int(x0), int(y0), int(x1), int(y1)), bool(eo), z[f"edges{i}"], z[f"cov{i}"]


def hip_scan(torch, edges, box, even_odd, mode=0, stride_pad=5):
    """(coverage h x w, status) of the polygon through flutl_launch_scan"""
    edges = np.ascontiguousarray(edges, np.int32).reshape(-1, 7)
    w, h = box[2] - box[0], box[3] - box[1]
    stride = w + stride_pad
    ed = torch.from_numpy(edges.view(np.uint8).reshape(-1).copy()).cuda() if len(edges) else \
        torch.zeros(64, dtype=torch.uint8, device="cuda")
    scratch = torch.empty(F.scan_scratch_bytes(len(edges), w, h) + 256, dtype=torch.uint8,
                          device="cuda")
    dst = torch.full((h * stride,), 0xAB, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    F.launch_scan(stream, ed, len(edges), box, even_odd, scratch, dst, stride, mode)
    status = F.scan_status(stream, scratch, len(edges), h) if len(edges) else 0
    torch.cuda.synchronize()
    out = dst.cpu().numpy().reshape(h, stride)
    assert (out[:, w:] == 0xAB).all(), "the kernel wrote beyond its rows"
    return out[:, :w].copy(), status


def test_committed_polygons_are_cairos_coverage(gpu):
    """rows in parallel, the sequential converter only where the rows flagged"""
    bad = []
    for i, kind, box, eo, edges, cov in _polygons():
        got, _ = hip_scan(gpu, edges, box, eo, mode=0)
        if not np.array_equal(got, cov):
            bad.append((i, kind, int((got != cov).sum())))
    assert not bad, f"{len(bad)} polygons differ from Cairo: {bad[:10]}"


def test_row_kernel_alone_decides_almost_every_polygon(gpu):
    """mode 1: no sequential converter behind the rows.  Whatever the rows did not flag
    must be Cairo's coverage already; what they flag (a tie of sort keys that the
    history rules cannot decide) must be rare."""
    flagged, n = 0, 0
    for i, kind, box, eo, edges, cov in _polygons():
        got, status = hip_scan(gpu, edges, box, eo, mode=1)
        n += 1
        if status:
            flagged += 1
            continue
        assert np.array_equal(got, cov), f"polygon {i} (kind {kind}): unflagged and wrong"
    # (the fills on coarse grids are all ties; text strokes flag about 1 in 700)
    assert flagged <= n // 4, f"{flagged} of {n} polygons flagged"


def test_sequential_converter_is_cairos_coverage(gpu):
    """mode 2: the one-lane sweep alone (every 4th polygon, the small ones: it is slow)"""
    n = 0
    for i, kind, box, eo, edges, cov in _polygons():
        if i % 4 or len(edges) > 1500:
            continue
        got, _ = hip_scan(gpu, edges, box, eo, mode=2)
        assert np.array_equal(got, cov), f"polygon {i} (kind {kind})"
        n += 1
    assert n >= 40


def test_random_polygons_match_the_restatement(gpu):
    """fresh random polygons (seeded), all snapped to grids that provoke ties, against
    oracle/flutl_tor_oracle.c"""
    rng = np.random.default_rng(7)
    for it in range(120):
        W, H = int(rng.integers(40, 300)), int(rng.integers(30, 200))
        snap = int(rng.choice([1, 1, 17, 64, 128, 256]))
        edges = []
        for sp in range(int(rng.integers(1, 4))):
            n = int(rng.integers(3, 40))
            px = rng.integers(256, (W - 1) * 256, n) // snap * snap
            py = rng.integers(256, (H - 1) * 256, n) // snap * snap
            for k in range(n):
                x1, y1, x2, y2 = int(px[k]), int(py[k]), int(px[(k + 1) % n]), int(py[(k + 1) % n])
                if y1 == y2:
                    continue
                d = 1
                if y1 > y2:
                    x1, y1, x2, y2, d = x2, y2, x1, y1, -1
                edges.append((x1, y1, x2, y2, y1, y2, d))
        if not edges:
            continue
        edges = np.array(edges, np.int32)
        box = O.polygon_box(edges)
        box = (max(int(box[0]), 0), max(int(box[1]), 0), min(int(box[2]), W), min(int(box[3]), H))
        eo = bool(it % 3 == 0)
        want = O.scan_polygon(edges, box, eo)
        got, _ = hip_scan(gpu, edges, box, eo, mode=0)
        assert np.array_equal(got, want), f"case {it}: {(got != want).sum()} px differ"


def _random_polygon(rng, W, H, n, snap):
    px = rng.integers(-20 * 256, (W + 20) * 256, n) // snap * snap
    py = rng.integers(-10 * 256, (H + 10) * 256, n) // snap * snap
    edges = []
    for k in range(n):
        x1, y1, x2, y2 = int(px[k]), int(py[k]), int(px[(k + 1) % n]), int(py[(k + 1) % n])
        if y1 == y2:
            continue
        d = 1
        if y1 > y2:
            x1, y1, x2, y2, d = x2, y2, x1, y1, -1
        edges.append((x1, y1, x2, y2, y1, y2, d))
    return np.array(edges, np.int32)


@pytest.mark.parametrize("W,H,n,snap", [
    (900, 24, 1800, 1),       # ~600 edges in every row: all sample rows together, full bins
    (900, 24, 1800, 64),      # ... with ties of cells all over
    (3900, 20, 500, 1),       # bins of 4 pixels
    (4090, 12, 900, 256),     # the widest box, pixel-aligned vertices
    (700, 16, 2400, 1),       # 1024 < edges per row <= 2048: one sample row after the other
    (700, 16, 2400, 128),
    (500, 10, 5200, 1),       # > 2048 edges in a row: the overflow flag, the sequential sweep
])
def test_crowded_rows_and_windows(gpu, W, H, n, snap):
    """rows with hundreds to thousands of edges (each path of the row kernel: sample rows
    together, one by one, overflow), vertices beyond every side of the box (edges clipped
    left, right, above, below), against oracle/flutl_tor_oracle.c"""
    rng = np.random.default_rng(W * 31 + n + snap)
    edges = _random_polygon(rng, W, H, n, snap)
    box = (0, 0, W, H)
    for eo in (False, True):
        want = O.scan_polygon(edges, box, eo)
        got, _ = hip_scan(gpu, edges, box, eo, mode=0)
        assert np.array_equal(got, want), f"{(got != want).sum()} px differ (even_odd {eo})"
    # how many edges the busiest row has (the case must exercise what it says)
    top = (edges[:, 4].astype(np.int64) * 15 + 128) >> 8
    bot = (edges[:, 5].astype(np.int64) * 15 + 128) >> 8
    busiest = max(int(((top < (r + 1) * 15) & (bot > r * 15)).sum()) for r in range(H))
    if n == 2400:
        assert 1024 < busiest <= 2048
    elif n == 5200:
        assert busiest > 2048
    else:
        assert busiest <= 1024


def test_polygon_becomes_an_atlas_entry_masks_can_read(gpu):
    """flutl_atlas_put_polygon: the coverage lands in the context's atlas in the MASK
    layout (4 zero bytes, rows of width + 4), composited by a MASK primitive it must give
    what the same mask uploaded from the host gives."""
    import gpu_util
    torch = gpu
    i, kind, box, eo, edges, cov = next(p for p in _polygons() if p[1] == 2)
    h, w = cov.shape
    W, H = box[2] + 8, box[3] + 8
    with F.Context(0) as ctx:
        off = ctx.atlas_put_polygon(edges, box, eo)
        fr = ctx.frame(W, H)
        try:
            R = np.zeros(1, F.REGION_DTYPE)
            R["x0"], R["y0"], R["x1"], R["y1"], R["first"], R["count"] = box[0], box[1], box[2], box[3], 0, 1
            P = np.zeros(1, F.PRIM_DTYPE)
            P["op"], P["rule"] = F.OP_MASK, F.RULE_LERP
            P["cx0"], P["cy0"], P["cx1"], P["cy1"] = box[0], box[1], box[2], box[3]
            P["color"] = 0xFF000000
            P["a"], P["b"], P["c"], P["d"] = off + 4, w + 4, box[0], box[1]
            fr.submit(R, P)
            fr.wait()
            got = fr.host().copy()
        finally:
            fr.close()
    # the same through the stateless path with the mask uploaded from the host
    padded = np.zeros((h, w + 4), np.uint8)
    padded[:, :w] = cov
    atlas = np.concatenate([np.zeros(F.ATLAS_SLACK + 4, np.uint8), padded.ravel(),
                            np.zeros(F.ATLAS_SLACK, np.uint8)])
    P2 = P.copy()
    P2["a"] = F.ATLAS_SLACK + 4
    want = gpu_util.hip_render(W, H, R, P2, atlas)
    assert np.array_equal(got, want)


def test_a_window_of_the_box_is_the_crop_of_the_whole_and_wide_entries_use_that(gpu):
    """The converter's boxes are at most 4096 pixels wide (a row's accumulators live in
    LDS); flutl_atlas_put_polygon converts a wider entry -- a line of 8K text -- in column
    windows.  That is the same coverage because the conversion of a window is the crop of
    the conversion of the whole box (checked here against the oracle for windows of a
    committed text stroke), and an entry 9 100 pixels wide must read back as the oracle's
    conversion of the whole box."""
    torch = gpu
    i, kind, box, eo, edges, cov = next(p for p in _polygons() if p[1] == 2)
    rng = np.random.default_rng(11)
    for _ in range(6):
        a = int(rng.integers(box[0], box[2] - 4))
        b = int(rng.integers(a + 2, box[2] + 1))
        got, _ = hip_scan(torch, edges, (a, box[1], b, box[3]), eo, mode=0)
        assert np.array_equal(got, cov[:, a - box[0]:b - box[0]])
    # a wide polygon: random, vertices beyond the box
    W, H = 9100, 14
    wide = _random_polygon(np.random.default_rng(5), W, H, 1200, 1)
    want = O.scan_polygon(wide, (0, 0, W, H), False)
    import gpu_util
    FW, FH = W + 8, H + 8
    R = np.zeros(1, F.REGION_DTYPE)
    R["x0"], R["y0"], R["x1"], R["y1"], R["first"], R["count"] = 0, 0, W, H, 0, 1
    P = np.zeros(1, F.PRIM_DTYPE)
    P["op"], P["rule"] = F.OP_MASK, F.RULE_LERP
    P["cx0"], P["cy0"], P["cx1"], P["cy1"] = 0, 0, W, H
    P["color"] = 0xFF000000
    with F.Context(0) as ctx:
        off = ctx.atlas_put_polygon(wide, (0, 0, W, H), False)
        fr = ctx.frame(FW, FH)
        try:
            P["a"], P["b"], P["c"], P["d"] = off + 4, W + 4, 0, 0
            fr.submit(R, P)
            fr.wait()
            got = fr.host().copy()
        finally:
            fr.close()
    # the oracle's coverage of the whole box as a mask uploaded from the host
    padded = np.zeros((H, W + 4), np.uint8)
    padded[:, :W] = want
    atlas = np.concatenate([np.zeros(F.ATLAS_SLACK + 4, np.uint8), padded.ravel(),
                            np.zeros(F.ATLAS_SLACK, np.uint8)])
    P2 = P.copy()
    P2["a"] = F.ATLAS_SLACK + 4
    assert np.array_equal(got, gpu_util.hip_render(FW, FH, R, P2, atlas))
    assert int((want > 0).sum()) > 10000


def test_entries_survive_the_growing_of_the_atlas(gpu):
    """The context's atlas starts at 320 MB (an element starts its caches over at 256 MB)
    and doubles when more is asked of it: a device allocation, a copy, both streams drained.
    An outline made on the device and a mask uploaded before the growing must composite the
    same afterwards."""
    import gpu_util
    torch = gpu
    i, kind, box, eo, edges, cov = next(p for p in _polygons() if p[1] == 2)
    h, w = cov.shape
    W, H = box[2] + 8, box[3] + 8
    R = np.zeros(1, F.REGION_DTYPE)
    R["x0"], R["y0"], R["x1"], R["y1"], R["first"], R["count"] = box[0], box[1], box[2], box[3], 0, 1
    P = np.zeros(1, F.PRIM_DTYPE)
    P["op"], P["rule"] = F.OP_MASK, F.RULE_LERP
    P["cx0"], P["cy0"], P["cx1"], P["cy1"] = box[0], box[1], box[2], box[3]
    P["color"] = 0xFF000000
    with F.Context(0) as ctx:
        off = ctx.atlas_put_polygon(edges, box, eo)
        P["a"], P["b"], P["c"], P["d"] = off + 4, w + 4, box[0], box[1]
        filler = np.full(48 << 20, 0x5A, np.uint8)
        before = ctx.atlas_size()
        for _ in range(8):              # 384 MB: beyond the first allocation
            ctx.atlas_put(filler)
        assert ctx.atlas_size() >= before + 8 * len(filler)
        off2 = ctx.atlas_put_polygon(edges, box, eo)      # ... and one made after it
        fr = ctx.frame(W, H)
        try:
            got = []
            for o in (off, off2):
                P["a"] = o + 4
                fr.submit(R, P)
                fr.wait()
                got.append(fr.host().copy())
        finally:
            fr.close()
    padded = np.zeros((h, w + 4), np.uint8)
    padded[:, :w] = cov
    atlas = np.concatenate([np.zeros(F.ATLAS_SLACK + 4, np.uint8), padded.ravel(),
                            np.zeros(F.ATLAS_SLACK, np.uint8)])
    P2 = P.copy()
    P2["a"] = F.ATLAS_SLACK + 4
    want = gpu_util.hip_render(W, H, R, P2, atlas)
    assert np.array_equal(got[0], want) and np.array_equal(got[1], want)


def test_edges_written_into_the_contexts_blocks_and_blocks_coming_round(gpu):
    """The element's stroke workers write a polygon's edges straight into one of the
    context's page-locked blocks (flutl_ctx_edge_block_acquire) and the device reads them in
    place.  Same entries as with edges copied by flutl_atlas_put_polygon; 100 polygons in a row
    pass through the 32 blocks three times over (a block is free again behind the event of its
    epoch); blocks held by a caller run out without anybody waiting, and come back on release."""
    import ctypes
    import gpu_util
    polys = [p for p in _polygons() if p[1] == 2][:5]
    W = max(p[2][2] for p in polys) + 8
    H = max(p[2][3] for p in polys) + 8

    def masks(ctx, via):
        out = []
        for rep in range(20):
            for (i, kind, box, eo, edges, cov) in polys:
                out.append((ctx.atlas_put_polygon(edges, box, eo, via_block=via), box, cov))
        return out

    def composite(ctx, entries):
        frames = []
        fr = ctx.frame(W, H)
        try:
            for off, box, cov in entries[::7]:
                h, w = cov.shape
                R = np.zeros(1, F.REGION_DTYPE)
                R["x0"], R["y0"], R["x1"], R["y1"], R["first"], R["count"] = box[0], box[1], box[2], box[3], 0, 1
                P = np.zeros(1, F.PRIM_DTYPE)
                P["op"], P["rule"] = F.OP_MASK, F.RULE_LERP
                P["cx0"], P["cy0"], P["cx1"], P["cy1"] = box[0], box[1], box[2], box[3]
                P["color"] = 0xFF000000
                P["a"], P["b"], P["c"], P["d"] = off + 4, w + 4, box[0], box[1]
                fr.submit(R, P)
                fr.wait()
                frames.append(fr.host().copy())
        finally:
            fr.close()
        return frames

    with F.Context(0) as ctx:
        a = composite(ctx, masks(ctx, "must"))
    with F.Context(0) as ctx:
        b = composite(ctx, masks(ctx, False))
        # blocks a caller holds: 32, then none (nobody waits), then one again after a release
        L = F.lib()
        held = []
        for _ in range(40):
            blk = ctypes.c_void_p()
            if L.flutl_ctx_edge_block_acquire(ctx._h, 4096, ctypes.byref(blk)) == 0:
                held.append(blk)
        assert len(held) == 32
        assert L.flutl_ctx_edge_block_acquire(ctx._h, (1 << 20) + 1, ctypes.byref(ctypes.c_void_p())) != 0
        L.flutl_ctx_edge_block_release(ctx._h, held.pop())
        blk = ctypes.c_void_p()
        assert L.flutl_ctx_edge_block_acquire(ctx._h, 4096, ctypes.byref(blk)) == 0
        # (a polygon put while every block is out: copied by an upload instead, still right)
        i, kind, box, eo, edges, cov = polys[0]
        off = ctx.atlas_put_polygon(edges, box, eo)
        c = composite(ctx, [(off, box, cov)])
        for h_ in held + [blk]:
            L.flutl_ctx_edge_block_release(ctx._h, h_)
    assert len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))
    assert np.array_equal(c[0], a[0])


def test_edges_prepared_inside_the_row_kernel(gpu):
    """mode | 64 (FLUTL_SCAN_FUSED): the row kernel's first workgroups prepare the edges and
    every workgroup waits for them -- the same coverage, for polygons of more slices than rows
    and of more rows than slices."""
    torch = gpu
    done = 0
    for i, kind, box, eo, edges, cov in _polygons():
        if i % 9:
            continue
        got, _ = hip_scan(torch, edges, box, eo, mode=64)
        assert np.array_equal(got, cov), (i, kind)
        done += 1
    assert done >= 30


def test_a_staged_frames_list_goes_up_with_its_outlines_conversion(gpu):
    """flutl_frame_stage / flutl_frame_submit_staged: the frame's list is staged, THEN its
    outline's polygon is converted into the atlas (the conversion's kernel carries the list up
    to the device), then the composite is queued -- the same frame as with the polygon made
    first and a plain submit; also with no conversion in between (the list goes up by a copy),
    and twice in a row."""
    polys = [p for p in _polygons() if p[1] == 2][:3]

    def frame_of(ctx, fr, off, box, cov, colour):
        h, w = cov.shape
        R = np.zeros(1, F.REGION_DTYPE)
        R["x0"], R["y0"], R["x1"], R["y1"], R["first"], R["count"] = box[0], box[1], box[2], box[3], 0, 1
        P = np.zeros(1, F.PRIM_DTYPE)
        P["op"], P["rule"] = F.OP_MASK, F.RULE_LERP
        P["cx0"], P["cy0"], P["cx1"], P["cy1"] = box[0], box[1], box[2], box[3]
        P["color"] = colour
        P["a"], P["b"], P["c"], P["d"] = off + 4, w + 4, box[0], box[1]
        return R, P

    W = max(p[2][2] for p in polys) + 8
    H = max(p[2][3] for p in polys) + 8
    got, want = [], []
    with F.Context(0) as ctx:
        fr = ctx.frame(W, H)
        try:
            for k, (i, kind, box, eo, edges, cov) in enumerate(polys * 2):
                # the entry's offset is known before it is made: the next free one
                off = (ctx.atlas_size() - F.ATLAS_SLACK + 15) & ~15
                R, P = frame_of(ctx, fr, off, box, cov, 0xFF000000 | (0x40 * (k + 1)))
                fr.stage(R, P)
                assert ctx.atlas_put_polygon(edges, box, eo) == off
                fr.submit_staged()
                fr.wait()
                got.append(fr.host().copy())
            # no conversion between stage and submit: an entry that exists already
            fr.stage(R, P)
            fr.submit_staged()
            fr.wait()
            got.append(fr.host().copy())
        finally:
            fr.close()
    with F.Context(0) as ctx:
        fr = ctx.frame(W, H)
        try:
            for k, (i, kind, box, eo, edges, cov) in enumerate(polys * 2):
                off = ctx.atlas_put_polygon(edges, box, eo)
                R, P = frame_of(ctx, fr, off, box, cov, 0xFF000000 | (0x40 * (k + 1)))
                fr.submit(R, P)
                fr.wait()
                want.append(fr.host().copy())
            want.append(want[-1])
        finally:
            fr.close()
    assert len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want))


def test_whatever_reads_the_atlas_comes_behind_the_conversions_queued_before_it(gpu):
    """The conversions run on a stream of their own.  Fifty of them queued back to back put
    that stream a millisecond behind the host; a reader queued right away on the other
    stream -- a frame submit, the synchronous render, the blend over a video frame -- must
    still see the LAST entry complete; and after flutl_atlas_reset the bytes are used again:
    a mask uploaded over them (on the compositor's stream) must not be overtaken by, nor
    overtake, the conversions that wrote there before."""
    import gpu_util
    torch = gpu
    i, kind, box, eo, edges, cov = next(p for p in _polygons() if p[1] == 2)
    h, w = cov.shape
    W, H = box[2] + 8, box[3] + 8
    R = np.zeros(1, F.REGION_DTYPE)
    R["x0"], R["y0"], R["x1"], R["y1"], R["first"], R["count"] = box[0], box[1], box[2], box[3], 0, 1
    P = np.zeros(1, F.PRIM_DTYPE)
    P["op"], P["rule"] = F.OP_MASK, F.RULE_LERP
    P["cx0"], P["cy0"], P["cx1"], P["cy1"] = box[0], box[1], box[2], box[3]
    P["color"] = 0xFF000000
    P["b"], P["c"], P["d"] = w + 4, box[0], box[1]
    padded = np.zeros((h, w + 4), np.uint8)
    padded[:, :w] = cov
    atlas = np.concatenate([np.zeros(F.ATLAS_SLACK + 4, np.uint8), padded.ravel(),
                            np.zeros(F.ATLAS_SLACK, np.uint8)])
    P2 = P.copy()
    P2["a"] = F.ATLAS_SLACK + 4
    want = gpu_util.hip_render(W, H, R, P2, atlas)
    video = np.full((H, W), 0xFF204060, np.uint32)
    import oracle_lib as O
    want_over = O.over(want.reshape(-1), video.reshape(-1)).reshape(H, W)

    def burst(ctx, n=50):
        off = 0
        for _ in range(n):
            off = ctx.atlas_put_polygon(edges, box, eo)
        return off

    with F.Context(0) as ctx:
        fr = ctx.frame(W, H)
        try:
            P["a"] = burst(ctx) + 4
            fr.submit(R, P)
            fr.wait()
            assert np.array_equal(fr.host(), want), "frame submit overtook the conversions"
            P["a"] = burst(ctx) + 4
            assert np.array_equal(ctx.render_frame_to_host(W, H, R, P), want), "render overtook them"
            P["a"] = burst(ctx) + 4
            v = video.copy()
            ctx.blend_over_host(W, H, R, P, v)
            assert np.array_equal(v, want_over), "blend over video overtook them"
            # the bytes used again: an upload over where conversions are still writing
            burst(ctx)
            ctx.atlas_reset()
            inverse = padded.copy()
            inverse[:, :w] = 255 - cov
            off = ctx.atlas_put(np.concatenate([np.zeros(4, np.uint8), inverse.ravel()]))
            P["a"] = off + 4
            got = ctx.render_frame_to_host(W, H, R, P)
            atlas_inv = np.concatenate([np.zeros(F.ATLAS_SLACK + 4, np.uint8), inverse.ravel(),
                                        np.zeros(F.ATLAS_SLACK, np.uint8)])
            assert np.array_equal(got, gpu_util.hip_render(W, H, R, P2, atlas_inv)), \
                "an upload after the reset was overwritten by an earlier conversion"
            # ... and a conversion over where an upload went, read by a submit
            ctx.atlas_put(np.full(4 << 20, 0x77, np.uint8))
            ctx.atlas_reset()
            P["a"] = burst(ctx, 3) + 4
            fr.submit(R, P)
            fr.wait()
            assert np.array_equal(fr.host(), want)
        finally:
            fr.close()
