"""The caption scan's cut-outs the reference's way (smr_hip_windows_create / smr_hip_draw_windows, windows.hip): every caption's
r x r window re-rasterised at the sub-pixel offset of its anchor, as cairo_smr_cut_out (src/smrules_cairo.c:1511-1533) replays
the recording.  Pinned against libcairo's OWN cut-outs: tests/golden/cutout_subpixel.npz holds, for 2 x 2000 captions with
non-integer anchors on a C2-like chart, find_angle on cd_cut_out's pixels (`ref`, made by tests/golden/make_cutout_vectors.py
where libcairo 1.16 is).  The crop convention of smr_hip_caprot on the chart's canvas differs from `ref` in 11-13 % of the
%.1f tags (tests/test_cutout_convention.py); the windows must not."""
import ctypes as C
import os
import numpy as np
import pytest
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "cutout_subpixel.npz"))
W, H, DPI = int(G["W"]), int(G["H"]), int(G["dpi"])


def tags(a):
    return np.array(["%.1f" % (v * 180 / np.pi) for v in a])


def scan_windows(sa, rules, cx, cy, fw, fh, fx, col=0, weight=1.0, bg=0xffffff, per_atlas=None):
    """find_angle for every anchor through atlases of windows; rules = [(DrawStyle, device arrays x, y, off, ref, n_ways, n_refs)]"""
    r = C.c_int(); L = po.orc()
    L.orc_num_steps(fw, fh, DPI, C.byref(r)); r = r.value
    pitch = (r + 3) & ~3
    per_atlas = per_atlas or 65536 // pitch
    out = np.zeros(len(cx))
    for k0 in range(0, len(cx), per_atlas):
        k1 = min(len(cx), k0 + per_atlas)
        at = sa.Canvas.windows(W, H, DPI, bg, cx[k0:k1], cy[k0:k1], r, float(np.hypot(fw, fh)))
        for st, x, y, off, ref, nw, nr in rules:
            at.draw_windows(st, x, y, off, ref, nw, nr)
        ax, ay = at.window_anchors()
        d = sa.DeviceArray((k1 - k0,), np.float64)
        at.caprot(sa.DeviceArray.from_numpy(ax), sa.DeviceArray.from_numpy(ay), k1 - k0, fw, fh, fx, col, 0.0, weight, d)
        sa.sync()
        out[k0:k1] = d.numpy()
        at.close()
    return out


def test_windows_equal_libcairos_own_cutouts(hip):
    """2 x 2000 captions: the angles are find_angle's on libcairo's sub-pixel cut-outs"""
    sa = hip
    x, y, off, ref = po.synth_ways(int(G["n_ways"]), W, H, seed=int(G["seed"]))
    D = sa.DeviceArray.from_numpy
    st = sa.DrawStyle(1, 1, 0x4080c0, 0x000000, 0.0, 25.4 / DPI, 0, 0, 0, 0, 0, 0)
    rules = [(st, D(x), D(y), D(off.astype(np.int64)), D(ref.astype(np.int32)), len(off) - 1, len(ref))]
    for tag in ("c4", "small"):
        fw, fh, fx = (int(v) for v in G[tag + "_plane"])
        got = scan_windows(sa, rules, G[tag + "_cx"], G[tag + "_cy"], fw, fh, fx)
        want = G[tag + "_ref"]
        same = tags(got) == tags(want)
        print(tag, "tags equal to libcairo's cut-outs:", int(same.sum()), "of", len(want), "| crop convention:",
              int((tags(G[tag + "_gpu"]) == tags(want)).sum()))
        assert same.mean() >= 0.999, (tag, int((~same).sum()))


STYLES = {
    # strokes as libcairo's own polygons (SMR_HIP_STROKE=polygon / ORC_STROKE_OUTLINE=1): their edges cross one another at inner
    # joins, also beyond the sides of the replay's limits, where libcairo has projected them onto the side before it looks
    "polygon strokes": po.Style(1, 1, 0x4080c0, 0x000000, 0.4, 0.3, 0),
    "translucent": po.Style(1, 1, 0x304080c0, 0x50102030, 0, 0.2, 0),
    "curve": None,          # made below: curve=1 fill + border
    "dashed": None,
    "open+closed wide": po.Style(1, 1, 0x4080c0, 0x000000, 0.5, 0.3, 0),
}


def _style(name):
    if name == "curve":
        return po.Style(1, 1, 0x4080c0, 0x203040, 0, 0.25, 0).with_curve(0.2)
    if name == "dashed":
        return po.Style(0, 1, 0, 0x204060, 0, 0.3, 0).with_dash(border_style=1)
    return STYLES[name]


@pytest.mark.parametrize("name", list(STYLES))
def test_window_pixels_equal_the_oracles_shifted_render(hip, name, monkeypatch):
    """a handful of windows pixel by pixel against the oracle rendering the same ways with the same 24.8 shift
    (orc_canvas_set_shift: what libcairo's replay at an offset does to the recorded path), closed and open ways"""
    sa = hip
    if name == "polygon strokes":
        monkeypatch.setenv("SMR_HIP_STROKE", "polygon")
        monkeypatch.setenv("ORC_STROKE_OUTLINE", "1")
    L = po.orc()
    L.orc_canvas_set_shift.argtypes = [C.c_void_p, C.c_double, C.c_double]
    x, y, off, ref = po.synth_ways(1500, 900, 700, seed=21)
    x2, y2, off2, ref2 = po.synth_ways(600, 900, 700, seed=22)
    keep = np.ones(len(ref2), bool); keep[off2[1:] - 1] = False               # without their closing refs: open ways
    ref2, off2 = ref2[keep], off2 - np.arange(len(off2))
    ways = po.ways_from_arrays(x, y, off, ref) + po.ways_from_arrays(x2, y2, off2, ref2)
    X, Y = np.concatenate([x, x2]), np.concatenate([y, y2])
    OFF = np.concatenate([off, off2[1:] + off[-1]]).astype(np.int64)
    REF = np.concatenate([ref, ref2 + len(x)]).astype(np.int32)
    sty = _style(name)
    rng = np.random.default_rng(3)
    n, r = 7, 93
    cx, cy = rng.uniform(-10, 910, n), rng.uniform(-10, 710, n)            # some windows hang over the chart's edge
    D = sa.DeviceArray.from_numpy
    at = sa.Canvas.windows(900, 700, DPI, 0xfff8e0, cx, cy, r)
    from golden_io import draw_style
    at.draw_windows(draw_style(sa, sty), D(X), D(Y), D(OFF), D(REF), len(OFF) - 1, len(REF))
    img = at.read()
    at.close()
    r_pt = r * 72 / DPI
    reach = 40 if name != "curve" else 120
    # libcairo's own cut-outs of the same chart (the recording stays in the driver after cairo_scene)
    cd = po.cairo()
    if cd is not None:
        from golden_io import within
        po.cairo_scene([(sty, ways)], DPI, 900, 700, 0xfff8e0)
        sub = np.zeros((r + 2) * (r + 2), np.uint32)
        fr = []
        for k in range(n):
            assert cd.cd_cut_out(cx[k] * 72 / DPI, cy[k] * 72 / DPI, r_pt, float(DPI), po.P(sub), len(sub)) == r
            ref_cut = sub[:r * r].reshape(r, r)
            got = img[k * at.window_pitch:k * at.window_pitch + r]
            fr.append(within(got, ref_cut))
        assert np.mean(fr) >= 0.995 and min(fr) >= 0.99, (name, fr)       # the north star's bar over all windows (each is 93 x 93 px only)
        if name == "polygon strokes":
            assert np.mean(fr) >= 0.9995, fr                              # as polygons the strokes are libcairo's but for a row here and there
    for k in range(n):
        c = L.orc_canvas_new(0, 0, r, r, DPI, 0xfff8e0)
        tx, ty = -cx[k] * 72 / DPI + r_pt / 2, -cy[k] * 72 / DPI + r_pt / 2
        L.orc_canvas_set_shift(c, tx, ty)
        L.orc_rule_begin(c)
        for wx, wy, closed in ways:
            wx = np.ascontiguousarray(wx, np.float64); wy = np.ascontiguousarray(wy, np.float64)
            if wx.max() < cx[k] - r / 2 - reach or wx.min() > cx[k] + r / 2 + reach or wy.max() < cy[k] - r / 2 - reach or wy.min() > cy[k] + r / 2 + reach:
                continue
            L.orc_draw_way(c, C.byref(sty), po.P(wx), po.P(wy), len(wx), int(closed), 0, 0, 0, 0)
        L.orc_rule_end(c)
        want = np.ctypeslib.as_array(C.cast(L.orc_canvas_pixels(c), C.POINTER(C.c_uint32)), (r, r)).copy()
        L.orc_canvas_free(c)
        got = img[k * at.window_pitch:k * at.window_pitch + r]
        # outside the chart the window is transparent; inside, the oracle's canvas started as background
        inside = np.zeros((r, r), bool)
        x0, y0 = -tx * DPI / 72, -ty * DPI / 72                                  # the window's corner on the chart, px
        xs, ys = np.arange(r) + x0, np.arange(r) + y0
        inside[np.ix_((ys >= 1) & (ys + 1 <= 699), (xs >= 1) & (xs + 1 <= 899))] = True
        if os.environ.get("SMR_TEST_DUMP") and not np.array_equal(got[inside], want[inside]):
            np.savez(os.path.join(os.environ["SMR_TEST_DUMP"], f"win_{name.split()[0]}_{k}.npz"), got=got, want=want, inside=inside, cx=cx[k], cy=cy[k])
        assert len(np.unique(got[inside])) > 3
        if name == "curve":
            # measured: 0 to 3 pixels of a window one level off at the window's edge rows (the limits of a decomposed curve)
            d = po.chan_diff(got, want); d = d.max(axis=-1) if d.ndim == 3 else d
            print("curve window", k, "pixels differing from the oracle:", int((d[inside] > 0).sum()), "max", int(d[inside].max()))
            assert (d[inside] > 0).sum() <= 4 and d[inside].max() <= 1, (k, int((d[inside] > 0).sum()), int(d[inside].max()))
            continue
        assert np.array_equal(got[inside], want[inside]), (k, int((got[inside] != want[inside]).sum()))
    assert np.all(img[r:at.window_pitch] == 0)                                  # the gutter rows below a window stay clear


def test_windows_at_whole_pixel_offsets_are_crops_of_the_chart(hip):
    """size-independent property, for a directional rule (one compound fill per window) after a bordered one with per-way
    colours: when anchor and r/2 are whole pixels the replay's shift is a whole number of pixels and a window inside the chart
    is the chart's own pixels -- up to the anti-aliasing of the rows where libcairo, limiting the polygons to the window instead
    of the chart, finds a vertex more or less"""
    sa = hip
    from golden_io import draw_style
    Wc, Hc = 1000, 800
    D = sa.DeviceArray.from_numpy
    x, y, off, ref = po.synth_ways(900, Wc, Hc, seed=5)
    rng = np.random.default_rng(8)
    fcol = rng.integers(0, 1 << 24, len(off) - 1).astype(np.uint32)
    bcol = rng.integers(0, 1 << 24, len(off) - 1).astype(np.uint32)
    rules = [(draw_style(sa, po.Style(1, 0, 0x90c0e0, 0, 0, 0, 0), directional=1), po.synth_ways(60, Wc, Hc, seed=6), None, None),
             (draw_style(sa, po.Style(1, 1, 0x4080c0, 0, 0, 0.15, 0)), (x, y, off, ref), fcol, bcol)]
    cv = sa.Canvas(Wc, Hc, DPI, 0xfff8e0)
    n, r = 12, 120
    cx, cy = rng.integers(r, Wc - r, n).astype(float), rng.integers(r, Hc - r, n).astype(float)
    at = sa.Canvas.windows(Wc, Hc, DPI, 0xfff8e0, cx, cy, r)
    for st, (ax, ay, ao, ar), fc, bc in rules:
        args = (st, D(ax), D(ay), D(ao.astype(np.int64)), D(ar.astype(np.int32)), len(ao) - 1, len(ar), D(fc) if fc is not None else None,
                D(bc) if bc is not None else None)
        cv.draw(*args)
        at.draw_windows(*args)
    chart, img = cv.read(), at.read()
    cv.close(); at.close()
    same = tot = 0
    for k in range(n):
        got = img[k * at.window_pitch:k * at.window_pitch + r]
        want = chart[int(cy[k]) - r // 2:int(cy[k]) + r // 2, int(cx[k]) - r // 2:int(cx[k]) + r // 2]
        assert len(np.unique(want)) > 3
        d = po.chan_diff(got, want)
        d = d.max(axis=-1) if d.ndim == 3 else d
        assert d.max() <= 8, (k, int(d.max()))      # (the sub-sampled rows of the cuts at the sides of the image libcairo replays into: a few levels)
        same += int((d == 0).sum()); tot += d.size
    assert same / tot >= 0.995, same / tot


def test_a_long_curved_segment_reaches_a_window_its_nodes_box_misses(hip):
    """curve=1: a Bezier bulges out of its way's node box by up to curve_factor times the segment's length (control_points,
    src/bspline_ctrl.c:64-105).  A closed way of four nodes 700 px apart with curve_factor 0.5 bulges some 180 px beyond its box: a
    window put on the bulge, outside the box by more than the stroke's reach, must still hold the curve (the window's pairs
    are found from the box grown by that share), like the oracle's shifted render of the same way."""
    sa = hip
    from golden_io import draw_style
    L = po.orc()
    L.orc_canvas_set_shift.argtypes = [C.c_void_p, C.c_double, C.c_double]
    Wc, Hc = 1200, 1000
    wx = np.array([250.0, 950.0, 950.0, 250.0, 250.0]); wy = np.array([200.0, 200.0, 800.0, 800.0, 200.0])
    sty = po.Style(1, 1, 0x4080c0, 0x000000, 0, 0.2, 0).with_curve(0.5)
    X, Y = wx[:4].copy(), wy[:4].copy()
    OFF, REF = np.array([0, 5], np.int64), np.array([0, 1, 2, 3, 0], np.int32)
    r = 61
    # (measured on the oracle: the shape reaches y = 12 .. 988 at x = 600 and x = 88 .. 1111 at y = 500, i.e. 160-190 px beyond the box)
    cx, cy = np.array([600.3, 1111.2, 600.7]), np.array([12.4, 500.6, 988.3])      # on the bulge above, right of and below the box
    D = sa.DeviceArray.from_numpy
    at = sa.Canvas.windows(Wc, Hc, DPI, 0xfff8e0, cx, cy, r)
    at.draw_windows(draw_style(sa, sty), D(X), D(Y), D(OFF), D(REF), 1, 5)
    img = at.read()
    at.close()
    r_pt = r * 72 / DPI
    for k in range(3):
        c = L.orc_canvas_new(0, 0, r, r, DPI, 0xfff8e0)
        L.orc_canvas_set_shift(c, -cx[k] * 72 / DPI + r_pt / 2, -cy[k] * 72 / DPI + r_pt / 2)
        L.orc_rule_begin(c)
        L.orc_draw_way(c, C.byref(sty), po.P(wx), po.P(wy), 5, 1, 0, 0, 0, 0)
        L.orc_rule_end(c)
        want = np.ctypeslib.as_array(C.cast(L.orc_canvas_pixels(c), C.POINTER(C.c_uint32)), (r, r)).copy()
        L.orc_canvas_free(c)
        got = img[k * at.window_pitch:k * at.window_pitch + r]
        assert len(np.unique(want)) > 3, k                    # the bulge does cross the window
        # (outside the chart a window is transparent, the oracle's canvas background: the chart's own pixels only)
        x0, y0 = cx[k] - r / 2, cy[k] - r / 2
        xs, ys = np.arange(r) + x0, np.arange(r) + y0
        inside = np.zeros((r, r), bool)
        inside[np.ix_((ys >= 1) & (ys + 1 <= Hc - 1), (xs >= 1) & (xs + 1 <= Wc - 1))] = True
        d = po.chan_diff(got, want); d = d.max(axis=-1) if d.ndim == 3 else d
        assert len(np.unique(got[inside])) > 3, k
        assert (d[inside] > 0).sum() <= 4 and d[inside].max() <= 1, (k, int((d[inside] > 0).sum()), int(d[inside].max()))


@pytest.mark.parametrize("bg", [0xfff8e0, 0x30c0d8f0])
@pytest.mark.parametrize("dpi", [300, 127])
def test_the_background_of_a_window_is_the_draw_passs_rectangle(hip, bg, dpi, monkeypatch):
    """the chart's background reaches the windows through a closed form of what the scan converter makes of the chart's
    rectangle (k_win_bgfill) instead of a draw pass per atlas: both must give the same pixels, in windows inside the chart, over
    each of its edges and corners, beyond it altogether, with whole-pixel and fractional anchors, opaque and translucent"""
    sa = hip
    Wc, Hc, r = 500, 400, 77
    rng = np.random.default_rng(5)
    cx = np.concatenate([rng.uniform(-30, Wc + 30, 40), [0.0, 0.0, Wc, Wc, 250.0, 250.5, -100.0, 38.5, Wc - 38.5, 10.25]])
    cy = np.concatenate([rng.uniform(-30, Hc + 30, 40), [0.0, Hc, 0.0, Hc, 200.0, 38.5, 200.0, Hc - 38.5, 12.75, Hc + 20.0]])
    at = sa.Canvas.windows(Wc, Hc, dpi, bg, cx, cy, r)
    got = at.read()
    at.reset_windows(cx[::-1].copy(), cy[::-1].copy(), r)           # and again through smr_hip_windows_reset, other anchors
    got_rev = at.read()
    at.close()
    monkeypatch.setenv("SMR_HIP_WINBG", "draw")
    at = sa.Canvas.windows(Wc, Hc, dpi, bg, cx, cy, r)
    want = at.read()
    at.close()
    assert np.array_equal(got, want), int((got != want).sum())
    assert len(np.unique(want)) > 20                                 # anti-aliased edges are in the picture
    p = at.window_pitch
    for k in range(len(cx)):
        assert np.array_equal(got_rev[(len(cx) - 1 - k) * p:(len(cx) - 1 - k) * p + r], want[k * p:k * p + r]), k


def test_the_first_drawing_into_every_windows_group_is_a_source(hip):
    """every window is a replay of its own for libcairo: the group of a rule starts clear in each, and the first drawing into a
    clear group is a SOURCE (a lerp also for a translucent colour; _fill_xrgb32_lerp_opaque_spans) -- the first primitive PER
    WINDOW, not of the atlas.  Found by tools/soak_windows.py: a closed way in the second window, an open one in the first, a
    translucent rule; before, the closed way's edge pixels came out one level off (OVER through the mask instead of the lerp)."""
    sa = hip
    from golden_io import draw_style
    L = po.orc()
    L.orc_canvas_set_shift.argtypes = [C.c_void_p, C.c_double, C.c_double]
    Wc, Hc, dpi, r, bg = 1256, 1013, 127, 239, 0xfff8e0
    cx, cy = np.array([349.5, 264.5]), np.array([914.0, 426.5])
    ax = np.array([307.1593, 306.0383, 302.0327, 298.4818, 294.6616, 290.7524, 288.2067, 286.1201, 281.3273, 284.1586, 280.3422, 279.8664, 281.8816, 280.5983, 281.4388, 283.5344, 282.5431, 284.1160, 286.7103, 291.2773, 293.5132, 297.5922, 300.8873, 303.9904, 306.6335, 309.3580, 311.9903, 317.8219, 318.7370, 321.1657, 322.1108, 319.3787, 322.0257, 319.1949, 318.1189, 316.2453, 310.4598, 307.1593])
    ay = np.array([397.7596, 391.7419, 392.7540, 389.8392, 390.8512, 391.3715, 394.8115, 397.9161, 398.2854, 404.4699, 406.1341, 409.4710, 412.8042, 415.9192, 418.9497, 421.3711, 425.8741, 429.3529, 432.1829, 431.2250, 436.1677, 434.9870, 431.0384, 430.8468, 429.1743, 427.8329, 426.1707, 426.4864, 422.6017, 419.4435, 415.6254, 411.8609, 407.6801, 404.5356, 400.8089, 397.3022, 398.3091, 397.7596])
    bx = np.array([447.8369, 453.1327, 460.3654, 463.7900, 469.3191, 472.1237, 475.7550, 474.0965, 477.8386, 472.2102, 468.8052, 465.8624, 459.6840, 452.8568, 445.5404, 440.1109, 435.3822, 430.4925, 428.6132, 431.3680, 432.4661, 436.5233, 440.2011])
    by = np.array([904.3116, 910.9756, 910.4231, 901.5194, 899.7533, 894.7728, 890.0992, 884.2373, 877.7225, 873.5788, 869.0940, 862.5129, 859.6443, 859.6478, 859.1206, 863.9986, 868.8864, 874.0788, 881.0042, 887.7871, 894.2718, 899.4489, 905.3345])
    ways = [(ax, ay, 1), (bx, by, 0)]
    D = sa.DeviceArray.from_numpy
    m = len(ax) - 1
    X, Y = np.concatenate([ax[:m], bx]), np.concatenate([ay[:m], by])
    REF = np.concatenate([np.arange(m), [0], m + np.arange(len(bx))]).astype(np.int32)
    OFF = np.array([0, m + 1, m + 1 + len(bx)], np.int64)
    r_pt = r * 72 / dpi
    for sty in (po.Style(1, 1, 0xc0eb4172, 0x9b01a2c5, 0.0, 0.244, 0), po.Style(1, 0, 0xc0eb4172, 0, 0, 0, 0), po.Style(0, 1, 0, 0x9b01a2c5, 0.0, 0.244, 0)):
        at = sa.Canvas.windows(Wc, Hc, dpi, bg, cx, cy, r)
        at.draw_windows(draw_style(sa, sty), D(X), D(Y), D(OFF), D(REF), 2, len(REF))
        img = at.read()
        at.close()
        for k in range(2):
            c = L.orc_canvas_new(0, 0, r, r, dpi, bg)
            L.orc_canvas_set_shift(c, -cx[k] * 72 / dpi + r_pt / 2, -cy[k] * 72 / dpi + r_pt / 2)
            L.orc_rule_begin(c)
            for wx, wy, closed in ways:
                L.orc_draw_way(c, C.byref(sty), po.P(np.ascontiguousarray(wx)), po.P(np.ascontiguousarray(wy)), len(wx), closed, 0, 0, 0, 0)
            L.orc_rule_end(c)
            want = np.ctypeslib.as_array(C.cast(L.orc_canvas_pixels(c), C.POINTER(C.c_uint32)), (r, r)).copy()
            L.orc_canvas_free(c)
            got = img[k * at.window_pitch:k * at.window_pitch + r]
            ys = np.arange(r) + cy[k] - r / 2
            inside = (ys >= 1) & (ys + 1 <= Hc - 1)                      # the first window hangs over the chart's bottom
            assert len(np.unique(got[inside])) > 3, k
            assert np.array_equal(got[inside], want[inside]), (k, int((got[inside] != want[inside]).sum()))


def test_stroked_windows_equal_the_oracle_at_the_atlas_rows(hip):
    """windows beyond an atlas's first, strokes of several kinds, bit for bit against the oracle rendering window k where the atlas
    holds it (orc_canvas_move_rows: canvas and shift moved down by k * pitch rows).  The stroker works in doubles on device
    coordinates, and a join's vertex now and then rounds the other way some hundred rows further down: against the oracle at rows
    0 one window in twenty-five is one level off in a pixel or two (tools/soak_windows.py), against the oracle at
    the atlas's rows none is."""
    sa = hip
    from golden_io import draw_style
    L = po.orc()
    L.orc_canvas_set_shift.argtypes = [C.c_void_p, C.c_double, C.c_double]
    L.orc_canvas_move_rows.argtypes = [C.c_void_p, C.c_int]
    Wc, Hc, dpi, bg = 1048, 474, 300, 0xfff8e0
    x, y, off, ref = po.synth_ways(900, Wc, Hc, seed=31)
    ways = po.ways_from_arrays(x, y, off, ref)
    box = np.array([[w[0].min(), w[0].max(), w[1].min(), w[1].max()] for w in ways])
    D = sa.DeviceArray.from_numpy
    rng = np.random.default_rng(12)
    n, r = 10, 180
    r_pt = r * 72 / dpi
    for sty in (po.Style(0, 1, 0, 0xb9acdd, 0.46, 0.186, 0), po.Style(1, 1, 0x4080c0, 0x102030, 0, 0.3, 0), po.Style(1, 1, 0x604080c0, 0x80102030, 0, 0.25, 0)):
        cx, cy = rng.uniform(60, Wc - 60, n), rng.uniform(60, Hc - 60, n)
        at = sa.Canvas.windows(Wc, Hc, dpi, bg, cx, cy, r)
        at.draw_windows(draw_style(sa, sty), D(x), D(y), D(off.astype(np.int64)), D(ref.astype(np.int32)), len(off) - 1, len(ref))
        img = at.read()
        pitch = at.window_pitch
        at.close()
        for k in range(n):
            c = L.orc_canvas_new(0, 0, r, r, dpi, bg)
            tx, ty = -cx[k] * 72 / dpi + r_pt / 2, -cy[k] * 72 / dpi + r_pt / 2
            L.orc_canvas_set_shift(c, tx, ty)
            L.orc_canvas_move_rows(c, k * pitch)
            L.orc_rule_begin(c)
            m = r / 2 + 60
            for i in np.nonzero(~((box[:, 1] < cx[k] - m) | (box[:, 0] > cx[k] + m) | (box[:, 3] < cy[k] - m) | (box[:, 2] > cy[k] + m)))[0]:
                wx, wy, closed = ways[i]
                L.orc_draw_way(c, C.byref(sty), po.P(np.ascontiguousarray(wx, np.float64)), po.P(np.ascontiguousarray(wy, np.float64)), len(wx), int(closed), 0, 0, 0, 0)
            L.orc_rule_end(c)
            want = np.ctypeslib.as_array(C.cast(L.orc_canvas_pixels(c), C.POINTER(C.c_uint32)), (r, r)).copy()
            L.orc_canvas_free(c)
            got = img[k * pitch:k * pitch + r]
            xs, ys = np.arange(r) + cx[k] - r / 2, np.arange(r) + cy[k] - r / 2
            inside = np.zeros((r, r), bool)
            inside[np.ix_((ys >= 1) & (ys + 1 <= Hc - 1), (xs >= 1) & (xs + 1 <= Wc - 1))] = True
            assert len(np.unique(got[inside])) > 3, k
            assert np.array_equal(got[inside], want[inside]), (k, int((got[inside] != want[inside]).sum()))
