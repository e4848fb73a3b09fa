"""Kernel 2 (edge setup, stroke expansion, binning, wave-per-tile rasterizer) through the
C-ABI vs the oracle restatement of act_draw + libcairo's tor scan converter.  Integer
arithmetic end to end, so the bar is bit-exact ARGB."""
import ctypes as C
import numpy as np
import pytest
from oracle import pyoracle as po
from golden_io import draw_style
from smrender_amd import synth

pytestmark = pytest.mark.gpu


def gpu_scene(sa, rules, dpi, W, H, bg, band_y0=0, band_h=None):
    cv = sa.Canvas(W, H, dpi, bg, band_y0, band_h)
    for st, (x, y, way_off, ref) in rules:
        gs = draw_style(sa, st)
        d_x, d_y = sa.DeviceArray.from_numpy(x), sa.DeviceArray.from_numpy(y)
        d_off = sa.DeviceArray.from_numpy(way_off.astype(np.int64))
        d_ref = sa.DeviceArray.from_numpy(ref.astype(np.int32))
        cv.draw(gs, d_x, d_y, d_off, d_ref, len(way_off) - 1, len(ref))
    out = cv.read()
    cv.close()
    return out


def oracle_of(rules, dpi, W, H, bg, y0=0, h=None):
    orules = [(st, po.ways_from_arrays(*arrs)) for st, arrs in rules]
    return po.oracle_scene(orules, dpi, W, H if h is None else h, bg, 0, y0)


def report(a, b):
    d = po.chan_diff(a, b)
    return f"mismatch px {(d > 0).sum()} of {d.size}, max {d.max()}, first at {np.argwhere(d > 0)[:5].tolist()}"


STYLES = {
    "fill+border": po.Style(1, 1, 0x4080c0, 0x000000, 0, 25.4 / 300, 0),
    "fill": po.Style(1, 0, 0xc04020, 0, 0, 0, 0),
    "border": po.Style(0, 1, 0, 0x204060, 0, 0.3, 0),
    "hairline": po.Style(0, 1, 0, 0x000000, 0, 0.0, 0),
    "translucent": po.Style(1, 1, 0x304080c0, 0x50102030, 0, 0.2, 0),
}


@pytest.mark.parametrize("style", list(STYLES))
@pytest.mark.parametrize("n_ways,size", [(1, 64), (40, 200), (300, 333)])
def test_closed_ways_bit_exact(hip, style, n_ways, size):
    st = STYLES[style]
    arrs = po.synth_ways(n_ways, size, size, seed=n_ways + len(style))
    rules = [(st, arrs)]
    g = gpu_scene(hip, rules, 300, size, size, 0xffffff)
    o = oracle_of(rules, 300, size, size, 0xffffff)
    assert np.array_equal(g, o), report(g, o)


def _open_ways(n_ways, W, H, seed):
    rng = np.random.default_rng(seed)
    xs, ys, off, ref = [], [], [0], []
    for w in range(n_ways):
        n = int(rng.integers(2, 14))
        x0, y0 = rng.uniform(0, W), rng.uniform(0, H)
        step = rng.uniform(2, 30)
        xs.append(x0 + np.cumsum(rng.normal(0, step, n)))
        ys.append(y0 + np.cumsum(rng.normal(0, step, n)))
        ref.extend(range(off[-1], off[-1] + n))
        off.append(off[-1] + n)
    return np.concatenate(xs), np.concatenate(ys), np.array(off, np.int64), np.array(ref, np.int32)


@pytest.mark.parametrize("style", ["border", "hairline", "translucent", "fill+border"])
def test_open_ways_bit_exact(hip, style):
    st = STYLES[style]
    if style == "fill+border":
        st = po.Style(1, 1, 0xe0c000, 0x000000, 0.4, 0.15, 0)
    arrs = _open_ways(60, 300, 220, seed=5)
    rules = [(st, arrs)]
    g = gpu_scene(hip, rules, 300, 300, 220, 0xffffff)
    o = oracle_of(rules, 300, 300, 220, 0xffffff)
    assert np.array_equal(g, o), report(g, o)


def test_two_rules_layers_and_600dpi(hip):
    a1 = po.synth_ways(80, 400, 260, seed=11)
    a2 = _open_ways(40, 400, 260, seed=12)
    rules = [(STYLES["fill+border"], a1), (STYLES["translucent"], a2), (STYLES["border"], a1)]
    g = gpu_scene(hip, rules, 600, 400, 260, 0x7f336699)
    o = oracle_of(rules, 600, 400, 260, 0x7f336699)
    assert np.array_equal(g, o), report(g, o)


def test_edge_cases(hip):
    sa = hip
    # empty batch, short ways (n<2 skipped: smrules_cairo.c:962-966), 3-ref "closed" way is open (smcoast.c:115-125),
    # degenerate (repeated) nodes, ways partly / fully outside the canvas
    x = np.array([10.0, 50, 30, 10, 5, 5, 5, 60, 60.0, 90, 70, 70, -40, 30, 500, 520, 510])
    y = np.array([10.0, 12, 40, 10, 5, 5, 5, 60, 60.0, 20, 35, 35, 20, -30, 500, 530, 515])
    ref = np.array([0, 1, 2, 0,   3, 4, 3,   5,   7, 8, 8, 9, 10, 10, 11,   12, 13, 1, 12,  14, 15, 16, 14], np.int32)
    off = np.array([0, 4, 7, 8, 15, 19, 23], np.int64)
    for st in (STYLES["fill+border"], STYLES["border"], po.Style(1, 1, 0x4080c0, 0, 0.5, 0.1, 1)):
        rules = [(st, (x, y, off, ref))]
        g = gpu_scene(sa, rules, 300, 128, 100, 0xffffff)
        o = oracle_of(rules, 300, 128, 100, 0xffffff)
        assert np.array_equal(g, o), report(g, o)
    cv = sa.Canvas(64, 64, 300, 0xffffff)
    gs = sa.DrawStyle(1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
    d = sa.DeviceArray((4,), np.float64)
    cv.draw(gs, d, d, sa.DeviceArray.from_numpy(np.zeros(1, np.int64)), None, 0, 0)      # nothing to draw
    assert np.all(cv.read() == 0xffffffff)
    gs2 = sa.DrawStyle(1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
    gs2.curve = 3                                       # an unknown curve mode fails loudly
    with pytest.raises(sa.SmrHipError) as e:
        cv.draw(gs2, d, d, sa.DeviceArray.from_numpy(np.array([0, 4], np.int64)), None, 1, 4)
    assert e.value.code == -2
    cv.close()


# ---- curve=1 (cairo_smr_poly_curve, src/smrules_cairo.c:697-749): Bezier chain, libcairo's spline decomposition,
# faces along the tangents.  The control points go through atan2/cos/sin/hypot on the device, whose last bit may
# differ from glibc's: the decomposition is integer from there on, so the bar stays bit-exact and a failure here
# with a handful of pixels would point at that.
def _inside(arrs, sx, sy, dx, dy):
    return (arrs[0] * sx + dx, arrs[1] * sy + dy, arrs[2], arrs[3])


CURVE_STYLES = {
    "fill": po.Style(1, 0, 0x4080c0, 0, 0, 0, 0),
    "border": po.Style(0, 1, 0, 0x000000, 0, 0.3, 0),
    "fill+border": po.Style(1, 1, 0x4080c0, 0x000000, 0, 0.2, 0),
    "hairline": po.Style(0, 1, 0, 0x000000, 0, 0.0, 0),
    "translucent": po.Style(1, 1, 0x304080c0, 0x50102030, 0.4, 0.15, 0),
}


@pytest.mark.parametrize("style", list(CURVE_STYLES))
@pytest.mark.parametrize("scene", ["closed_inside", "closed_crossing", "open"])
def test_curves_bit_exact(hip, style, scene):
    st = CURVE_STYLES[style].with_curve(0.35 if style == "translucent" else 0.2)
    if scene == "closed_inside":
        arrs = _inside(po.synth_ways(12, 200, 150, seed=5, rmin=10.0, rmax=40.0), 0.6, 0.6, 50, 40)
    elif scene == "closed_crossing":       # ways leaving the chart: curves off the image are lines (libcairo's limits)
        arrs = po.synth_ways(40, 256, 200, seed=6, rmin=10.0, rmax=60.0)
    else:
        arrs = _open_ways(30, 256, 200, seed=7)
    rules = [(st, arrs)]
    g = gpu_scene(hip, rules, 300, 256, 200, 0xffffff)
    o = oracle_of(rules, 300, 256, 200, 0xffffff)
    assert np.array_equal(g, o), report(g, o)


def test_curves_600dpi_many_ways_and_bands(hip):
    st = po.Style(1, 1, 0x4080c0, 0x000000, 0, 25.4 / 300, 0).with_curve()
    arrs = po.synth_ways(3000, 1024, 1024, seed=8)
    rules = [(st, arrs)]
    full = gpu_scene(hip, rules, 600, 1024, 1024, 0xffffff)
    o = oracle_of(rules, 600, 1024, 1024, 0xffffff)
    assert np.array_equal(full, o), report(full, o)
    # a band renders its rows of the full chart: the curve limits are the chart's, not the band's
    band = gpu_scene(hip, rules, 600, 1024, 1024, 0xffffff, 300, 200)
    assert np.array_equal(band, full[300:500])


def test_curve_refusals(hip):
    sa = hip
    cv = sa.Canvas(64, 64, 300, 0xffffff)
    d = sa.DeviceArray.from_numpy(np.array([5.0, 20, 40, 50]))
    off = sa.DeviceArray.from_numpy(np.array([0, 4], np.int64))
    gs = sa.DrawStyle(0, 1, 0, 0, 0, 0.2, 0, 0, 0, 0, 0, 0).set_curve(0.0)   # curve_factor must be positive
    with pytest.raises(sa.SmrHipError) as e:
        cv.draw(gs, d, d, off, None, 1, 4)
    assert e.value.code == -2
    cv.close()


@pytest.mark.parametrize("case", ["stroke", "curve_stroke", "fill_border", "zigzag"])
def test_long_primitives_bit_exact(hip, case):
    """ways of thousands of nodes (contour lines, coastlines): their vertex-free rows are analysed window by window by
    k_big_rows, and a stroke's full rows are built from the items binned to the tile alone"""
    from smrender_amd import synth
    W, H = 1024, 1024
    land, rings = synth.synth_coast(6000, 0, 2, W, H, seed=5)
    if case == "stroke":
        rules = [(po.Style(0, 1, 0, 0x204060, 0, 0.25, 0), rings)]
    elif case == "curve_stroke":
        rules = [(po.Style(0, 1, 0, 0x204060, 0, 0.25, 0).with_curve(), rings)]
    elif case == "fill_border":
        rules = [(po.Style(1, 1, 0xf0e0a0, 0x000000, 0, 0.1, 0), land)]
    else:
        # long steep segments, 3 px wide, across several tile columns: vertex-free rows in a stroke of many items
        n = 24          # 46 edges span the middle rows: under the 64 the exact whole-row path keeps (beyond: sub-sampled + reported)
        x = 100.0 + 20.0 * np.arange(n) + np.where(np.arange(n) % 2, 7.3, 0.0)
        y = np.where(np.arange(n) % 2, 900.4, 80.7) + 3.1 * np.arange(n)
        rules = [(po.Style(0, 1, 0, 0x000000, 0, 0.254, 0), (x, y, np.array([0, n], np.int64), np.arange(n, dtype=np.int32)))]
    g = gpu_scene(hip, rules, 300, W, H, 0xffffff)
    o = oracle_of(rules, 300, W, H, 0xffffff)
    assert np.array_equal(g, o), report(g, o)


def test_bands_stitch_to_full_canvas(hip):
    """multi-GPU layout: each rank owns a band of rows; bands must stitch bit-identically"""
    arrs = po.synth_ways(150, 320, 301, seed=21)
    rules = [(STYLES["fill+border"], arrs)]
    full = gpu_scene(hip, rules, 300, 320, 301, 0xffffff)
    parts = []
    bounds = [0, 77, 150, 222, 301]
    for a, b in zip(bounds[:-1], bounds[1:]):
        parts.append(gpu_scene(hip, rules, 300, 320, 301, 0xffffff, a, b - a))
    assert np.array_equal(np.concatenate(parts, 0), full)


def test_dense_scene_vs_oracle(hip):
    arrs = po.synth_ways(2500, 512, 512, seed=31)
    rules = [(STYLES["fill+border"], arrs)]
    g = gpu_scene(hip, rules, 300, 512, 512, 0xffffff)
    o = oracle_of(rules, 300, 512, 512, 0xffffff)
    assert np.array_equal(g, o), report(g, o)


def _ways_to_arrays(ways):
    xs, ys, off, ref = [], [], [0], []
    n = 0
    for x, y, closed in ways:
        k = len(x) - 1 if closed else len(x)
        xs.append(np.asarray(x[:k], np.float64)); ys.append(np.asarray(y[:k], np.float64))
        ref.extend(range(n, n + k))
        if closed:
            ref.append(n)
        n += k
        off.append(len(ref))
    return np.concatenate(xs), np.concatenate(ys), np.array(off, np.int64), np.array(ref, np.int32)


@pytest.mark.parametrize("col,size", [(0xc0d8f0, (200, 160)), (0x30c0d8f0, (200, 160)), (0x4080c0, (700, 333))])
def test_directional_compound_fill(hip, col, size):
    """directional=1 (act_draw_fini :1084-1140): all closed ways of the rule are one non-zero fill, open ways
    are drawn first; holes and islands follow from the orientation of the ways"""
    from test_oracle_cairo import _nested_scene
    sa = hip
    W, H = size
    ways = _nested_scene(W, H, 3 + W, n_small=40 if W < 300 else 400)
    st = po.Style(1, 0, col, 0, 0.3, 0, 0)
    want = po.oracle_directional(st, ways, 300, W, H, 0xffffff)
    x, y, off, ref = _ways_to_arrays(ways)
    cv = sa.Canvas(W, H, 300, 0xffffff)
    gs = sa.DrawStyle(1, 1, col, 0x000000, 0.3, 0.2, 0, 0, 0, 0, 1, 0)       # the border is ignored in directional mode (:674-680)
    cv.draw(gs, sa.DeviceArray.from_numpy(x), sa.DeviceArray.from_numpy(y), sa.DeviceArray.from_numpy(off),
            sa.DeviceArray.from_numpy(ref), len(off) - 1, len(ref))
    got = cv.read()
    cv.close()
    assert np.array_equal(got, want), report(got, want)


def test_rows_beyond_the_exact_path_are_subsampled_and_reported(hip):
    """A comb with 45 slanted teeth: 90 edges span vertex-free pixel rows, more than the exact whole-row path
    keeps.  The library sub-samples such rows (libcairo has no such limit), says so through
    smr_hip_draw_status(), and stays within a few levels of the oracle; rows with vertices are exact."""
    sa = hip
    W, H = 420, 90
    xs, ys = [], []
    for k in range(45):
        x0 = 10.3 + k * 8.8
        xs += [x0, x0 + 6.1, x0 + 9.3, x0 + 5.7]; ys += [70.2, 12.4, 12.4, 70.2]
    xs += [W - 8.0, 10.3]; ys += [80.5, 80.5]
    x, y = np.array(xs), np.array(ys)
    n = len(x)
    off, ref = np.array([0, n + 1], np.int64), np.append(np.arange(n), 0).astype(np.int32)
    st = po.Style(1, 0, 0x204060, 0, 0, 0, 0)
    want = oracle_of([(st, (x, y, off, ref))], 300, W, H, 0xffffff)
    got = gpu_scene(sa, [(st, (x, y, off, ref))], 300, W, H, 0xffffff)
    assert sa.lib().smr_hip_draw_status() & 1
    d = po.chan_diff(got, want)
    assert 0 < d.max() <= 4 and (d > 0).mean() < 0.2, report(got, want)
    assert d[:13].max() == 0 and d[70:].max() == 0        # rows with vertices / with two spanning edges only
    # an ordinary scene afterwards clears the flag
    gpu_scene(sa, [(st, po.synth_ways(5, 64, 64, seed=1))], 300, 64, 64, 0xffffff)
    assert sa.lib().smr_hip_draw_status() == 0


def test_soak_scene_with_a_wide_curved_stroke(hip):
    """The first scene of tools/soak.py's random runs that differed from the oracle (tests/golden/soak_scene4.npz, 31 ways at
    600 dpi, curve=1): way 9's border is an 11.8 px wide stroke of a tightly curved line whose flattened pieces put 86 edges
    across the vertex-free pixel row 133.  The row analysis of a stroke compares up to SMR_SCANCAP (576) spanning edges, and a
    tile whose list of one row fills up keeps only the pieces near it: exact, and nothing to report."""
    sa = hip
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "soak_scene4.npz"))
    x, y, off, ref = d["x"], d["y"], d["off"], d["ref"]
    W, H, dpi = int(d["W"]), int(d["H"]), int(d["dpi"])
    st = po.Style.from_saved(d["style"])
    want = oracle_of([(st, (x, y, off, ref))], dpi, W, H, 0xffffff)
    got = gpu_scene(sa, [(st, (x, y, off, ref))], dpi, W, H, 0xffffff)
    assert np.array_equal(got, want), report(got, want)
    assert sa.lib().smr_hip_draw_status() == 0


def test_soak_scene_whose_tiles_overflow_runs_again_with_reported_rows(hip):
    """tests/golden/soak_scene1_37.npz (900 dpi, a 39 px wide curved stroke): vertex-free rows with hundreds of spanning edges
    pass the row analysis, but one 64-px tile would have to hold more than SMR_ROWCAP of them.  draw_pass restores the canvas
    from its copy and runs again with such rows sub-sampled: no error, the rows are reported, the picture stays within a few
    levels of the oracle on a few pixels."""
    sa = hip
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "soak_scene1_37.npz"))
    x, y, off, ref = d["x"], d["y"], d["off"], d["ref"]
    W, H, dpi = int(d["W"]), int(d["H"]), int(d["dpi"])
    st = po.Style.from_saved(d["style"])
    want = oracle_of([(st, (x, y, off, ref))], dpi, W, H, 0xffffff)
    got = gpu_scene(sa, [(st, (x, y, off, ref))], dpi, W, H, 0xffffff)
    if os.environ.get("SMR_HIP_STROKE") != "polygon":       # (as a polygon the stroke has a fraction of the pieces' spanning edges: nothing to report)
        assert sa.lib().smr_hip_draw_status() & 1
    dd = po.chan_diff(got, want)
    assert dd.max() <= 12 and (dd > 0).mean() < 0.02, report(got, want)


def test_soak_scene_with_one_pixel_dots(hip):
    """tests/golden/soak_scene21_2.npz, way 17 (tools/soak.py 21 60 dash): a dotted fill stroke (style=dotted, no border, so the
    dots are one device pixel long) at 600 dpi puts 17 dots on two sub-rows of one tile, two more than a sub-row's list in LDS
    holds.  (The interval still open at the flush used to be pushed into the full list after the overflow flag had been read, and
    was dropped without a word: one pixel at alpha 220 instead of 254.)  A tile whose lists overflow writes nothing and is rendered
    again by the second instantiation of k_raster, whose lists in global memory hold every edge slot binned to it: bit-exact, for
    the way alone and for the whole scene."""
    sa = hip
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "soak_scene21_2.npz"))
    x, y, off, ref = d["x"], d["y"], d["off"], d["ref"]
    W, H, dpi = int(d["W"]), int(d["H"]), int(d["dpi"])
    st = po.Style.from_saved(d["style"])
    r1 = np.ascontiguousarray(ref[off[17]:off[18]]); o1 = np.array([0, len(r1)], np.int64)
    for arrs in ((x, y, o1, r1), (x, y, off, ref)):
        want = oracle_of([(st, arrs)], dpi, W, H, 0xffffff)
        got = gpu_scene(sa, [(st, arrs)], dpi, W, H, 0xffffff)
        assert np.array_equal(got, want), report(got, want)


def _comb(teeth, x0, pitch, width, y_top, y_bot, W):
    """a closed comb: `teeth` slanted teeth of `width` px every `pitch` px, joined by a bar below them"""
    xs, ys = [], []
    for k in range(teeth):
        a = x0 + k * pitch
        xs += [a, a + 0.35 * width, a + 1.35 * width, a + width]; ys += [y_bot, y_top, y_top, y_bot]
    xs += [x0 + teeth * pitch, x0 - 1.0]; ys += [y_bot + 6.3, y_bot + 6.3]
    x, y = np.array(xs), np.array(ys)
    n = len(x)
    return x, y, np.array([0, n + 1], np.int64), np.append(np.arange(n), 0).astype(np.int32)


@pytest.mark.parametrize("style", ["fill", "fill+border", "hairline"])
def test_tiles_with_more_crossings_than_the_lds_lists_hold_are_rendered_by_the_second_pass(hip, style):
    """26 teeth inside one 64-px tile: 52 crossings of the fill, or (16 teeth) 32 disjoint stroke intervals, on every sub-row,
    where the fast instantiation's lists hold 30 / 15.  No error, no reported row, every pixel the oracle's; the tiles beside
    (drawn by the fast instantiation) and a second, ordinary rule on top come out right as well."""
    sa = hip
    W, H = 200, 64
    # (a stroke of 26 teeth would put 104 edges across the vertex-free rows of one tile, beyond the exact path's 64: 16 teeth)
    comb = _comb(26 if style == "fill" else 16, 66.2, 2.37, 1.1, 7.3, 41.8, W)
    st = STYLES[style]
    cv = sa.Canvas(W, H, 300, 0xffffff)
    dev = lambda arrs: [sa.DeviceArray.from_numpy(a) for a in (arrs[0], arrs[1], arrs[2].astype(np.int64), arrs[3].astype(np.int32))]
    cv.draw(draw_style(sa, st), *dev(comb), 1, len(comb[3]))
    assert sa.lib().smr_hip_draw_status() == 2          # SMR_DRAW_SECOND_PASS and nothing else
    over = po.synth_ways(12, W, H, seed=3)
    cv.draw(draw_style(sa, STYLES["translucent"]), *dev(over), len(over[2]) - 1, len(over[3]))
    assert sa.lib().smr_hip_draw_status() == 0
    got = cv.read(); cv.close()
    want = oracle_of([(st, comb), (STYLES["translucent"], over)], 300, W, H, 0xffffff)
    assert np.array_equal(got, want), report(got, want)


def test_scribble_that_used_to_end_in_the_crossing_limit(hip):
    """a random walk of 400 short segments over a 40-px patch, stroked 1.3 px wide: dozens of disjoint intervals per sub-row
    (tools/soak.py ... big ended in SMR_E_CAPACITY for such ways in round 1)"""
    rng = np.random.default_rng(77)
    n = 400
    x = 100 + np.cumsum(rng.normal(0, 6, n)); y = 60 + np.cumsum(rng.normal(0, 6, n))
    x = 80 + np.abs((x - 80) % 80 - 40); y = 40 + np.abs((y - 40) % 80 - 40)           # folded back into the patch
    arrs = (x, y, np.array([0, n], np.int64), np.arange(n, dtype=np.int32))
    st = po.Style(0, 1, 0, 0x102030, 0, 0.11, 0)
    want = oracle_of([(st, arrs)], 300, 200, 140, 0xffffff)
    got = gpu_scene(hip, [(st, arrs)], 300, 200, 140, 0xffffff)
    assert np.array_equal(got, want), report(got, want)


def test_a_pass_that_cannot_be_finished_leaves_the_canvas_as_it_was(hip, monkeypatch):
    """The second pass sizes its lists for every edge slot of the fullest overflowed tile, up to 2^20 slots (a tile with 16 384
    entries); beyond, the rule fails with SMR_E_CAPACITY and the canvas is restored from the copy taken before the pass.
    SMR_HIP_DEBUG_BIG_SLOTS lowers the limit so that the comb reaches it."""
    sa = hip
    W, H = 200, 64
    cv = sa.Canvas(W, H, 300, 0xffffff)
    base = po.synth_ways(12, W, H, seed=3)
    dev = lambda arrs: [sa.DeviceArray.from_numpy(a) for a in (arrs[0], arrs[1], arrs[2].astype(np.int64), arrs[3].astype(np.int32))]
    cv.draw(draw_style(sa, STYLES["fill+border"]), *dev(base), len(base[2]) - 1, len(base[3]))
    before = cv.read()
    comb = _comb(26, 66.2, 2.37, 1.1, 7.3, 41.8, W)
    monkeypatch.setenv("SMR_HIP_DEBUG_BIG_SLOTS", "32")
    with pytest.raises(sa.SmrHipError) as e:
        cv.draw(draw_style(sa, STYLES["fill"]), *dev(comb), 1, len(comb[3]))
    assert e.value.code == -6 and "as before the rule" in str(e.value)
    assert np.array_equal(cv.read(), before)
    monkeypatch.delenv("SMR_HIP_DEBUG_BIG_SLOTS")
    cv.draw(draw_style(sa, STYLES["fill"]), *dev(comb), 1, len(comb[3]))          # and the same rule goes through without the limit
    want = oracle_of([(STYLES["fill+border"], base), (STYLES["fill"], comb)], 300, W, H, 0xffffff)
    got = cv.read(); cv.close()
    assert np.array_equal(got, want), report(got, want)


def test_soak_scene_directional_rule_with_a_way_beside_the_chart(hip):
    """tests/golden/soak_scene70_37_directional.npz (tools/soak.py 70 60 dir): a directional rule of 103 ways, one of the closed
    ones (way 77) wholly LEFT of the chart.  It draws nothing, but it is part of the rule's one polygon, and its vertices in
    row 77 make libcairo sub-sample that row; the set-up used to drop such a member with its row mask, so the row took the exact
    whole-row path (18 px off by up to 5 levels).  The closed ways must equal the oracle; the hairline strokes of the open ways
    are compared where the oracle itself equals libcairo."""
    sa = hip
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "soak_scene70_37_directional.npz"))
    x, y, off, ref = d["x"], d["y"], d["off"], d["ref"]
    W, H, dpi = int(d["W"]), int(d["H"]), int(d["dpi"])
    st = po.Style.from_saved(d["style"])
    ways = po.ways_from_arrays(x, y, off, ref)
    want = po.oracle_directional(st, ways, dpi, W, H, 0xffffff)
    cv = sa.Canvas(W, H, dpi, 0xffffff)
    cv.draw(draw_style(sa, st, directional=1), *(sa.DeviceArray.from_numpy(a) for a in (x, y, off, ref)), len(off) - 1, len(ref))
    got = cv.read(); cv.close()
    assert np.array_equal(got, want), report(got, want)


def test_soak_scene_with_a_span_that_ends_right_of_the_tile(hip):
    """tests/golden/soak_scene38_49.npz (tools/soak.py 38 60), way 35: a curved closed fill of 223 edges (four items).  In the
    vertex-free row 276 the span starts at x = 41 and ends at x = 230, at an edge of the first item, whose box lies right of
    the tile 64..127 and is not binned there.  The exact whole-row path used to close the span at the tile's last edge (the
    oracle's rule for unbalanced input, which closed paths never are): 58 px of the row were left white."""
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "soak_scene38_49.npz"))
    x, y, off, ref = d["x"], d["y"], d["off"], d["ref"]
    W, H, dpi = int(d["W"]), int(d["H"]), int(d["dpi"])
    st = po.Style.from_saved(d["style"])
    want = oracle_of([(st, (x, y, off, ref))], dpi, W, H, 0xffffff)
    got = gpu_scene(hip, [(st, (x, y, off, ref))], dpi, W, H, 0xffffff)
    assert np.array_equal(got, want), report(got, want)


@pytest.mark.parametrize("scene", ["soak_scene5_16", "soak_scene5_43"])
def test_soak_scenes_with_segments_that_run_straight_back(hip, scene):
    """Two more scenes of tools/soak.py (coordinates snapped to half pixels, curve=1): a segment runs exactly back along its
    neighbour, the helper point of control_points (src/bspline_ctrl.c:64-105) falls onto the segment's other end and the control
    point's angle is atan2 of rounding noise, i.e. decided by the last bits of the host's libm.  The library takes such control
    points from the host (smr_control_points / smr_control_points_host)."""
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", scene + ".npz"))
    x, y, off, ref = d["x"], d["y"], d["off"], d["ref"]
    W, H, dpi = int(d["W"]), int(d["H"]), int(d["dpi"])
    st = po.Style.from_saved(d["style"])
    want = oracle_of([(st, (x, y, off, ref))], dpi, W, H, 0xffffff)
    got = gpu_scene(hip, [(st, (x, y, off, ref))], dpi, W, H, 0xffffff)
    assert np.array_equal(got, want), report(got, want)


@pytest.mark.parametrize("style", ["border", "fill+border"])
def test_curved_ways_that_double_back(hip, style):
    """A B A C ...: ways with a spur, as OSM data has them; every spur makes two ill-conditioned control points"""
    rng = np.random.default_rng(31)
    xs, ys, off, ref = [], [], [0], []
    for w in range(40):
        n = int(rng.integers(3, 9))
        px = rng.uniform(10, 290) + np.cumsum(rng.normal(0, 14, n)); py = rng.uniform(10, 190) + np.cumsum(rng.normal(0, 14, n))
        k = int(rng.integers(1, n - 1))
        order = list(range(k + 1)) + [k - 1] + list(range(k + 1, n))          # ... k-1, k, k-1, k+1 ...
        base = sum(len(a) for a in xs)
        xs.append(px); ys.append(py)
        ref += [base + o for o in order]
        off.append(len(ref))
    x, y = np.concatenate(xs), np.concatenate(ys)
    arrs = (x, y, np.array(off, np.int64), np.array(ref, np.int32))
    st = CURVE_STYLES[style].with_curve(0.35)
    g = gpu_scene(hip, [(st, arrs)], 300, 300, 200, 0xffffff)
    o = oracle_of([(st, arrs)], 300, 300, 200, 0xffffff)
    assert np.array_equal(g, o), report(g, o)


@pytest.mark.parametrize("case", ["dashed", "dotted", "rounddot", "pipe", "dash4", "hairline-dashed", "fill+dashed-border",
                                  "dashed-fill-stroke", "pipe+fill-closed", "rounddot+dashed-fill"])
def test_dashed_strokes_bit_exact(hip, case):
    """style=/bstyle=/dash=/bdash= (cairo_smr_dash :898-923, the pipe's second stroke :979-986): the lane-per-way
    dasher against the oracle's restatement of libcairo's dashing stroker, bit for bit"""
    from test_oracle_cairo import DASH_CASES, _dash_scene
    S = po.Style
    extra = {"pipe+fill-closed": S(1, 1, 0x30c0d8f0, 0x103050, 0, 0.15, 1).with_dash(border_style=S.PIPE),
             "rounddot+dashed-fill": S(1, 1, 0x4080c0, 0x802010, 0.5, 0.2, 0).with_dash(border_style=S.ROUNDDOT, fill_style=S.DOTTED)}
    st = DASH_CASES.get(case) or extra[case]
    W, H = 220, 180
    ways = _dash_scene(W, H, 5)
    arrs = _ways_to_arrays(ways)
    want = po.oracle_scene([(st, ways)], 300, W, H, 0xffffff)
    got = gpu_scene(hip, [(st, arrs)], 300, W, H, 0xffffff)
    assert np.array_equal(got, want), report(got, want)
    assert (got != 0xffffffff).sum() > 2000


@pytest.mark.parametrize("case", ["rounddot-border+solid-fill-stroke", "pipe-border+solid-fill-stroke", "solid-border+pipe-fill-stroke",
                                  "solid-border+rounddot-fill-stroke", "dashed-border+rounddot-fill-stroke", "no-border+pipe-fill-stroke"])
def test_sticky_round_caps_bit_exact(hip, case):
    """the rule's caps stay round once a rounddot / pipe stroke has set them (cairo_smr_dash :909-921): solid fill strokes after a
    round border, solid or dashed borders of the ways after the first open way's round fill stroke, the dots of style=pipe on a
    fill stroke; whole canvas and two bands, against the oracle (pinned on libcairo: test_oracle_cairo.py::test_sticky_round_caps)"""
    from test_oracle_cairo import STICKY_CASES, _sticky_scene
    st = STICKY_CASES[case]
    W, H = 240, 200
    ways = _sticky_scene(W, H, 7)
    arrs = _ways_to_arrays(ways)
    want = po.oracle_scene([(st, ways)], 300, W, H, 0xffffff)
    got = gpu_scene(hip, [(st, arrs)], 300, W, H, 0xffffff)
    assert np.array_equal(got, want), report(got, want)
    assert (got != 0xffffffff).sum() > 5000
    for y0, h in ((0, 96), (96, 104)):
        band = gpu_scene(hip, [(st, arrs)], 300, W, H, 0xffffff, y0, h)
        assert np.array_equal(band, want[y0:y0 + h]), (y0, report(band, want[y0:y0 + h]))


def test_dash_styles_the_gpu_path_refuses(hip):
    sa = hip
    x, y, off, ref = _ways_to_arrays([(np.array([3.0, 40.0, 50.0]), np.array([4.0, 30.0, 8.0]), False)])
    cv = sa.Canvas(64, 64, 300, 0xffffff)
    args = (sa.DeviceArray.from_numpy(x), sa.DeviceArray.from_numpy(y), sa.DeviceArray.from_numpy(off),
            sa.DeviceArray.from_numpy(ref), 1, len(ref))
    S = po.Style
    for st in (S(0, 1, 0, 0, 0, 0.1, 0).with_dash(border_style=S.DASHED, border_dash=(0, 0)),):   # libcairo: invalid dash
        with pytest.raises(sa.SmrHipError):
            cv.draw(draw_style(sa, st), *args)
    cv.close()


WAVY_STYLES = {
    "hairline": po.Style(1, 0, 0xff00ff, 0, 0.0, 0, 0),                    # draw:color=magenta;wavy=yes (rules/available/seamark_area.osm:38)
    "line_02": po.Style(1, 0, 0xff00ff, 0, 0.2, 0, 0),                     # test/rules/0131.osm: width=.2
    "border_only": po.Style(0, 1, 0, 0x204060, 0, 0.3, 0),
    "bordered": po.Style(1, 1, 0x30c0d8f0, 0x000000, 0.3, 0.1, 0),
}


@pytest.mark.parametrize("style", list(WAVY_STYLES))
@pytest.mark.parametrize("dpi,closed", [(300, False), (600, False), (300, True)])
def test_explicit_bezier_paths_bit_exact(hip, style, dpi, closed):
    """curve=2: the ways are p0 + (c1, c2, p) triples (what cairo_smr_wavy records for wavy=1, src/smrules_cairo.c:752-840);
    the GPU decomposes and strokes them like libcairo: bit-exact against the oracle (pinned against libcairo in
    tests/test_oracle_cairo.py::test_explicit_bezier_paths), whole canvas and per band"""
    sa = hip
    W, H = 320, 256
    st = WAVY_STYLES[style].with_knots()
    ways = synth.synth_wavy_knots(24, W, H, seed=5, wave_px=(6.0, 30.0) if dpi == 300 else (12.0, 60.0), closed=closed)
    want = po.oracle_scene([(st, ways)], dpi, W, H, 0xffffff)
    x = np.concatenate([w[0] for w in ways]); y = np.concatenate([w[1] for w in ways])
    off = np.concatenate([[0], np.cumsum([len(w[0]) for w in ways])]).astype(np.int64)
    ref = np.arange(len(x), dtype=np.int32)
    if closed:                       # the closing point of a knot way is the REF of its first node
        ref[off[1:] - 1] = off[:-1]
    dx, dy, doff, dref = (sa.DeviceArray.from_numpy(a) for a in (x, y, off, ref))
    cv = sa.Canvas(W, H, dpi, 0xffffff)
    cv.draw(draw_style(sa, st), dx, dy, doff, dref, len(ways), len(x))
    got = cv.read(); cv.close()
    assert len(np.unique(got)) > 2
    assert np.array_equal(got, want), int((got != want).sum())
    for y0, h in ((0, 100), (100, 156)):
        cb = sa.Canvas(W, H, dpi, 0xffffff, y0, h)
        cb.draw(draw_style(sa, st), dx, dy, doff, dref, len(ways), len(x))
        assert np.array_equal(cb.read(), want[y0:y0 + h]); cb.close()


def test_explicit_bezier_paths_refuse_closed_ways(hip):
    sa = hip
    cv = sa.Canvas(64, 64, 300, 0xffffff)
    xy = sa.DeviceArray.from_numpy(np.array([5.0, 20, 40, 50, 30, 10]))
    off = sa.DeviceArray.from_numpy(np.array([0, 7], np.int64))
    ref = sa.DeviceArray.from_numpy(np.array([0, 1, 2, 3, 4, 5, 0], np.int32))      # first ref == last ref
    with pytest.raises(sa.SmrHipError) as e:
        cv.draw(sa.DrawStyle(1, 0, 0, 0, 0.2, 0, 0, 0, 0, 0, 0, 0).set_knots(), xy, xy, off, ref, 1, 7)
    assert e.value.code == -5
    cv.close()


@pytest.mark.parametrize("n_ways", [700, 6000])
def test_many_primitives_on_one_tile(hip, n_ways):
    """hundreds / thousands of small translucent polygons on the same 64x4-px tiles: the tile's work stream is sorted in LDS sized
    for the fullest tile of the launch (700 ways) or, beyond 4096 entries, by the same network in global memory (6000 ways);
    the drawing order decides every pixel"""
    sa = hip
    W, H, dpi = 128, 64, 300
    rng = np.random.default_rng(17)
    xs, ys, off = [], [], [0]
    for _ in range(n_ways):
        cx, cy, r = rng.uniform(8, 56), rng.uniform(8, 20), rng.uniform(2, 6)
        a = np.sort(rng.uniform(0, 2 * np.pi, 3))
        x, y = cx + r * np.cos(a), cy + r * np.sin(a)
        xs.append(np.append(x, x[0])); ys.append(np.append(y, y[0])); off.append(off[-1] + 4)
    x, y, off = np.concatenate(xs), np.concatenate(ys), np.array(off, np.int64)
    ref = np.arange(len(x), dtype=np.int32); ref[off[1:] - 1] = off[:-1]
    cols = (rng.integers(0, 1 << 24, n_ways) | (rng.integers(0x20, 0x70, n_ways) << 24)).astype(np.uint32)      # translucent, all different
    st = po.Style(1, 0, 0, 0, 0, 0, 0)
    ways = po.ways_from_arrays(x, y, off, ref)
    L = po.orc()
    c = L.orc_canvas_new(0, 0, W, H, dpi, 0xffffff)
    L.orc_rule_begin(c)
    for (wx, wy, closed), col in zip(ways, cols):
        wx = np.ascontiguousarray(wx, np.float64); wy = np.ascontiguousarray(wy, np.float64)
        L.orc_draw_way(c, C.byref(st), po.P(wx), po.P(wy), len(wx), 1, 1, int(col), 0, 0)
    L.orc_rule_end(c)
    want = np.ctypeslib.as_array(C.cast(L.orc_canvas_pixels(c), C.POINTER(C.c_uint32)), (H, W)).copy()
    L.orc_canvas_free(c)
    cv = sa.Canvas(W, H, dpi, 0xffffff)
    cv.draw(draw_style(sa, st), *(sa.DeviceArray.from_numpy(a) for a in (x, y, off, ref)), n_ways, len(ref),
            d_fill_col=sa.DeviceArray.from_numpy(cols))
    got = cv.read(); cv.close()
    assert np.array_equal(got, want), int((got != want).sum())


def test_directional_rule_of_knot_ways(hip):
    """directional=1 with explicit Bezier knots (wavy=1): open knot ways first, then ALL closed knot ways as one non-zero fill,
    each sub-path closed by the fill itself (act_draw_fini :1084-1140 calls cairo_smr_poly, i.e. cairo_smr_wavy, per way)"""
    sa = hip
    W, H, dpi = 320, 256, 300
    ways = synth.synth_wavy_knots(8, W, H, seed=3, closed=True) + synth.synth_wavy_knots(6, W, H, seed=4)
    ways[1] = (ways[1][0][::-1].copy(), ways[1][1][::-1].copy(), True)       # one ring the other way round (its knots reversed: still p0 ... p0)
    st = po.Style(1, 0, 0x4080c0c0, 0, 0.25, 0, 0).with_knots()
    want = po.oracle_directional(st, ways, dpi, W, H, 0xffffff)
    x = np.concatenate([w[0] for w in ways]); y = np.concatenate([w[1] for w in ways])
    off = np.concatenate([[0], np.cumsum([len(w[0]) for w in ways])]).astype(np.int64)
    ref = np.arange(len(x), dtype=np.int32)
    for k, w in enumerate(ways):
        if w[2]:
            ref[off[k + 1] - 1] = off[k]
    cv = sa.Canvas(W, H, dpi, 0xffffff)
    cv.draw(draw_style(sa, st, directional=1), *(sa.DeviceArray.from_numpy(a) for a in (x, y, off, ref)), len(ways), len(x))
    got = cv.read(); cv.close()
    assert len(np.unique(got)) > 2
    assert np.array_equal(got, want), int((got != want).sum())


def test_bad_batches_are_refused_before_anything_is_drawn(hip):
    """smr_hip_draw checks the device-side batch itself (the way table starts at 0, ascends, ends at n_refs; every ref inside
    the node arrays; no node more than 32 768 px beyond the chart, where an edge's 60 * dy no longer fits the 31 bits of the 24.8 edge arithmetic): SMR_E_INVAL / SMR_E_CAPACITY,
    the canvas untouched, and the next good rule draws as if nothing had happened."""
    sa = hip
    W, H = 96, 80
    x, y, off, ref = po.synth_ways(6, W, H, seed=2)
    off = off.astype(np.int64); ref = ref.astype(np.int32)
    cv = sa.Canvas(W, H, 300, 0xffffff)
    gs = draw_style(sa, STYLES["fill+border"])
    D = sa.DeviceArray.from_numpy
    bad_ref = ref.copy(); bad_ref[7] = len(x) + 5
    neg_ref = ref.copy(); neg_ref[0] = -1
    bad_off = off.copy(); bad_off[2], bad_off[3] = off[3], off[2]            # not ascending
    short_off = off.copy(); short_off[-1] -= 1                                # does not end at n_refs
    late_off = off.copy(); late_off[0] = 1                                    # does not start at 0
    far_x = x.copy(); far_x[3] = 2.0 ** 21
    nan_y = y.copy(); nan_y[5] = np.nan
    cases = [(x, y, off, bad_ref, -2), (x, y, off, neg_ref, -2), (x, y, bad_off, ref, -2), (x, y, short_off, ref, -2),
             (x, y, late_off, ref, -2), (far_x, y, off, ref, -6), (x, nan_y, off, ref, -6)]
    for cx, cy, co, cr, code in cases:
        with pytest.raises(sa.SmrHipError) as e:
            cv.draw(gs, D(cx), D(cy), D(co), D(cr), len(co) - 1, len(cr))
        assert e.value.code == code, str(e.value)
        assert np.all(cv.read() == 0xffffffff)
    with pytest.raises(sa.SmrHipError) as e:          # identity refs beyond the node arrays: caught on the host
        cv.draw(gs, D(x[:5]), D(y[:5]), D(np.array([0, 8], np.int64)), None, 1, 8)
    assert e.value.code == -2
    cv.draw(gs, D(x), D(y), D(off), D(ref), len(off) - 1, len(ref))
    got = cv.read(); cv.close()
    want = oracle_of([(STYLES["fill+border"], (x, y, off, ref))], 300, W, H, 0xffffff)
    assert np.array_equal(got, want), report(got, want)


def _hairpin_ways(n_ways, W, H, seg_px, seed, closed_share=0.0):
    """ways of sharp turns and short segments: joins whose inner corner pokes through the neighbouring segment"""
    rng = np.random.default_rng(seed)
    xs, ys, off, ref = [], [], [0], []
    for w in range(n_ways):
        n = int(rng.integers(3, 10))
        x = [rng.uniform(0, W)]; y = [rng.uniform(0, H)]
        ang = rng.uniform(0, 2 * np.pi)
        for k in range(n - 1):
            ang += rng.choice([rng.uniform(2.6, 3.6), rng.uniform(-0.5, 0.5), rng.uniform(1.5, 2.5)])
            L = rng.uniform(0.3, 3.0) * seg_px
            if rng.random() < 0.1:
                L = 0.0                               # a repeated point
            x.append(x[-1] + L * np.cos(ang)); y.append(y[-1] + L * np.sin(ang))
        r = list(range(off[-1], off[-1] + n))
        if rng.random() < closed_share and n >= 4:
            r[-1] = r[0]                              # a closed way: its last ref is its first node (the node of its own stays unused)
        xs.append(np.array(x)); ys.append(np.array(y)); ref.extend(r); off.append(off[-1] + n)
    return np.concatenate(xs), np.concatenate(ys), np.array(off, np.int64), np.array(ref, np.int32)


@pytest.mark.parametrize("case", ["border_wide", "border_narrow", "open_fill_stroke", "fill+border_mixed", "translucent_600dpi"])
def test_hairpin_ways_take_the_strokers_polygon(hip, case):
    """Where a face's inner corner pokes through a neighbouring segment shorter than the stroke's half width, the union of the
    stroke's convex pieces is not libcairo's polygon (a loop wound the other way cancels coverage there): such ways are
    stroked as the polygon itself (k_stroke_risk).  GPU == oracle bit for bit; and against the real libcairo the
    gross differences of the pieces (up to 200 levels) are gone."""
    W, H, dpi = 400, 300, 300
    st, seg, closed = {
        "border_wide": (po.Style(0, 1, 0, 0x203040, 0, 1.2, 0), 8, 0.0),
        "border_narrow": (po.Style(0, 1, 0, 0x203040, 0, 0.3, 0), 3, 0.0),
        "open_fill_stroke": (po.Style(1, 0, 0xc04020, 0, 0.8, 0, 0), 6, 0.0),
        "fill+border_mixed": (po.Style(1, 1, 0xe0c000, 0x000000, 0.6, 0.3, 0), 6, 0.4),
        "translucent_600dpi": (po.Style(0, 1, 0, 0x60102030, 0, 0.8, 0), 10, 0.0),
    }[case]
    if case == "translucent_600dpi":
        dpi = 600
    arrs = _hairpin_ways(40, W, H, seg, seed=len(case), closed_share=closed)
    rules = [(st, arrs)]
    g = gpu_scene(hip, rules, dpi, W, H, 0xffffff)
    o = oracle_of(rules, dpi, W, H, 0xffffff)
    assert np.array_equal(g, o), report(g, o)
    # the rule really met poked-through joins
    L = po.orc()
    L.orc_stroke_risky.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int]
    x, y, off, ref = arrs
    width_mm = st.border_width_mm if st.border_used and not st.fill_used else (st.fill_width_mm if not st.border_used else 2 * st.border_width_mm + st.fill_width_mm)
    risky = 0
    for w in range(len(off) - 1):
        r = ref[off[w]:off[w + 1]]
        dev = np.zeros(2 * len(r), np.int32)
        L.orc_px_to_dev(po.P(np.ascontiguousarray(x[r])), po.P(np.ascontiguousarray(y[r])), len(r), dpi, po.P(dev))
        risky += L.orc_stroke_risky(po.P(dev), len(r), width_mm * 72 / 25.4 / 2, dpi)
    assert risky >= 5, risky
    if po.cairo() is not None and case != "fill+border_mixed":
        c = po.cairo_scene([(st, po.ways_from_arrays(*arrs))], dpi, W, H, 0xffffff)
        d = po.chan_diff(g, c)[2:-2, 2:-2]        # (the image's last row and column: the translucent layer's own edge)
        assert (d > 0).mean() < 0.004 and (d > 64).sum() <= 4, ((d > 0).sum(), d.max(), (d > 64).sum())


@pytest.mark.parametrize("style", ["fill+border", "translucent"])
def test_alpha_word_tiles_composite_alike_in_both_kernels(hip, style, monkeypatch):
    """tiles whose stream holds k_cover's alpha words only are composited by k_blend, the others by k_raster: with
    SMR_HIP_BLEND=0 (every tile through k_raster) and SMR_HIP_COVER=0 (no alpha words at all) the chart is the same, and the
    oracle's; two rules on one canvas, bands included (the second rule meets the first one's pixels)"""
    sa = hip
    W, H = 700, 520
    rules = [(STYLES[style], synth.synth_ways(900, W, H, seed=31, rmin=3.0, rmax=40.0)),
             (STYLES["translucent"], synth.synth_ways(300, W, H, seed=32, rmin=3.0, rmax=70.0))]
    want = oracle_of(rules, 300, W, H, 0xfff0e0d0)
    got = gpu_scene(sa, rules, 300, W, H, 0xfff0e0d0)
    assert np.array_equal(got, want), report(got, want)
    # (OVERLAP: the binning after k_cover instead of beside it; LINEAR: no fill takes k_cover's crossing-by-crossing path; SPLIT: a
    # primitive's strips are one wavefront's, as on a full chart, instead of shared out among up to eight as for a rule this small)
    for var in ("SMR_HIP_BLEND", "SMR_HIP_COVER", "SMR_HIP_OVERLAP", "SMR_HIP_LINEAR", "SMR_HIP_SPLIT"):
        monkeypatch.setenv(var, "0")
        other = gpu_scene(sa, rules, 300, W, H, 0xfff0e0d0)
        monkeypatch.delenv(var)
        assert np.array_equal(other, got), (var, report(other, got))
    for y0, h in ((0, 200), (200, 320)):
        band = gpu_scene(sa, rules, 300, W, H, 0xfff0e0d0, y0, h)
        assert np.array_equal(band, want[y0:y0 + h]), report(band, want[y0:y0 + h])


def _star_ways(n_ways, W, H, seed, rmax=50.0, big=0):
    """closed ways that cross themselves ({n/k} stars, closed random walks); `big` of them with thousands of nodes"""
    rng = np.random.default_rng(seed)
    ways = []
    for i in range(n_ways):
        cx, cy, r = rng.uniform(0.2 * W, 0.8 * W), rng.uniform(0.2 * H, 0.8 * H), rng.uniform(6, rmax)
        if i < big:
            n = 4500 + 200 * i; k = 7
            t = 2 * np.pi * k * np.arange(n) / n + rng.uniform(0, 6)
            rr = 0.35 * min(W, H) * (0.6 + 0.4 * np.cos(11 * 2 * np.pi * np.arange(n) / n))
            cx, cy = W / 2, H / 2
        elif i % 3 == 2:
            n = int(rng.integers(5, 12))
            x, y = cx + rng.uniform(-r, r, n), cy + rng.uniform(-r, r, n)
            ways.append((np.append(x, x[0]), np.append(y, y[0]), True))
            continue
        else:
            n = int(rng.choice([5, 7, 8, 9, 11])); k = 2 if n < 8 else 4 if n == 9 else 3
            t = rng.uniform(0, 6) + 2 * np.pi * k * np.arange(n) / n * rng.choice([-1, 1])
            rr = r * rng.uniform(0.8, 1.0, n)
        x, y = cx + rr * np.cos(t), cy + rr * np.sin(t)
        ways.append((np.append(x, x[0]), np.append(y, y[0]), True))
    return ways


@pytest.mark.parametrize("col,bcol", [(0x4080c0, 0x000000), (0x40c04020, 0x30102030), (0xc04020, None)])
def test_even_odd_fill_rule_bit_exact(hip, col, bcol, monkeypatch):
    """smr_draw_style_t.evenodd (libcairo's CAIRO_FILL_RULE_EVEN_ODD; the reference keeps the default rule, see the header):
    self-crossing ways through k_cover + k_blend, through k_raster alone (SMR_HIP_COVER=0), in bands, and against the same
    ways under the winding rule (the pictures must differ)"""
    sa = hip
    W, H = 520, 400
    ways = _star_ways(120, W, H, seed=11)
    arrs = _ways_to_arrays(ways)
    st = po.Style(1, 1 if bcol is not None else 0, col, bcol or 0, 0, 0.12, 0).with_evenodd()
    want = oracle_of([(st, arrs)], 300, W, H, 0xffffff)
    got = gpu_scene(sa, [(st, arrs)], 300, W, H, 0xffffff)
    assert np.array_equal(got, want), report(got, want)
    st_w = po.Style(1, 1 if bcol is not None else 0, col, bcol or 0, 0, 0.12, 0)
    assert (gpu_scene(sa, [(st_w, arrs)], 300, W, H, 0xffffff) != got).sum() > 2000
    monkeypatch.setenv("SMR_HIP_COVER", "0")
    other = gpu_scene(sa, [(st, arrs)], 300, W, H, 0xffffff)
    monkeypatch.delenv("SMR_HIP_COVER")
    assert np.array_equal(other, got), report(other, got)
    for y0, h in ((0, 140), (140, 260)):
        band = gpu_scene(sa, [(st, arrs)], 300, W, H, 0xffffff, y0, h)
        assert np.array_equal(band, want[y0:y0 + h]), report(band, want[y0:y0 + h])


def test_even_odd_long_primitives_and_compound(hip):
    """the rule reaches the long-primitive path (a self-crossing way of ~5 000 nodes: k_big_* + k_raster's own scan conversion)
    and the ONE polygon of a directional rule (all closed ways together, even-odd: overlapping ways cancel)"""
    sa = hip
    W, H = 640, 480
    ways = _star_ways(3, W, H, seed=3, big=2)
    arrs = _ways_to_arrays(ways)
    st = po.Style(1, 0, 0x4080c0, 0, 0, 0, 0).with_evenodd()
    want = oracle_of([(st, arrs)], 300, W, H, 0xffffff)
    got = gpu_scene(sa, [(st, arrs)], 300, W, H, 0xffffff)
    assert np.array_equal(got, want), report(got, want)
    ways = _star_ways(60, W, H, seed=4, rmax=90.0)
    st = po.Style(1, 0, 0x30c0d8f0, 0, 0.3, 0, 0).with_evenodd()
    want = po.oracle_directional(st, ways, 300, W, H, 0xffffff)
    x, y, off, ref = _ways_to_arrays(ways)
    cv = sa.Canvas(W, H, 300, 0xffffff)
    gs = draw_style(sa, st, directional=1)
    cv.draw(gs, sa.DeviceArray.from_numpy(x), sa.DeviceArray.from_numpy(y), sa.DeviceArray.from_numpy(off),
            sa.DeviceArray.from_numpy(ref), len(off) - 1, len(ref))
    got = cv.read()
    cv.close()
    assert np.array_equal(got, want), report(got, want)
    st_w = po.Style(1, 0, 0x30c0d8f0, 0, 0.3, 0, 0)
    assert (po.oracle_directional(st_w, ways, 300, W, H, 0xffffff) != want).sum() > 2000


def _shape_ways(W, H, seed):
    """closed ways that try the two fill paths of k_cover: rectangles on whole and half pixels (vertical edges, rows without a
    vertex, crossings that tie), stars with deep teeth (exact rows with four and six spanning edges), L and U shapes and
    spirals (not star-shaped: the strip path), slivers, tiny polygons inside one pixel, polygons 120 px wide and 200 px tall"""
    rng = np.random.default_rng(seed)
    ways = []

    def add(x, y):
        ways.append((np.append(x, x[0]), np.append(y, y[0]), True))
    for i in range(260):
        cx, cy = rng.uniform(0, W), rng.uniform(0, H)
        kind = i % 13
        if kind == 0:        # rectangle on the pixel grid
            w, h = rng.integers(1, 40), rng.integers(1, 40)
            x0, y0 = np.floor(cx), np.floor(cy)
            add(np.array([x0, x0 + w, x0 + w, x0]), np.array([y0, y0, y0 + h, y0 + h]))
        elif kind == 1:      # rectangle on half pixels, clockwise
            w, h = rng.integers(1, 30) + 0.5, rng.integers(1, 30) + 0.5
            x0, y0 = np.floor(cx) + 0.5, np.floor(cy) + 0.5
            add(np.array([x0, x0, x0 + w, x0 + w]), np.array([y0, y0 + h, y0 + h, y0]))
        elif kind in (2, 3): # star with deep teeth: long radial edges, rows without a vertex crossed four and six times
            n = int(rng.choice([6, 8, 10, 14])); r = rng.uniform(20, 60)
            t = rng.uniform(0, 6) + 2 * np.pi * np.arange(n) / n
            rr = np.where(np.arange(n) % 2, r, r * rng.uniform(0.15, 0.5))
            add(cx + rr * np.cos(t), cy + rr * np.sin(t))
        elif kind == 4:      # L shape
            a, b = rng.uniform(10, 50), rng.uniform(3, 9)
            add(cx + np.array([0, a, a, b, b, 0]), cy + np.array([0, 0, b, b, a, a]))
        elif kind == 5:      # U shape: two arms on the same rows
            a, b = rng.uniform(12, 50), rng.uniform(2, 5)
            add(cx + np.array([0, b, b, a - b, a - b, a, a, 0]), cy + np.array([0, 0, a - b, a - b, 0, 0, a, a]))
        elif kind == 6:      # spiral
            n = 40; t = np.linspace(0, 5 * np.pi, n); r = 2 + 3 * t
            xo, yo = cx + r * np.cos(t), cy + r * np.sin(t)
            xi, yi = cx + (r - 2.5) * np.cos(t), cy + (r - 2.5) * np.sin(t)
            add(np.concatenate([xo, xi[::-1]]), np.concatenate([yo, yi[::-1]]))
        elif kind == 7:      # sliver
            L_, a = rng.uniform(20, 100), rng.uniform(0, 6)
            add(cx + np.array([0, L_ * np.cos(a), 0.3 * np.sin(a)]), cy + np.array([0, L_ * np.sin(a), -0.3 * np.cos(a)]))
        elif kind == 8:      # inside one pixel
            n = int(rng.integers(3, 30)); t = np.sort(rng.uniform(0, 2 * np.pi, n))
            add(cx + 0.4 * rng.uniform(0.3, 1, n) * np.cos(t), cy + 0.4 * rng.uniform(0.3, 1, n) * np.sin(t))
        elif kind == 9:      # wide and tall ellipse of 60 nodes
            n = 60; t = 2 * np.pi * np.arange(n) / n
            add(cx + 59 * np.cos(t) * rng.uniform(0.9, 1, n), cy + 100 * np.sin(t) * rng.uniform(0.9, 1, n))
        elif kind == 10:     # convex polygon with many collinear points on its sides
            x0, y0, w, h = cx, cy, rng.uniform(10, 60), rng.uniform(10, 60)
            k = 6
            xs = np.concatenate([x0 + w * np.arange(k) / k, np.full(k, x0 + w), x0 + w - w * np.arange(k) / k, np.full(k, x0)])
            ys = np.concatenate([np.full(k, y0), y0 + h * np.arange(k) / k, np.full(k, y0 + h), y0 + h - h * np.arange(k) / k])
            add(xs, ys)
        else:                # C2's own generator shape, either orientation
            n = int(rng.integers(4, 64)); r = rng.uniform(3, 62)
            t = 2 * np.pi * np.arange(n) / n * rng.choice([-1, 1]) + rng.uniform(0, 6)
            rr = r * rng.uniform(0.7, 1.0, n)
            add(cx + rr * np.cos(t), cy + rr * np.sin(t))
    return ways


@pytest.mark.parametrize("col", [0xc04020, 0x40c04020])
@pytest.mark.parametrize("dpi", [300, 97])
def test_fill_paths_of_k_cover_agree_with_the_oracle_and_with_each_other(hip, col, dpi, monkeypatch):
    """k_cover fills a star-shaped polygon crossing by crossing (no lists, no sort) and everything else strip by strip: both must be
    the oracle's pixels, on shapes chosen to sit on the border between the two and on the rules that look at neighbouring
    crossings (touching spans, exact rows whose span ends where the next begins)"""
    sa = hip
    W, H = 640, 520
    arrs = _ways_to_arrays(_shape_ways(W, H, seed=dpi))
    st = po.Style(1, 0, col, 0, 0, 0, 0)
    want = oracle_of([(st, arrs)], dpi, W, H, 0xffffff)
    sa.lib().smr_hip_set_profiling(1)
    got = gpu_scene(sa, [(st, arrs)], dpi, W, H, 0xffffff)
    tm = sa.timings()
    sa.lib().smr_hip_set_profiling(0)
    assert np.array_equal(got, want), report(got, want)
    assert 0.4 * tm["n_cover_prims"] < tm["n_cover_linear"] < 0.9 * tm["n_cover_prims"], tm      # both paths were taken
    monkeypatch.setenv("SMR_HIP_LINEAR", "0")
    other = gpu_scene(sa, [(st, arrs)], dpi, W, H, 0xffffff)
    assert np.array_equal(other, got), report(other, got)
    monkeypatch.delenv("SMR_HIP_LINEAR")
    for y0, h in ((0, 203), (203, 317)):
        band = gpu_scene(sa, [(st, arrs)], dpi, W, H, 0xffffff, y0, h)
        assert np.array_equal(band, want[y0:y0 + h]), report(band, want[y0:y0 + h])


@pytest.mark.parametrize("alpha,style", [(255, po.Style(0, 0, 0, 0, 0, 0, 0)), (140, po.Style(0, 1, 0, 0x000000, 0, 0.2, 0)),
                                         (200, po.Style(1, 1, 0x604080c0, 0x102030, 0.3, 0.15, 0))])
def test_image_fill_bit_exact(hip, alpha, style):
    """smr_hip_draw_pattern (render_poly_line's image fill, src/smrules_cairo.c:989-997): the closed ways filled with the repeated
    image in a group below the rule's own; against the oracle (pinned on libcairo's surface pattern in
    tests/test_oracle_cairo.py::test_image_fill_of_closed_ways), whole canvas and two bands (the tiling starts at the CHART's origin)"""
    from test_oracle_cairo import _pattern, _dash_scene, _nested_scene
    sa = hip
    W, H = 200, 160
    pat = _pattern(37, 23, 3, alpha)
    ways = _dash_scene(W, H, 11)
    before = [(po.Style(1, 0, 0xe0f0e0, 0, 0, 0, 0), _nested_scene(W, H, 2))]
    want = po.oracle_pattern_scene(before, (style, ways, pat), [], 300, W, H, 0xffffff)

    def gpu(y0, h):
        cv = sa.Canvas(W, H, 300, 0xffffff, y0, h)
        def dev(ws):
            x, y, off, ref = _ways_to_arrays(ws)
            return (sa.DeviceArray.from_numpy(x), sa.DeviceArray.from_numpy(y), sa.DeviceArray.from_numpy(off.astype(np.int64)),
                    sa.DeviceArray.from_numpy(ref.astype(np.int32)), len(off) - 1, len(ref))
        for st, ws in before:
            cv.draw(draw_style(sa, st), *dev(ws))
        d = dev(ways)
        cv.draw_pattern(draw_style(sa, style), pat, *d)
        if style.fill_used or style.border_used:
            cv.draw(draw_style(sa, style), *d)
        out = cv.read()
        cv.close()
        return out

    got = gpu(0, H)
    assert np.array_equal(got, want), report(got, want)
    plain = po.oracle_scene(before + ([(style, ways)] if style.fill_used or style.border_used else []), 300, W, H, 0xffffff)
    assert (got != plain).sum() > 3000
    for y0, h in ((0, 60), (60, 100)):
        band = gpu(y0, h)
        assert np.array_equal(band, want[y0:y0 + h]), (y0, report(band, want[y0:y0 + h]))


# ---- dash styles along curves (round 3): libcairo's dashing stroker walks the decomposed curve with faces along the chords, round
# joins between them and tangent faces where a curve_to ends (oracle: orc_dash_edges_curve, pinned on libcairo in
# tests/test_oracle_cairo.py::test_dashed_curves); k_dash does the same walk, one lane per way
DASHED_CURVE_STYLES = {
    "dashed-border": po.Style(0, 1, 0, 0x203060, 0, 0.3, 0).with_dash(border_style=1),
    "dotted-border": po.Style(0, 1, 0, 0x203060, 0, 0.4, 0).with_dash(border_style=2),
    "rounddot-border": po.Style(0, 1, 0, 0x203060, 0, 0.5, 0).with_dash(border_style=5),
    "pipe-border": po.Style(0, 1, 0, 0x203060, 0, 0.2, 0).with_dash(border_style=4),
    "dashed-fill-stroke": po.Style(1, 0, 0xa02040, 0, 0.4, 0.2, 0).with_dash(fill_style=1),
    "wide-dashes": po.Style(0, 1, 0, 0x40203060, 0, 1.0, 0).with_dash(border_style=1, border_dash=(3, 1, 1, 1)),
    "fill+dashed-border": po.Style(1, 1, 0x4080c0, 0x000000, 0.3, 0.2, 0).with_dash(border_style=1),
    "both-dashed": po.Style(1, 1, 0x4080c0, 0x000000, 0.4, 0.15, 0).with_dash(border_style=2, fill_style=1),
    # the solid stroker has image limits for curves, the dashing one has none: one decomposition, the border skips (k_curve's bskip)
    "solid-border+dashed-fill-stroke": po.Style(1, 1, 0x4080c0, 0x000000, 0.4, 0.15, 0).with_dash(fill_style=1),
    # the rule's caps are sticky (:909-921): a SOLID stroke after a rounddot / pipe stroke has round caps; along a curve it stays the
    # polygon stroker's (tangent pieces, a half disc at either end: k_dash's solid branch, oracle orc_stroke_edges_round_tan)
    "rounddot-border+solid-fill-stroke": po.Style(1, 1, 0x4080c0, 0x000000, 0.5, 0.15, 0).with_dash(border_style=5),
    "pipe-border+solid-fill-stroke": po.Style(1, 1, 0x4080c0, 0x103050, 0.4, 0.15, 0).with_dash(border_style=4),
    "solid-border+pipe-fill-stroke": po.Style(1, 1, 0xc04020, 0x000000, 0.5, 0.2, 0).with_dash(fill_style=4),
    "solid-border+rounddot-fill-stroke": po.Style(1, 1, 0x40c04020, 0x204060, 0.4, 0.15, 0).with_dash(fill_style=5),
}


@pytest.mark.parametrize("factor", [0.2, 0.35])
@pytest.mark.parametrize("name", list(DASHED_CURVE_STYLES))
def test_dashed_curves_bit_exact(hip, name, factor):
    W, H = 400, 300
    st = DASHED_CURVE_STYLES[name].with_curve(factor)
    closed = synth.synth_ways(14, W, H, seed=41, rmin=12.0, rmax=70.0)
    # (some open ways with legs far beyond the image: the solid stroker's limits turn curves out there into line_to's)
    far = [(np.array([30.0, -900.0, -700.0, 120.0, 200.0]), np.array([40.0, 100.0, -800.0, -1200.0, 150.0]), False),
           (np.array([380.0, 1500.0, 1700.0, 300.0]), np.array([250.0, 300.0, 1600.0, 1400.0]), False)]
    rules = [(st, closed), (st, _open_ways(16, W, H, seed=42)), (st, _ways_to_arrays(far))]
    want = oracle_of(rules, 300, W, H, 0xfff8e0)
    got = gpu_scene(hip, rules, 300, W, H, 0xfff8e0)
    assert np.array_equal(got, want), report(got, want)
    assert (want != 0xfffff8e0).sum() > 3000
    for y0, h in ((0, 120), (120, 180)):
        band = gpu_scene(hip, rules, 300, W, H, 0xfff8e0, y0, h)
        assert np.array_equal(band, want[y0:y0 + h]), report(band, want[y0:y0 + h])


@pytest.mark.gpu
def test_deep_overdraw_with_opaque_and_translucent_colours_per_way(hip):
    """k_blend's rounds (every lane blends its own next entry): hundreds of large polygons over one another on a small canvas, so
    that a tile holds far more than 64 entries (several header batches, several 16-entry batches of words), with a colour per way,
    opaque and translucent ones interleaved: lanes of one round meet pixman's lerp and its OVER side by side"""
    sa = hip
    W, H = 320, 96
    rng = np.random.default_rng(17)
    n_ways = 360
    x, y, off, ref = synth.synth_ways(n_ways, W, H, seed=23, rmin=20.0, rmax=110.0)
    fill_col = rng.integers(0, 1 << 24, n_ways).astype(np.uint32)
    border_col = rng.integers(0, 1 << 24, n_ways).astype(np.uint32)
    tr = rng.uniform(size=n_ways) < 0.5
    fill_col[tr] |= (rng.integers(0x10, 0x70, int(tr.sum())).astype(np.uint32) << 24)
    tb = rng.uniform(size=n_ways) < 0.3
    border_col[tb] |= (rng.integers(0x10, 0x70, int(tb.sum())).astype(np.uint32) << 24)
    st = po.Style(1, 1, 0, 0, 0, 0.15, 0)
    ways = [(wx, wy, c, int(fill_col[k]), int(border_col[k])) for k, (wx, wy, c) in enumerate(po.ways_from_arrays(x, y, off, ref))]
    want = po.oracle_scene([(st, ways)], 300, W, H, 0xfff8e0)
    cv = sa.Canvas(W, H, 300, 0xfff8e0)
    gs = draw_style(sa, st)
    sa.set_profiling(True)
    try:
        cv.draw(gs, sa.DeviceArray.from_numpy(x), sa.DeviceArray.from_numpy(y), sa.DeviceArray.from_numpy(off.astype(np.int64)),
                sa.DeviceArray.from_numpy(ref.astype(np.int32)), len(off) - 1, len(ref),
                sa.DeviceArray.from_numpy(fill_col), sa.DeviceArray.from_numpy(border_col))
        t = sa.timings()
    finally:
        sa.set_profiling(False)
    got = cv.read()
    cv.close()
    assert np.array_equal(got, want), report(got, want)
    assert t["n_tile_refs"] > 100 * t["n_tiles"] or t["n_tile_refs"] > 20000


@pytest.mark.gpu
def test_vertices_a_hair_above_a_rows_first_sample_line(hip):
    """the two sides of a vertex that start at a row's first sub-row in one cell stand in libcairo's active list as the path added
    them, and right side before left side makes it sub-sample the row (oracle: tor_set_path_order, pinned on libcairo in
    tests/test_oracle_cairo.py; device: smr_tie_check): hundreds of triangles with such a vertex, both orientations and every
    starting point, at 72 dpi (a pixel is a pt: the vertex sits where it is put, 8/256 px below a row's top)"""
    sa = hip
    rng = np.random.default_rng(11)
    cols, rows, cell = 24, 16, 48
    W, H = cols * cell, rows * cell
    ways = []
    for k in range(cols * rows):
        ox, oy = (k % cols) * cell, (k // cols) * cell
        R0 = int(rng.integers(8, 20))
        vy, vx, h = R0 + 8 / 256, np.rint(rng.uniform(15, 33) * 256) / 256, rng.uniform(3, 12)
        lx, rx = vx - rng.uniform(0.5, 12), vx + rng.uniform(0.5, 12)
        ly, ry = vy + h * rng.uniform(0.6, 1.4), vy + h * rng.uniform(0.6, 1.4)
        poly = [(vx, vy), (rx, ry), (lx, ly)] if rng.integers(0, 2) else [(vx, vy), (lx, ly), (rx, ry)]
        start = int(rng.integers(0, 3))
        poly = poly[start:] + poly[:start]
        poly.append(poly[0])
        ways.append((np.array([p[0] + ox for p in poly]), np.array([p[1] + oy for p in poly]), True))
    st = po.Style(1, 0, 0x000000, 0, 0, 0, 0)
    arrs = _ways_to_arrays(ways)
    want = oracle_of([(st, arrs)], 72, W, H, 0xffffff)
    got = gpu_scene(sa, [(st, arrs)], 72, W, H, 0xffffff)
    assert np.array_equal(got, want), report(got, want)
    for y0, h in ((0, 5 * cell + 9), (5 * cell + 9, H - 5 * cell - 9)):
        band = gpu_scene(sa, [(st, arrs)], 72, W, H, 0xffffff, y0, h)
        assert np.array_equal(band, want[y0:y0 + h]), report(band, want[y0:y0 + h])
