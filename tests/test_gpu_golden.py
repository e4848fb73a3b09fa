"""HIP draw path vs (a) the committed golden images produced by the real libcairo through the
reference's call sequence -- north_star bar: every channel within +-2/255 on >= 99.5 % of the
pixels -- and (b) the oracle bit-exactly on the same scenes; plus the full-size C2 canvas
checked through windows against the oracle."""
import numpy as np
import pytest
from oracle import pyoracle as po
from smrender_amd import synth
from golden_io import BARS, SCENES, draw_style, load_scene, within

pytestmark = pytest.mark.gpu


def gpu_render(sa, meta, rules):
    W, H = meta["page"][2], meta["page"][3]          # the page (libcairo cuts polygon edges at its sides), the scene's rows as a band
    cv = sa.Canvas(W, H, meta["dpi"], meta["bg"], meta["y0"], meta["H"])
    for st, (x, y, off, ref) in rules:
        gs = draw_style(sa, st)
        cv.draw(gs, sa.DeviceArray.from_numpy(x), sa.DeviceArray.from_numpy(y),
                sa.DeviceArray.from_numpy(off.astype(np.int64)), sa.DeviceArray.from_numpy(ref.astype(np.int32)),
                len(off) - 1, len(ref))
    out = cv.read(meta["x0"], 0, meta["W"], meta["H"])
    cv.close()
    return out


@pytest.mark.parametrize("name", SCENES)
def test_golden_scene(hip, name):
    meta, rules, want = load_scene(name)
    got = gpu_render(hip, meta, rules)
    assert within(got, want, 2) >= max(BARS[name][0], 0.995) and within(got, want, 0) >= BARS[name][1]
    orules = [(st, po.ways_from_arrays(*arrs)) for st, arrs in rules]
    o = po.oracle_scene(orules, meta["dpi"], meta["W"], meta["H"], meta["bg"], meta["x0"], meta["y0"], image=meta["page"])
    assert np.array_equal(got, o)


@pytest.mark.parametrize("name", ["hairpins_300", "c2_fill_border_300", "open_border_300", "layers_translucent_600", "curves_300", "dashes_300"])
def test_golden_scene_with_every_stroke_as_the_strokers_polygon(hip, name, monkeypatch):
    """SMR_HIP_STROKE=polygon (the oracle's ORC_STROKE_OUTLINE=1): every solid stroke is filled as libcairo's stroke polygon
    itself instead of the union of its convex pieces (the default takes the polygon only for ways with a poked-through join).
    Costs 10 % of a C2 step; the hairpin scene and the C2 scene (fills with borders, ways crossing the page's
    sides) are then libcairo's images pixel for pixel, the open-border scene but for one pixel by one level."""
    monkeypatch.setenv("SMR_HIP_STROKE", "polygon")
    monkeypatch.setenv("ORC_STROKE_OUTLINE", "1")
    meta, rules, want = load_scene(name)
    got = gpu_render(hip, meta, rules)
    orules = [(st, po.ways_from_arrays(*arrs)) for st, arrs in rules]
    o = po.oracle_scene(orules, meta["dpi"], meta["W"], meta["H"], meta["bg"], meta["x0"], meta["y0"], image=meta["page"])
    assert np.array_equal(got, o)
    if name in ("hairpins_300", "c2_fill_border_300"):
        assert np.array_equal(got, want)          # libcairo's image, pixel for pixel
    else:
        assert within(got, want, 0) >= {"open_border_300": 0.9999, "layers_translucent_600": 0.9985, "curves_300": 0.9999, "dashes_300": 0.9999}[name]


def test_full_size_c2_windows_vs_oracle(hip):
    """BASELINE config 2 at full size (1 M ways, 16384^2): windows of the device canvas must equal the
    oracle's render of the ways that reach the window (drawing order preserved)."""
    sa = hip
    W = H = 16384
    x, y, off, ref = synth.synth_ways(1000000, W, H, seed=1234)
    cv = sa.Canvas(W, H, 300, 0xffffff)
    gs = sa.DrawStyle(1, 1, 0x4080c0, 0x000000, 0.0, 25.4 / 300, 0, 0, 0, 0, 0, 0)
    cv.draw(gs, sa.DeviceArray.from_numpy(x), sa.DeviceArray.from_numpy(y), sa.DeviceArray.from_numpy(off),
            sa.DeviceArray.from_numpy(ref), len(off) - 1, len(ref))
    x0b, y0b, x1b, y1b = synth.way_bbox(x, y, off, ref)
    st = po.Style(1, 1, 0x4080c0, 0x000000, 0.0, 25.4 / 300, 0)
    S = 192
    for wx, wy in [(0, 0), (W - S, H - S), (8000, 3000), (1234, 15000), (16000, 77)]:
        got = cv.read(wx, wy, S, S)
        keep = (x1b > wx - 8) & (x0b < wx + S + 8) & (y1b > wy - 8) & (y0b < wy + S + 8)
        sx, sy, soff, sref = synth.select_ways(x, y, off, ref, keep)
        want = po.oracle_scene([(st, po.ways_from_arrays(sx, sy, soff, sref))], 300, S, S, 0xffffff, wx, wy, image=(0, 0, W, H))
        assert np.array_equal(got, want), (wx, wy, int((got != want).sum()))
    # idempotence of the pipeline: a second identical chart gives the identical canvas (no order races)
    first = cv.read(4000, 4000, 512, 512)
    cv.clear()
    cv.draw(gs, sa.DeviceArray.from_numpy(x), sa.DeviceArray.from_numpy(y), sa.DeviceArray.from_numpy(off),
            sa.DeviceArray.from_numpy(ref), len(off) - 1, len(ref))
    assert np.array_equal(cv.read(4000, 4000, 512, 512), first)
    cv.close()


def test_full_size_c3_windows_vs_oracle(hip):
    """BASELINE config 3 at full size (A0 at 600 dpi, 19866 x 28087 px): a fractal coastline of 200 k vertices + 5 000 islands
    as ONE directional curve=1 fill, 50 depth contours of ~10 k nodes as curve=1 borders.  Windows on the coast must equal
    the oracle's render of the whole ways clipped to the window (curve limits = the full chart)."""
    sa = hip
    W, H, dpi = 19866, 28087, 600
    land, rings = synth.synth_coast(200000, 5000, 50, W, H, seed=1234)
    st_land = po.Style(1, 0, 0xf0e0a0, 0, 0, 0, 0).with_curve()
    st_ring = po.Style(0, 1, 0, 0x4060a0, 0, 0.15, 0).with_curve()
    from golden_io import draw_style

    def draw(cv, st, arrs, **kw):
        x, y, off, ref = arrs
        cv.draw(draw_style(sa, st, **kw), sa.DeviceArray.from_numpy(x), sa.DeviceArray.from_numpy(y),
                sa.DeviceArray.from_numpy(off.astype(np.int64)), sa.DeviceArray.from_numpy(ref.astype(np.int32)), len(off) - 1, len(ref))
    S = 160
    wins = [(int(land[0][k]) - S // 2, int(land[1][k]) - S // 2) for k in (0, 50000, 123456)]
    cv = sa.Canvas(W, H, dpi, 0xc0d8f0)
    draw(cv, st_land, land, directional=1)
    got_land = [cv.read(wx, wy, S, S) for wx, wy in wins]
    cv.clear()
    draw(cv, st_ring, rings)
    rwins = [(int(rings[0][k]) - S // 2, int(rings[1][k]) - S // 2) for k in (10, 7000, 33333)]
    got_ring = [cv.read(wx, wy, S, S) for wx, wy in rwins]
    assert sa.lib().smr_hip_draw_status() == 0
    cv.close()
    lways, rways = po.ways_from_arrays(*land), po.ways_from_arrays(*rings)
    for (wx, wy), g in zip(wins, got_land):
        o = po.oracle_directional(st_land, lways, dpi, S, S, 0xc0d8f0, wx, wy, image=(0, 0, W, H))
        assert np.array_equal(g, o), (wx, wy, int((g != o).sum()))
        assert len(np.unique(g)) > 2          # the window does sit on the coast
    for (wx, wy), g in zip(rwins, got_ring):
        o = po.oracle_scene([(st_ring, rways)], dpi, S, S, 0xc0d8f0, wx, wy, image=(0, 0, W, H))
        assert np.array_equal(g, o), (wx, wy, int((g != o).sum()))
        assert len(np.unique(g)) > 2


@pytest.mark.parametrize("mode", ["default", "polygon"])
def test_long_contour_lines_as_the_strokers_polygon(hip, mode, monkeypatch):
    """Contour lines of 10 000 and 8 000 nodes, 1.5 px from node to node under a 1.8-px pen: nearly every join pokes through its
    neighbour, so libcairo's image is the stroker's polygon and nothing else.  Their 80 000 slots go through the long-primitive
    set-up (k_big_*), which used to read a stroke's arena as pieces only (empty slots of a polygon counted as vertices at the
    origin; ways of 512 refs and more were therefore kept in pieces)."""
    sa = hip
    if mode == "polygon":
        monkeypatch.setenv("SMR_HIP_STROKE", "polygon")
        monkeypatch.setenv("ORC_STROKE_OUTLINE", "1")
    W = H = 6000
    dpi = 300
    _, rings = synth.synth_coast(40000, 0, 2, W, H, seed=99)
    x, y, off, ref = rings
    assert list(np.diff(off)) == [10001, 8001]
    st = po.Style(0, 1, 0, 0x203060, 0, 0.15, 0)
    cv = sa.Canvas(W, H, dpi, 0xffffff)
    cv.draw(draw_style(sa, st), sa.DeviceArray.from_numpy(x), sa.DeviceArray.from_numpy(y), sa.DeviceArray.from_numpy(off.astype(np.int64)),
            sa.DeviceArray.from_numpy(ref.astype(np.int32)), len(off) - 1, len(ref))
    status = sa.lib().smr_hip_draw_status()
    S = 200
    wins = [(min(max(int(x[k]) - S // 2, 0), W - S), min(max(int(y[k]) - S // 2, 0), H - S)) for k in (0, 2500, 7777, 10000 + 4000)]
    got = [cv.read(wx, wy, S, S) for wx, wy in wins]
    cv.close()
    ways = po.ways_from_arrays(x, y, off, ref)
    for (wx, wy), g in zip(wins, got):
        o = po.oracle_scene([(st, ways)], dpi, S, S, 0xffffff, wx, wy, image=(0, 0, W, H))
        assert np.array_equal(g, o), (mode, status, wx, wy, int((g != o).sum()))
        assert len(np.unique(g)) > 2
    assert status & 1 == 0       # no SMR_DRAW_DEGRADED_ROWS (SMR_DRAW_SECOND_PASS is informational: a tile of the ring's tangent rows overflowed the fast lists)
