"""The drop-in boundary on the GPU: rules of the form `draw@libsmr_hip.so:...` / `cap@libsmr_hip.so:...`
driven through the reference's own libsmrender (oracle/_ref) by the miniature host, compared with
the oracle restatement of act_draw / cap_coord + find_angle on the same objects."""
import ctypes as C
import os
import numpy as np
import pytest
from oracle import pyoracle as po
import hostsim as hs

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not hs.available(), reason="oracle/_ref not built")]

W, H, DPI = 384, 300, 300


def plugin_canvas(sa):
    L = sa.lib()
    h = L.smr_hip_plugin_canvas()
    assert h
    out = np.empty((H, W), np.uint32)
    assert L.smr_hip_canvas_read(h, 0, 0, W, H, out.ctypes.data) == 0
    return out


@pytest.fixture
def chart(hip):
    """the plugin's chart with the caption scan on CROPS of the canvas (SMR_HIP_CUTOUT=crop): the convention the oracle helpers of
    this file restate; the default (the reference's re-rasterised cut-outs) has its own tests at the end of the file"""
    sa = hip
    sa.lib().smr_hip_plugin_reset()
    os.environ["SMR_HIP_CUTOUT"] = "crop"
    proj = sa.Projection.from_window(45.0, 15.0, 25000.0, float(W), float(H), DPI)
    hs.set_chart(proj, 0xfff8e0)
    yield sa, proj
    os.environ.pop("SMR_HIP_CUTOUT", None)
    sa.lib().smr_hip_plugin_reset()


@pytest.fixture
def chart_replay(hip):
    sa = hip
    sa.lib().smr_hip_plugin_reset()
    os.environ.pop("SMR_HIP_CUTOUT", None)
    proj = sa.Projection.from_window(45.0, 15.0, 25000.0, float(W), float(H), DPI)
    hs.set_chart(proj, 0xfff8e0)
    yield sa, proj
    sa.lib().smr_hip_plugin_reset()


_next_id = [1000]


def load_ways(sa, proj, arrs, tags=None, base=None):
    """puts the synthetic ways into the host's object tree; returns (way ids, oracle way list)"""
    from smrender_amd import synth
    x, y, off, ref = arrs
    lat, lon = synth.px_to_geo(proj, x, y)
    if base is None:
        base = _next_id[0]
        _next_id[0] += len(x) + len(off) + 10
    hs.add_nodes(base + np.arange(len(x)), lat, lon)
    wids = []
    for w in range(len(off) - 1):
        wid = base + len(x) + w
        hs.add_way(wid, base + ref[off[w]:off[w + 1]].astype(np.int64), tags[w] if tags else "")
        wids.append(wid)
    op = po.make_proj(45.0, 15.0, 25000.0, float(W), float(H), DPI)
    ox, oy, _, _, _ = po.project(op, lat, lon)
    return wids, po.ways_from_arrays(ox, oy, off, ref)


def test_draw_rule_through_the_action_api(chart):
    sa, proj = chart
    wids, oways = load_ways(sa, proj, po.synth_ways(60, W, H, seed=77))
    rc = hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "color=#4080c0;bcolor=black;bwidth=0.1", wids)
    assert rc == 0, sa.lib().smr_hip_last_error()
    want = po.oracle_scene([(po.Style(1, 1, 0x4080c0, 0, 0, 0.1, 0), oways)], DPI, W, H, 0xfff8e0)
    got = plugin_canvas(sa)
    assert np.array_equal(got, want)


def test_object_ids_beyond_32_bits(chart):
    """OSM ids passed 2^32 years ago, and smrender gives its own objects negative ones: the host's tree is built as smrender's
    configure builds it (bx_hash_t = int64_t, BX_RES = 4: oracle/Makefile), and two sets of ways whose ids differ by exactly 2^32
    (one object under a 32-bit hash) are two sets of ways at the boundary"""
    sa, proj = chart
    a = po.synth_ways(30, W, H, seed=301)
    b = po.synth_ways(30, W, H, seed=302)
    big = 6_000_000_000_000
    wa, oa = load_ways(sa, proj, a, base=big)
    wb, ob = load_ways(sa, proj, b, base=big + (1 << 32))
    wn, on = load_ways(sa, proj, po.synth_ways(10, W, H, seed=303), base=-7_000_000_000)
    assert hs.way_refs(wb[0])[0] == big + (1 << 32) + int(b[3][0]) and hs.way_refs(wa[0])[0] == big + int(a[3][0])
    ids = sorted(wn) + wa + wb                   # the tree hands objects over in id order
    rc = hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "color=#4080c0;bcolor=black;bwidth=0.1", ids)
    assert rc == 0, sa.lib().smr_hip_last_error()
    want = po.oracle_scene([(po.Style(1, 1, 0x4080c0, 0, 0, 0.1, 0), on + oa + ob)], DPI, W, H, 0xfff8e0)
    assert np.array_equal(plugin_canvas(sa), want)


def test_draw_relation_members_and_key_colours(chart):
    sa, proj = chart
    cols = ["#c04020", "#3020a0c0", "green", "#102030"]
    arrs = po.synth_ways(24, W, H, seed=78)
    tags = [f"colour={cols[w % 4]};name=w{w}" for w in range(24)]
    wids, oways = load_ways(sa, proj, arrs, tags)
    rid = _next_id[0]; _next_id[0] += 1
    hs.add_rel(rid, wids[:10])
    # rule 1: a relation rule draws the member ways (src/smrules_cairo.c:1054-1068)
    assert hs.apply_rule(sa.lib(), "draw", hs.OSM_REL, "bcolor=#204060;bwidth=0.3", [rid]) == 0
    # rule 2: colour per way from a tag (color=%key, color_by_cs :2042-2076)
    assert hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "color=%colour", wids[10:]) == 0
    pc = {"#c04020": 0xc04020, "#3020a0c0": 0x3020a0c0, "green": 0x00ff00, "#102030": 0x102030}
    r1 = (po.Style(0, 1, 0, 0x204060, 0, 0.3, 0), oways[:10])
    r2 = (po.Style(1, 0, 0, 0, 0, 0, 0), [(x, y, c, pc[cols[(10 + k) % 4]], None) for k, (x, y, c) in enumerate(oways[10:])])
    want = po.oracle_scene([r1, r2], DPI, W, H, 0xfff8e0)
    assert np.array_equal(plugin_canvas(sa), want)


@pytest.mark.parametrize("params,style", [
    ("color=#c04020;width=0.5;bcolor=black;bwidth=0.2;style=pipe", dict(fill_style=4)),
    ("color=#c04020;width=0.4;bcolor=#204060;bwidth=0.15;style=rounddot", dict(fill_style=5)),
    ("color=#4080c0;width=0.5;bcolor=black;bwidth=0.15;bstyle=rounddot", dict(border_style=5))])
def test_rules_whose_caps_stay_round_through_the_action_api(chart, params, style):
    """style=pipe|rounddot on fill strokes and solid strokes after a round-capped one (cairo_smr_dash :909-921: the rule's context
    keeps CAIRO_LINE_CAP_ROUND): accepted by act_draw_ini and drawn as the oracle does (pinned on libcairo in test_oracle_cairo.py)"""
    from test_oracle_cairo import _sticky_scene
    from test_gpu_raster import _ways_to_arrays
    sa, proj = chart
    wids, oways = load_ways(sa, proj, _ways_to_arrays(_sticky_scene(W, H, 7)))
    rc = hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, params, wids)
    assert rc == 0, sa.lib().smr_hip_last_error()
    col = lambda k: int(params.split(k + "=#")[1][:6], 16) if k + "=#" in params else 0
    w = lambda k: float(params.split(";" + k + "=")[1].split(";")[0])
    st = po.Style(1, 1, col("color"), col("bcolor"), w("width"), w("bwidth"), 0).with_dash(**style)
    want = po.oracle_scene([(st, oways)], DPI, W, H, 0xfff8e0)
    assert np.array_equal(plugin_canvas(sa), want)


@pytest.mark.parametrize("params,style", [
    ("bcolor=#204060;bwidth=0.25;bstyle=dashed", dict(border_style=1)),
    ("bcolor=#204060;bwidth=0.25;bstyle=dashed;bdash=5:2:1:2", dict(border_style=1, border_dash=(5, 2, 1, 2))),
    ("bcolor=#204060;bwidth=0.25;bstyle=pipe", dict(border_style=4)),
    ("bcolor=#204060;bwidth=0.25;bstyle=rounddot;bdash=0:3mm", dict(border_style=5, border_dash=(0, 3))),
    ("bcolor=#204060;bwidth=0.25;bdash=2:1", dict(border_style=0, border_dash=(2, 1)))])       # a dash array makes any style dashed
def test_dashed_rule_through_the_action_api(chart, params, style):
    """bstyle= / bdash= parsed by the plugin as act_draw_ini does (:623, :645; parse_dash_style src/smrparse.c:1077-1102)"""
    from test_oracle_cairo import _dash_scene
    from test_gpu_raster import _ways_to_arrays
    sa, proj = chart
    wids, oways = load_ways(sa, proj, _ways_to_arrays(_dash_scene(W, H, 9)))
    rc = hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, params, wids)
    assert rc == 0, sa.lib().smr_hip_last_error()
    st = po.Style(0, 1, 0, 0x204060, 0, 0.25, 0).with_dash(**style)
    want = po.oracle_scene([(st, oways)], DPI, W, H, 0xfff8e0)
    assert np.array_equal(plugin_canvas(sa), want)


@pytest.mark.parametrize("params,st", [
    ("bcolor=#203060;bwidth=0.3;curve=1;bstyle=dashed", po.Style(0, 1, 0, 0x203060, 0, 0.3, 0).with_dash(border_style=1)),
    ("bcolor=#203060;bwidth=0.4;curve=1;bstyle=rounddot;curve_factor=0.35", po.Style(0, 1, 0, 0x203060, 0, 0.4, 0).with_dash(border_style=5)),
    ("color=#4080c0;width=0.4;bcolor=black;bwidth=0.15;curve=1;bstyle=dotted;style=dashed", po.Style(1, 1, 0x4080c0, 0, 0.4, 0.15, 0).with_dash(border_style=2, fill_style=1)),
    ("color=#a02040;width=0.4;curve=1;style=pipe;ignore_open=0", po.Style(1, 0, 0xa02040, 0, 0.4, 0, 0).with_dash(fill_style=4)),
    ("color=#4080c0;width=0.4;bcolor=black;bwidth=0.15;curve=1;style=dashed", po.Style(1, 1, 0x4080c0, 0, 0.4, 0.15, 0).with_dash(fill_style=1)),
    # a solid stroke whose caps the rule's other stroke has made round (the last combination refused until round 3)
    ("color=#c04020;width=0.5;bcolor=black;bwidth=0.2;curve=1;style=pipe", po.Style(1, 1, 0xc04020, 0, 0.5, 0.2, 0).with_dash(fill_style=4)),
    ("color=#4080c0;width=0.5;bcolor=black;bwidth=0.15;curve=1;bstyle=rounddot", po.Style(1, 1, 0x4080c0, 0, 0.5, 0.15, 0).with_dash(border_style=5))])
def test_dashed_curve_rules_through_the_action_api(chart, params, st):
    """dash styles along curve=1 ways (round 3): accepted by act_draw_ini and drawn as the oracle's dashing stroker does along the
    decomposed curve (pinned on libcairo: tests/test_oracle_cairo.py::test_dashed_curves)"""
    from test_oracle_cairo import _dash_scene
    from test_gpu_raster import _ways_to_arrays
    sa, proj = chart
    wids, oways = load_ways(sa, proj, _ways_to_arrays(_dash_scene(W, H, 9)))
    rc = hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, params, wids)
    assert rc == 0, sa.lib().smr_hip_last_error()
    want = po.oracle_scene([(st.with_curve(0.35 if "curve_factor" in params else 0.2), oways)], DPI, W, H, 0xfff8e0)
    got = plugin_canvas(sa)
    assert (got != 0xfffff8e0).sum() > 2000
    assert np.array_equal(got, want), int((got != want).sum())


@pytest.mark.parametrize("params,factor", [("color=#4080c0;bcolor=black;bwidth=0.2;curve=1", 0.2),
                                           ("color=#4080c0;bcolor=black;bwidth=0.2;width=0.3;curve=yes;curve_factor=0.35", 0.35)])
def test_curve_rule_through_the_action_api(chart, params, factor):
    """curve=1 / curve_factor= parsed as act_draw_ini does (:625-631); the ways go through cairo_smr_poly_curve's path (:697-749)"""
    from test_oracle_cairo import _dash_scene
    from test_gpu_raster import _ways_to_arrays
    sa, proj = chart
    wids, oways = load_ways(sa, proj, _ways_to_arrays(_dash_scene(W, H, 9)))
    rc = hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, params, wids)
    assert rc == 0, sa.lib().smr_hip_last_error()
    st = po.Style(1, 1, 0x4080c0, 0x000000, 0.3 if "width=0.3" in params else 0, 0.2, 0).with_curve(factor)
    want = po.oracle_scene([(st, oways)], DPI, W, H, 0xfff8e0)
    assert np.array_equal(plugin_canvas(sa), want)


@pytest.mark.parametrize("params,width_mm,length_mm", [("color=#ff00ff;wavy=yes", 0.0, None),                    # rules/available/seamark_area.osm:38 (magenta)
                                                     ("color=#ff00ff;width=.2;wavy=yes;wavy_length=1.5", 0.2, 1.5),    # test/rules/0131.osm
                                                     ("color=#ff00ff;width=.3;wavy=yes;wavy_length=1.5;style=dashed;ignore_open=0", 0.3, 1.5)])   # dashes along the waves (round 3)
def test_wavy_rule_through_the_action_api(chart, params, width_mm, length_mm):
    """wavy=1 / wavy_length= (act_draw_ini :632-642): the plugin resamples the way on the host as cairo_smr_wavy does (:752-840)
    and hands the knots to the GPU as explicit Bezier paths; the expected picture takes the knots from the oracle's own
    restatement (oracle/wavy.c) of the same loop.  The closed way of the rule is filled (cairo_fill closes its open wavy path)."""
    import ctypes as C
    from smrender_amd import synth
    sa, proj = chart
    rng = np.random.default_rng(3)
    xs, ys, off = [], [], [0]
    for _ in range(8):                                   # open ways of a few long legs: submarine cables
        n = int(rng.integers(3, 8))
        xs.append(rng.uniform(20, W - 20) + np.cumsum(rng.normal(0, 90, n))); ys.append(rng.uniform(20, H - 20) + np.cumsum(rng.normal(0, 90, n)))
        off.append(off[-1] + n)
    ring = np.array([100.0, 200, 200, 100, 100]), np.array([100.0, 100, 200, 200, 100])
    x, y = np.concatenate(xs + [ring[0][:4]]), np.concatenate(ys + [ring[1][:4]])
    ref = np.concatenate([np.arange(off[-1]), off[-1] + np.array([0, 1, 2, 3, 0])]).astype(np.int32)
    off = np.array(off + [off[-1] + 5], np.int64)
    wids, _ = load_ways(sa, proj, (x, y, off, ref))
    rc = hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, params, wids)
    assert rc == 0, sa.lib().smr_hip_last_error()
    lat, lon = synth.px_to_geo(proj, x, y)
    dist = 0.0015 if length_mm is None else length_mm / 25.4 * DPI * (proj.wc * np.cos(np.radians((proj.lat_ll + proj.lat_ru) / 2))) / proj.w
    L = po.orc()
    L.orc_wavy_path.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_int]
    op = po.make_proj(45.0, 15.0, 25000.0, float(W), float(H), DPI)
    oways = []
    for w in range(len(off) - 1):                        # the last way is the ring: its knot way ends in one more point = p0
        la, lo = np.ascontiguousarray(lat[ref[off[w]:off[w + 1]]]), np.ascontiguousarray(lon[ref[off[w]:off[w + 1]]])
        ring_w = w == len(off) - 2
        ola, olo = np.zeros(4096), np.zeros(4096)
        m = L.orc_wavy_path(po.P(la), po.P(lo), len(la), dist, po.P(ola), po.P(olo), 4096)
        assert m >= 1
        if m >= 4:
            if ring_w:
                ola[m], olo[m] = ola[0], olo[0]; m += 1
            ox, oy, _, _, _ = po.project(op, ola[:m], olo[:m])
            oways.append((ox, oy, ring_w))
    assert len(oways) >= 4 and oways[-1][2]
    st = po.Style(1, 0, 0xff00ff, 0, width_mm, 0, 0)
    if "style=dashed" in params:
        st = st.with_dash(fill_style=1)
    want = po.oracle_scene([(st.with_knots(), oways)], DPI, W, H, 0xfff8e0)
    got = plugin_canvas(sa)
    assert len(np.unique(got)) > 2
    assert np.array_equal(got, want), int((got != want).sum())


@pytest.mark.parametrize("sea_first", [True, False])
def test_directional_rule_with_page_rectangle(chart, sea_first, keycol=False):
    """directional=1 through the action API (act_draw_ini :673-686, _main :1039-1052, _fini :1084-1140): the way
    with the largest area decides; if it runs clockwise the page rectangle of page_way() becomes the outer ring"""
    from test_oracle_cairo import _nested_scene
    from test_gpu_raster import _ways_to_arrays
    sa, proj = chart
    ways = _nested_scene(W, H, 11)[0 if not sea_first else 1:]     # with / without the outer frame

    def area_on_chart(x, y):            # > 0: counterclockwise with north up (pixel y points down)
        return -0.5 * np.sum(x[:-1] * y[1:] - x[1:] * y[:-1])
    big = max((w for w in ways if w[2]), key=lambda w: abs(area_on_chart(w[0], w[1])))
    if (area_on_chart(big[0], big[1]) < 0) != sea_first:
        # make the largest way clockwise (page rectangle expected) / counterclockwise (none) as the case wants
        ways = [(x[::-1].copy(), y[::-1].copy(), c) if c else (x, y, c) for x, y, c in ways]
    arrs = _ways_to_arrays(ways)
    fill = 0x4080c0
    if keycol:
        # color=%col: the fill takes the colour of the largest way, or, when that one runs clockwise, no colour is set and the
        # source of the last way of the list sorted by area stays: the smallest closed way's (act_draw_fini :1099, :1117-1125)
        cols = [((37 * k + 16) % 256) << 16 | ((91 * k + 32) % 256) << 8 | ((53 * k + 48) % 256) for k in range(len(ways))]
        wids, oways = load_ways(sa, proj, arrs, [f"col=#{c:06x}" for c in cols])
        closed = [k for k, w in enumerate(ways) if w[2]]
        by_area = sorted(closed, key=lambda k: abs(area_on_chart(ways[k][0], ways[k][1])))
        fill = cols[by_area[0]] if sea_first else cols[by_area[-1]]
        rc = hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "color=%col;width=0.3;directional=1", wids)
    else:
        wids, oways = load_ways(sa, proj, arrs)
        rc = hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "color=#4080c0;bcolor=black;bwidth=0.2;width=0.3;directional=1", wids)
    assert rc == 0, sa.lib().smr_hip_last_error()
    if sea_first:
        op = po.make_proj(45.0, 15.0, 25000.0, float(W), float(H), DPI)
        plat = np.array([proj.lat_ru, proj.lat_ll, proj.lat_ll, proj.lat_ru, proj.lat_ru])
        plon = np.array([proj.lon0, proj.lon0, proj.lon_ru, proj.lon_ru, proj.lon0])
        px, py, _, _, _ = po.project(op, plat, plon)
        oways = oways + [(px, py, True)]
    if keycol:
        # open ways are stroked in their own colours; the oracle helper takes per-way colours as (x, y, closed, fill, border)
        oways = [(w[0], w[1], w[2], (cols[k] if k < len(cols) else None), None) if not w[2] else w for k, w in enumerate(oways)]
    want = po.oracle_directional(po.Style(1, 0, fill, 0, 0.3, 0, 0), oways, DPI, W, H, 0xfff8e0)
    got = plugin_canvas(sa)
    assert np.array_equal(got, want), (int((got != want).sum()), [hex(v) for v in np.unique(got[got != want])[:6]], [hex(v) for v in np.unique(want[got != want])[:6]])
    # with the page rectangle the chart's corner is painted; without, it is outside every ring
    assert ((got[0, 0] & 0xffffff) != 0xfff8e0) == sea_first


@pytest.mark.parametrize("sea_first", [True, False])
def test_directional_rule_with_key_colours(chart, sea_first):
    test_directional_rule_with_page_rectangle(chart, sea_first, keycol=True)


def test_directional_wavy_rule_with_page_rectangle(chart):
    """directional=1 together with wavy=1: every closed way (and the page rectangle, which the largest, clockwise way calls for)
    goes through cairo_smr_wavy as a sub-path of ONE fill (act_draw_fini :1084-1140 -> cairo_smr_poly :926-937)"""
    import ctypes as C
    from smrender_amd import synth
    sa, proj = chart
    rings = [(np.array([60.0, 320, 320, 60, 60]), np.array([40.0, 40, 260, 260, 40])),          # big, clockwise on the chart (north up)
             (np.array([120.0, 120, 200, 200, 120]), np.array([100.0, 180, 180, 100, 100])),    # counterclockwise inside it
             (np.array([230.0, 290, 260, 230]), np.array([90.0, 150, 220, 90]))]
    x, y = np.concatenate([r[0][:-1] for r in rings]), np.concatenate([r[1][:-1] for r in rings])
    off, ref = [0], []
    for r in rings:
        n = len(r[0]) - 1
        ref += list(range(off[-1] - len(off) + 1 + 0, off[-1] - len(off) + 1 + n)) + [off[-1] - len(off) + 1]
        off.append(off[-1] + n + 1)
    wids, _ = load_ways(sa, proj, (x, y, np.array(off, np.int64), np.array(ref, np.int32)))
    rc = hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "color=#4080c0;directional=1;wavy=1;wavy_length=2", wids)
    assert rc == 0, sa.lib().smr_hip_last_error()
    lat, lon = synth.px_to_geo(proj, x, y)
    dist = 2.0 / 25.4 * DPI * (proj.wc * np.cos(np.radians((proj.lat_ll + proj.lat_ru) / 2))) / proj.w
    L = po.orc()
    L.orc_wavy_path.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_int]
    op = po.make_proj(45.0, 15.0, 25000.0, float(W), float(H), DPI)
    geo = [(lat[np.array(ref[off[k]:off[k + 1]])], lon[np.array(ref[off[k]:off[k + 1]])]) for k in range(len(rings))]
    geo.append((np.array([proj.lat_ru, proj.lat_ll, proj.lat_ll, proj.lat_ru, proj.lat_ru]),
                np.array([proj.lon0, proj.lon0, proj.lon_ru, proj.lon_ru, proj.lon0])))      # page_way(): NW, SW, SE, NE, NW
    oways = []
    for la, lo in geo:
        ola, olo = np.zeros(8192), np.zeros(8192)
        m = L.orc_wavy_path(po.P(np.ascontiguousarray(la)), po.P(np.ascontiguousarray(lo)), len(la), dist, po.P(ola), po.P(olo), 8192)
        assert m >= 4
        ola[m], olo[m] = ola[0], olo[0]; m += 1
        ox, oy, _, _, _ = po.project(op, ola[:m], olo[:m])
        oways.append((ox, oy, True))
    want = po.oracle_directional(po.Style(1, 0, 0x4080c0, 0, 0, 0, 0).with_knots(), oways, DPI, W, H, 0xfff8e0)
    got = plugin_canvas(sa)
    assert np.array_equal(got, want), int((got != want).sum())
    assert (got[1, 1] & 0xffffff) != 0xfff8e0           # the page rectangle is the outer ring


def oracle_caption_tags(img, names, px, py, size_mm, col, weight):
    """cap_coord for every caption of one rule in object order (src/smrules_cairo.c:2202-2392): geometry through
    the real libcairo, find_angle restated (oracle/caprot.c), and the rule's private background copy that
    collects one filled rectangle per finished caption (auto_ctx, :2376-2382).  Returns [(angle tag, align tag)]
    and the number of captions whose angle the earlier rectangles changed."""
    cd, L = po.cairo(), po.orc()
    oc = L.orc_canvas_new(0, 0, W, H, DPI, 0xffffff)
    bgpix = np.ctypeslib.as_array(C.cast(L.orc_canvas_pixels_rw(oc), C.POINTER(C.c_uint32)), (H, W))
    bgpix[:] = img
    xoff = size_mm * 72 / 25.4 / 2
    crnd = lambda v: int(np.sign(v) * np.floor(abs(v) + 0.5))
    out, changed = [], 0
    for k in range(len(names)):
        geo = np.zeros(3, np.int32)
        cd.cd_caption_geometry(b"serif", size_mm, names[k].encode(), xoff, xoff, 1, W * 72 / DPI, H * 72 / DPI, po.P(geo))
        met = np.zeros(4)
        cd.cd_last_metrics(po.P(met))
        pw, ph, pxx = [int(v) for v in geo]
        fw, fh = crnd(pw * DPI / 72), crnd(ph * DPI / 72)
        fg = np.zeros((fh, fw), np.uint32)
        assert cd.cd_plane(pw, ph, pxx, col, DPI, po.P(fg)) == 0
        r = C.c_int()
        steps = L.orc_num_steps(fw, fh, DPI, C.byref(r)); r = r.value

        def find(bg):
            ox, oy = crnd(px[k] - r / 2.0), crnd(py[k] - r / 2.0)
            cut = np.zeros((r, r), np.uint32)
            xs0, ys0, xs1, ys1 = max(ox, 0), max(oy, 0), min(ox + r, W), min(oy + r, H)
            if xs1 > xs0 and ys1 > ys0:
                cut[ys0 - oy:ys1 - oy, xs0 - ox:xs1 - ox] = bg[ys0:ys1, xs0:xs1]
            return L.orc_find_angle(po.P(cut), r, po.P(fg), fw, fh, steps, 0.0, weight, None)

        a = find(bgpix)
        changed += a != find(img)
        pos_e = True
        if np.pi / 2 < a < 3 * np.pi / 2:
            a -= np.pi
            pos_e = False
        out.append(("%.1f" % np.degrees(a), "east" if pos_e else "west", a))
        # the rectangle: translate(x, y) rotate(a) rectangle(ox, oy, tw + xoff, -ascent); colour of the rule's first object
        tw, asc, desc, th = met
        hgt = max(th, asc + desc)
        ox_, oy_ = (xoff if pos_e else -tw - xoff), hgt / 2
        wr, hr = tw + xoff, -asc
        ca, sn = np.cos(a), np.sin(a)
        xpt, ypt = px[k] * 72 / DPI, py[k] * 72 / DPI
        pts = []
        for ux, uy in ((ox_, oy_), (ox_ + wr, oy_), (ox_ + wr, oy_ + hr), (ox_, oy_ + hr)):
            pts += [(ca * ux + -sn * uy) + xpt, (sn * ux + ca * uy) + ypt]
        L.orc_direct_fill_pt(oc, po.P(np.array(pts)), 4, col)
    L.orc_canvas_free(oc)
    return out, changed


def run_caption_rule(chart, seed, cx, cy, names, size_mm, colname, col, weight):
    sa, proj = chart
    from smrender_amd import synth
    if po.cairo() is None:
        pytest.skip("libcairo (text extents) not loadable")
    wids, oways = load_ways(sa, proj, po.synth_ways(90, W, H, seed=seed))
    assert hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "color=#4080c0;bcolor=black;bwidth=0.1", wids) == 0
    img = plugin_canvas(sa)
    lat, lon = synth.px_to_geo(proj, cx, cy)
    n = len(names)
    base = _next_id[0]; _next_id[0] += n + 1
    nids = base + np.arange(n)
    hs.add_nodes(nids, lat, lon, [f"name={t}" for t in names])
    params = f"key=name;size={size_mm};angle=auto;color={colname};halign=east;font=serif;weight={weight}"
    assert hs.apply_rule(sa.lib(), "cap", hs.OSM_NODE, params, nids) == 0, sa.lib().smr_hip_last_error()
    op = po.make_proj(45.0, 15.0, 25000.0, float(W), float(H), DPI)
    px, py, _, _, _ = po.project(op, lat, lon)
    want, changed = oracle_caption_tags(img, names, px, py, size_mm, col, weight)
    for k in range(n):
        assert hs.get_tag(hs.OSM_NODE, nids[k], "name:autoangle") == want[k][0], k
        assert hs.get_tag(hs.OSM_NODE, nids[k], "name:autoalign") == want[k][1], k
    return changed


def test_cap_autoangle_tags(chart):
    rng = np.random.default_rng(5)
    n = 12
    cx, cy = rng.uniform(20, W - 20, n), rng.uniform(20, H - 20, n)
    names = [f"Cap {k}" if k % 3 else "Longer Caption Text" for k in range(n)]
    run_caption_rule(chart, 79, cx, cy, names, 2.0, "black", 0x000000, 0.7)


def test_cap_same_rule_captions_see_earlier_rectangles(chart):
    """the reference scans a rule's captions one after the other; each leaves a filled rectangle on the
    rule's private background copy (auto_ctx, src/smrules_cairo.c:2376-2382) that later ones see.
    The plugin runs independent captions as one batch and dependent ones level by level."""
    rng = np.random.default_rng(9)
    # two clusters of mutually overlapping captions + a few isolated ones
    cx = np.concatenate([rng.uniform(60, 100, 5), rng.uniform(250, 290, 5), rng.uniform(20, W - 20, 4)])
    cy = np.concatenate([rng.uniform(60, 90, 5), rng.uniform(180, 210, 5), rng.uniform(20, H - 20, 4)])
    changed = run_caption_rule(chart, 91, cx, cy, [f"Name {k}" for k in range(14)], 1.6, "#203040", 0x203040, 1.0)
    assert changed >= 2          # the dependency really changed some angles in this scene


def test_every_shipped_draw_rule_is_accepted_or_refused_for_a_documented_reason(chart):
    """tests/golden/draw_params.txt holds the parameter strings of all `draw` actions of the reference's rulesets and test
    rules (collected by tests/golden/make_draw_params.py).  act_draw_ini must take every one of them (and _fini must run on an
    empty rule): all 82, the two image fills (file=wooded.jpg, the reference's test/wooded.jpg kept as a fixture) included."""
    import os
    sa, proj = chart
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    lines = [l.rstrip("\n") for l in open(os.path.join(golden, "draw_params.txt"))]
    assert len(lines) >= 80 and sum("file=" in l for l in lines) == 2
    cwd = os.getcwd()
    os.chdir(golden)                 # file= names are relative to where smrender runs
    try:
        for params in lines:
            rc = hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, params, [])
            assert rc == 0, (params, rc, sa.lib().smr_hip_last_error())
    finally:
        os.chdir(cwd)


@pytest.mark.parametrize("params,directional", [("file=wooded.jpg;scale=0.6", False), ("file=wooded.jpg;scale=0.6;directional=yes", True),
                                                ("file=wooded.jpg;scale=0.4;transparency=0.35;color=#30c04020;bcolor=black;bwidth=0.15", False)])
def test_shipped_image_fill_rules(chart, params, directional):
    """draw file=wooded.jpg;scale=0.6[;directional=yes] (the two image fills among the shipped rules): the plugin decodes the JPEG,
    scales it as img_ini does (libcairo), and the device fills the closed ways with it.  Expected: the oracle's image fill
    (pinned on libcairo's surface pattern) with the pattern made by the same steps through the test's own libcairo driver."""
    import os
    from test_oracle_cairo import _dash_scene, _nested_scene
    from test_gpu_raster import _ways_to_arrays
    sa, proj = chart
    if po.cairo() is None:
        pytest.skip("libcairo not loadable")
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    scale = float(params.split("scale=")[1].split(";")[0])
    trans = float(params.split("transparency=")[1].split(";")[0]) if "transparency=" in params else 0.0
    pat = po.scaled_image(po.decode_jpeg(os.path.join(golden, "wooded.jpg")), scale, trans)
    ways = _nested_scene(W, H, 4) if directional else _dash_scene(W, H, 9)
    wids, oways = load_ways(sa, proj, _ways_to_arrays(ways))
    cwd = os.getcwd()
    os.chdir(golden)
    try:
        rc = hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, params, wids)
    finally:
        os.chdir(cwd)
    assert rc == 0, sa.lib().smr_hip_last_error()
    if directional:
        st = po.Style(0, 0, 0, 0, 0, 0, 0)
        want = po.oracle_directional(st, [w for w in oways if w[2]], DPI, W, H, 0xfff8e0, pattern=pat)
    else:
        st = po.Style(1, 1, 0x30c04020, 0, 0, 0.15, 0) if "bcolor" in params else po.Style(0, 0, 0, 0, 0, 0, 0)
        want = po.oracle_pattern_scene([], (st, oways, pat), [], DPI, W, H, 0xfff8e0)
    got = plugin_canvas(sa)
    assert (got != 0xfffff8e0).sum() > 5000
    assert np.array_equal(got, want), int((got != want).sum())


def test_img_on_ways_is_an_image_fill(chart, tmp_path):
    """img on ways (act_img_ini :2683-2704, act_img_main -> img_fill :2742-2751): the closed ways filled with the repeated image, open
    ways passed over; a PNG at scale 1 is the pattern pixel for pixel.  A rotated pattern is left to the built-in action."""
    from test_oracle_cairo import _dash_scene, _pattern
    from test_gpu_raster import _ways_to_arrays
    sa, proj = chart
    cd = po.cairo()
    if cd is None:
        pytest.skip("libcairo not loadable")
    pat = _pattern(29, 17, 5, 255)
    png = str(tmp_path / "tile.png")
    assert cd.cd_write_png_data(po.P(pat), 29, 17, png.encode()) == 0
    wids, oways = load_ways(sa, proj, _ways_to_arrays(_dash_scene(W, H, 9)))
    assert hs.apply_rule(sa.lib(), "img", hs.OSM_WAY, f"file={png}", wids) == 0, sa.lib().smr_hip_last_error()
    want = po.oracle_pattern_scene([], (po.Style(0, 0, 0, 0, 0, 0, 0), oways, pat), [], DPI, W, H, 0xfff8e0)
    got = plugin_canvas(sa)
    assert np.array_equal(got, want), int((got != want).sum())
    assert (got != 0xfffff8e0).sum() > 3000
    assert hs.apply_rule(sa.lib(), "img", hs.OSM_WAY, f"file={png};angle=30", wids) == 2001      # "Rule will be ignored"


REAL_RULES = [   # (parameters as a ruleset would give them, the oracle's style, directional)
    ("color=#e0e0e0;bcolor=#0000ff;curve=1", po.Style(1, 1, 0xe0e0e0, 0x0000ff, 0, 0, 0).with_curve(0.2), False),            # rules: color=#e0e0e0;bcolor=blue;curve=1
    ("bcolor=#40FF00FF;bwidth=0.5;bstyle=dashed", po.Style(0, 1, 0, 0x40ff00ff, 0, 0.5, 0).with_dash(border_style=1), False),
    ("color=#dde7eb;ignore_open=1;directional=1;curve=1", po.Style(1, 0, 0xdde7eb, 0, 0, 0, 1).with_curve(0.2), True),
    ("color=#ff00ff;width=.2;wavy=yes;wavy_length=1.5", None, False),                                                           # test/rules/0131.osm
    ("bcolor=#ff00ff;bstyle=pipe;bwidth=0.2", po.Style(0, 1, 0, 0xff00ff, 0, 0.2, 0).with_dash(border_style=4), False),      # pipeline_submarine
    ("color=#00600000;width=0.3;curve=1", po.Style(1, 0, 0x00600000, 0, 0.3, 0, 0).with_curve(0.2), False),
]


@pytest.mark.parametrize("k", range(len(REAL_RULES)))
def test_reference_testdata_through_the_action_api(hip, k):
    """The reference's own test data (test/testdata.osm: 103 nodes, 11 ways, kept as the fixture tests/golden/testdata.osm) on its
    test chart (test/Makefile:21-23: window 15E34.35:43N44.06:40000, page 50x50 mm, 300 dpi = 590 px), one shipped draw rule at
    a time applied to every way in bxtree order (negative ids ascending): what the plugin draws equals the oracle's picture of
    the same ways bit for bit."""
    import ctypes as C, os
    import xml.etree.ElementTree as ET
    sa = hip
    params, st, directional = REAL_RULES[k]
    Wc = Hc = 590
    sa.lib().smr_hip_plugin_reset()
    proj = sa.Projection.from_window(43.0 + 44.06 / 60, 15.0 + 34.35 / 60, 40000.0, float(Wc), float(Hc), DPI)
    hs.set_chart(proj, 0xffffff)
    t = ET.parse(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "testdata.osm")).getroot()
    base = _next_id[0]; _next_id[0] += 100000
    # fresh ids per test (the host's tree lives as long as the process), keeping the file's order of negative ids
    nid = {n.get("id"): base + 50000 + int(n.get("id")) for n in t.findall("node")}
    hs.add_nodes(list(nid.values()), [float(n.get("lat")) for n in t.findall("node")], [float(n.get("lon")) for n in t.findall("node")])
    geo = {n.get("id"): (float(n.get("lat")), float(n.get("lon"))) for n in t.findall("node")}
    ways = sorted(((int(w.get("id")), [nd.get("ref") for nd in w.findall("nd")]) for w in t.findall("way")))
    wids = []
    for wid, refs in ways:
        hs.add_way(base + 50000 + wid, np.array([nid[r] for r in refs], np.int64))
        wids.append(base + 50000 + wid)
    try:
        rc = hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, params, wids)
        assert rc == 0, sa.lib().smr_hip_last_error()
        got = np.empty((Hc, Wc), np.uint32)
        assert sa.lib().smr_hip_canvas_read(sa.lib().smr_hip_plugin_canvas(), 0, 0, Wc, Hc, got.ctypes.data) == 0
    finally:
        sa.lib().smr_hip_plugin_reset()
    op = po.make_proj(43.0 + 44.06 / 60, 15.0 + 34.35 / 60, 40000.0, float(Wc), float(Hc), DPI)
    oways = []
    for wid, refs in ways:
        la, lo = np.array([geo[r][0] for r in refs]), np.array([geo[r][1] for r in refs])
        closed = len(refs) >= 4 and refs[0] == refs[-1]
        if st is None:                                   # wavy: knots from the oracle's restatement of cairo_smr_wavy
            L = po.orc()
            L.orc_wavy_path.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_int]
            dist = 1.5 / 25.4 * DPI * (proj.wc * np.cos(np.radians((proj.lat_ll + proj.lat_ru) / 2))) / proj.w
            ola, olo = np.zeros(8192), np.zeros(8192)
            m = L.orc_wavy_path(po.P(np.ascontiguousarray(la)), po.P(np.ascontiguousarray(lo)), len(la), dist, po.P(ola), po.P(olo), 8192)
            if m < 4:
                continue
            if closed:                                   # a closed way is filled: its knot way ends in one more point = p0
                ola[m], olo[m] = ola[0], olo[0]; m += 1
            la, lo = ola[:m], olo[:m]
        ox, oy, _, _, _ = po.project(op, la, lo)
        oways.append((ox, oy, closed))
    if st is None:
        want = po.oracle_scene([(po.Style(1, 0, 0xff00ff, 0, 0.2, 0, 0).with_knots(), oways)], DPI, Wc, Hc, 0xffffff)
    elif directional:
        want = po.oracle_directional(st, oways, DPI, Wc, Hc, 0xffffff)
    else:
        want = po.oracle_scene([(st, oways)], DPI, Wc, Hc, 0xffffff)
    assert len(np.unique(got)) > 2
    assert np.array_equal(got, want), int((got != want).sum())


def test_cap_on_closed_ways_uses_the_centroid(chart):
    """cap_way (src/smrules_cairo.c:2395-2444): a closed way is captioned at poly_area's centroid; open ways are passed
    over.  The host is the miniature one with the reference's libsmrender; the centroid comes from ITS poly_area."""
    sa, proj = chart
    from smrender_amd import synth
    if po.cairo() is None:
        pytest.skip("libcairo (text extents) not loadable")
    wids, _ = load_ways(sa, proj, po.synth_ways(90, W, H, seed=83))
    assert hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "color=#4080c0;bcolor=black;bwidth=0.1", wids) == 0
    img = plugin_canvas(sa)
    # six named closed ways and two named open ones
    rng = np.random.default_rng(12)
    names, cap_ids, cen = [], [], []
    for k in range(8):
        n = 7
        ccx, ccy = rng.uniform(60, W - 60), rng.uniform(60, H - 60)
        a = np.sort(rng.uniform(0, 2 * np.pi, n))
        x, y = ccx + 25 * rng.uniform(0.6, 1.0, n) * np.cos(a), ccy + 25 * rng.uniform(0.6, 1.0, n) * np.sin(a)
        lat, lon = synth.px_to_geo(proj, x, y)
        base = _next_id[0]; _next_id[0] += n + 2
        ids = base + np.arange(n)
        hs.add_nodes(ids, lat, lon)
        closed = k < 6
        hs.add_way(base + n, np.append(ids, ids[0]) if closed else ids, f"name=Area {k}")
        cap_ids.append(base + n)
        if closed:
            names.append(f"Area {k}")
            clat, clon, _ = hs.poly_area(base + n)
            cen.append((clat, clon))
    params = "key=name;size=2.0;angle=auto;color=black;halign=east;font=serif;weight=1.0"
    assert hs.apply_rule(sa.lib(), "cap", hs.OSM_WAY, params, cap_ids) == 0, sa.lib().smr_hip_last_error()
    op = po.make_proj(45.0, 15.0, 25000.0, float(W), float(H), DPI)
    px, py, _, _, _ = po.project(op, np.array([c[0] for c in cen]), np.array([c[1] for c in cen]))
    want, _ = oracle_caption_tags(img, names, px, py, 2.0, 0x000000, 1.0)
    for k in range(6):
        assert hs.get_tag(hs.OSM_WAY, cap_ids[k], "name:autoangle") == want[k][0], k
        assert hs.get_tag(hs.OSM_WAY, cap_ids[k], "name:autoalign") == want[k][1], k
    for k in (6, 7):
        assert hs.get_tag(hs.OSM_WAY, cap_ids[k], "name:autoangle") is None          # open ways get no caption


def test_cap_glyphs_are_composited_onto_the_device_canvas(chart):
    """SURVEY 8f rank 1 for captions: after the scan the plugin's cap draws the text itself (cap_coord :2363-2388): host
    libcairo + FreeType rasterise every caption on the chart's pixel grid, smr_hip_canvas_blit paints the images OVER the
    device canvas.  Compared with the real libcairo recording and replaying the same chart: draw rule, then one rule with
    cairo_show_text per caption at the angle the scan chose.  hide=1 leaves the canvas as it was (only the tags)."""
    sa, proj = chart
    from smrender_amd import synth
    from golden_io import within
    cd = po.cairo()
    if cd is None:
        pytest.skip("libcairo not loadable")
    arrs = po.synth_ways(70, W, H, seed=85)
    wids, oways = load_ways(sa, proj, arrs)
    assert hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "color=#4080c0;bcolor=black;bwidth=0.1", wids) == 0
    before = plugin_canvas(sa)
    rng = np.random.default_rng(31)
    n = 10
    cx, cy = rng.uniform(40, W - 90, n), rng.uniform(30, H - 30, n)
    names = [f"Caption {k}" for k in range(n)]
    lat, lon = synth.px_to_geo(proj, cx, cy)
    size_mm, col = 2.5, 0x802010
    for hide in (1, 0):
        base = _next_id[0]; _next_id[0] += n + 1
        nids = base + np.arange(n)
        hs.add_nodes(nids, lat, lon, [f"name={t}" for t in names])
        params = f"key=name;size={size_mm};angle=auto;color=#802010;halign=east;font=serif;weight=1.0;hide={hide}"
        assert hs.apply_rule(sa.lib(), "cap", hs.OSM_NODE, params, nids) == 0, sa.lib().smr_hip_last_error()
        got = plugin_canvas(sa)
        if hide:
            assert np.array_equal(got, before)
    op = po.make_proj(45.0, 15.0, 25000.0, float(W), float(H), DPI)
    px, py, _, _, _ = po.project(op, lat, lon)
    xoff = size_mm * 72 / 25.4 / 2

    # the angles as find_angle returns them (the tags hold them rounded to 0.1 degree, which moves glyphs by whole pixels)
    tags, _ = oracle_caption_tags(before, names, px, py, size_mm, col, 1.0)

    def captions(L):
        L.cd_rule_begin()
        for k in range(n):
            assert hs.get_tag(hs.OSM_NODE, nids[k], "name:autoangle") == tags[k][0]
            a, east = tags[k][2], tags[k][1] == "east"
            geo = np.zeros(3, np.int32)
            L.cd_caption_geometry(b"serif", size_mm, names[k].encode(), xoff, xoff, 1, W * 72 / DPI, H * 72 / DPI, po.P(geo))
            met = np.zeros(4); L.cd_last_metrics(po.P(met))
            tw, asc, desc, th = met
            hgt = max(th, asc + desc)
            ox, oy = (xoff if east else -tw - xoff), hgt / 2          # pos_offset (:1478-1508): E/W aligned, vertically centred
            assert L.cd_show_caption(b"serif", size_mm, names[k].encode(), px[k] * 72 / DPI, py[k] * 72 / DPI, a, ox, oy, col) == 0
        L.cd_rule_end()

    want = po.cairo_scene([(po.Style(1, 1, 0x4080c0, 0, 0, 0.1, 0), oways)], DPI, W, H, 0xfff8e0, before_render=captions)
    plain = po.cairo_scene([(po.Style(1, 1, 0x4080c0, 0, 0, 0.1, 0), oways)], DPI, W, H, 0xfff8e0)
    text = np.any(po.chan_diff(want, plain) > 0, axis=-1) if po.chan_diff(want, plain).ndim == 3 else po.chan_diff(want, plain) > 0
    assert text.sum() > 2000                                   # the captions really are in the reference picture
    assert within(got, want) >= 0.995                          # north star: +-2/255 on >= 99.5 % of the pixels
    # and where the text is: the glyph pixels themselves agree (same FreeType masks, same positions)
    d = po.chan_diff(got, want)
    d = d.max(axis=-1) if d.ndim == 3 else d
    assert (d[text] <= 2).mean() >= 0.97, float((d[text] <= 2).mean())


def test_cap_on_a_banded_plugin_canvas_is_refused(chart, monkeypatch):
    """SMR_HIP_CUTOUT=crop (this fixture): the crop of a caption near a band boundary needs the neighbour's rows; the plugin has
    nobody to ask, so a banded plugin canvas draws but refuses cap ('Two canvases'); dist.py is the banded crop
    path.  With the default cut-outs a banded canvas scans: test_cap_on_a_banded_plugin_canvas_with_the_references_cutouts"""
    sa, proj = chart
    monkeypatch.setenv("SMR_HIP_BAND_Y0", "100")
    monkeypatch.setenv("SMR_HIP_BAND_H", "100")
    try:
        wids, oways = load_ways(sa, proj, po.synth_ways(30, W, H, seed=3))
        assert hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "color=#4080c0", wids) == 0
        want = po.oracle_scene([(po.Style(1, 0, 0x4080c0, 0, 0, 0, 0), oways)], DPI, W, H, 0xfff8e0)
        got = np.empty((100, W), np.uint32)
        assert sa.lib().smr_hip_canvas_read(sa.lib().smr_hip_plugin_canvas(), 0, 0, W, 100, got.ctypes.data) == 0      # band-local rows
        assert np.array_equal(got, want[100:200])
        from smrender_amd import synth
        lat, lon = synth.px_to_geo(proj, np.array([150.0]), np.array([150.0]))
        nid = _next_id[0]; _next_id[0] += 2
        hs.add_nodes(np.array([nid]), lat, lon, ["name=Band"])
        assert hs.apply_rule(sa.lib(), "cap", hs.OSM_NODE, "key=name;size=2;angle=auto;color=black", np.array([nid])) != 0
        assert hs.get_tag(hs.OSM_NODE, nid, "name:autoangle") is None
    finally:
        monkeypatch.undo()
        sa.lib().smr_hip_plugin_reset()


def test_hipsave_writes_the_plugin_canvas(chart, tmp_path):
    """hipsave@libsmr_hip.so:file=... -- the chart's PNG comes from the device canvas (smr_hip_canvas_write_png)"""
    sa, proj = chart
    wids, oways = load_ways(sa, proj, po.synth_ways(60, W, H, seed=77))
    assert hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "color=#4080c0;bcolor=black;bwidth=0.1", wids) == 0
    f = tmp_path / "chart.png"
    assert hs.apply_rule(sa.lib(), "hipsave", hs.OSM_NODE, f"file={f}", np.zeros(0, np.int64)) == 0, sa.lib().smr_hip_last_error()
    assert np.array_equal(po.png_decode(f), po.png_pixels(plugin_canvas(sa)))


def test_hipsave_writes_a_kap_chart(chart, tmp_path):
    """hipsave@libsmr_hip.so:file=chart.kap -- what `smrender -k` writes (save_kap, src/smkap.c:196-245): the header of
    gen_kap_header (:143-167) with the corners un-projected, then palette, rows and offset table from the device canvas"""
    import ctypes as C
    sa, proj = chart
    wids, oways = load_ways(sa, proj, po.synth_ways(60, W, H, seed=79))
    assert hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "color=#4080c0;bcolor=#203040;bwidth=0.3", wids) == 0
    f = tmp_path / "chart.kap"
    assert hs.apply_rule(sa.lib(), "hipsave", hs.OSM_NODE, f"file={f}", np.zeros(0, np.int64)) == 0, sa.lib().smr_hip_last_error()
    data = f.read_bytes()
    cut = data.index(b"OST/1\r\n")
    header = data[:cut].decode()
    lines = header.split("\r\n")
    assert lines[1] == "VER/2.0" and f"RA={W},{H},DU={DPI}" in lines[3] and "PR=MERCATOR" in lines[4] and lines[-2] == "CPH/0.0" and lines[-1] == ""
    op = po.make_proj(45.0, 15.0, 25000.0, float(W), float(H), DPI)
    corners = {"REF/1": (0, 0), "REF/2": (W, H), "REF/3": (0, H), "PLY/1": (0, 0), "PLY/2": (W, 0), "PLY/3": (W, H), "PLY/4": (0, H)}
    seen = 0
    for ln in lines:
        key = ln.split(",")[0]
        if key in corners:
            lon, lat = C.c_double(), C.c_double()
            po.orc().orc_pxf2geo(C.byref(op), float(corners[key][0]), float(corners[key][1]), C.byref(lon), C.byref(lat))
            got_lat, got_lon = [float(v) for v in ln.split(",")[-2:]]
            assert abs(got_lat - lat.value) < 2e-8 and abs(got_lon - lon.value) < 2e-8, ln
            seen += 1
    assert seen == 7
    body, pal = po.kap_body(plugin_canvas(sa), cut)
    assert data[cut:] == body.tobytes()


def symbol_png(path, w=37, h=23):
    """an asymmetric, partly translucent symbol written by libcairo's own PNG writer"""
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.where((xx + yy) % 7 < 4, 255, 120).astype(np.uint32)
    a[(xx - w * 0.3) ** 2 + (yy - h * 0.5) ** 2 > (h * 0.62) ** 2] = 0
    r, g, b = (200 * xx // w).astype(np.uint32), (60 + 150 * yy // h).astype(np.uint32), np.full((h, w), 40, np.uint32)
    img = (a << 24 | (r * a // 255) << 16 | (g * a // 255) << 8 | (b * a // 255)).astype(np.uint32)
    assert po.cairo().cd_write_png_data(po.P(img), w, h, str(path).encode()) == 0
    return path


def test_img_symbols_are_composited_onto_the_device_canvas(chart, tmp_path):
    """SURVEY 8f rank 1 for symbols: img@libsmr_hip.so on nodes (act_img_*, src/smrules_cairo.c:2512-2871).  The symbol is
    decoded and scaled by the host's libcairo as in img_ini, every placement recorded as in img_place, replayed onto the chart's
    pixel grid and painted OVER the device canvas.  Compared with the real libcairo recording and replaying the same chart
    (draw rule, then the img rule), for fixed angles (angle, anglekey, alignkey, scale, transparency) and for angle=auto, where
    the angle is find_angle with the symbol as the plane (oracle/caprot.c on the canvas before the rule)."""
    sa, proj = chart
    from smrender_amd import synth
    from golden_io import within
    cd = po.cairo()
    if cd is None:
        pytest.skip("libcairo not loadable")
    f = symbol_png(tmp_path / "sym.png")
    arrs = po.synth_ways(70, W, H, seed=85)
    rng = np.random.default_rng(77)
    n = 9
    cx, cy = rng.uniform(30, W - 30, n), rng.uniform(30, H - 30, n)
    lat, lon = synth.px_to_geo(proj, cx, cy)
    op = po.make_proj(45.0, 15.0, 25000.0, float(W), float(H), DPI)
    px, py, _, _, _ = po.project(op, lat, lon)
    style = po.Style(1, 1, 0x4080c0, 0, 0, 0.1, 0)
    for mode in ("fixed", "auto"):
        sa.lib().smr_hip_plugin_reset()
        wids, oways = load_ways(sa, proj, arrs)
        assert hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "color=#4080c0;bcolor=black;bwidth=0.1", wids) == 0
        before = plugin_canvas(sa)
        base = _next_id[0]; _next_id[0] += n + 1
        nids = base + np.arange(n)
        scale, trans = (1.5, 0.25) if mode == "fixed" else (1.0, 0.0)
        wh = np.zeros(2)
        assert cd.cd_img_load_png(str(f).encode(), scale, trans, po.P(wh)) == 0
        chk = np.zeros((int(wh[1]), int(wh[0])), np.uint32)
        assert cd.cd_img_pixels(po.P(chk)) == 0
        assert int((chk >> 24).max()) in ((191, 192) if mode == "fixed" else (255,))      # transparency=0.25 leaves 75 % (DEST_OUT)
        if mode == "fixed":
            tags = [("" if k % 3 == 0 else f"dir={17.5 * k}") + (";side=west" if k % 4 == 1 else ";side=east") for k in range(n)]
            tags = [t.lstrip(";") for t in tags]
            hs.add_nodes(nids, lat, lon, tags)
            params = f"file={f};scale=1.5;transparency=0.25;angle=30;anglekey=dir;alignkey=side"
            ang = [np.radians(17.5 * k if k % 3 else 0.0) + (np.pi / 2 if k % 4 == 1 else 0.0) + np.radians(360 - 30) for k in range(n)]
        else:
            hs.add_nodes(nids, lat, lon)
            params = f"file={f};angle=auto;weight=0.8"
            fw, fh = int(wh[0]), int(wh[1])
            fg = np.zeros((fh, fw), np.uint32)
            assert cd.cd_img_pixels(po.P(fg)) == 0
            L = po.orc()
            r = C.c_int()
            steps = L.orc_num_steps(fw, fh, DPI, C.byref(r)); r = r.value
            crnd = lambda v: int(np.sign(v) * np.floor(abs(v) + 0.5))
            ang = []
            for k in range(n):
                ox, oy = crnd(px[k] - r / 2.0), crnd(py[k] - r / 2.0)
                cut = np.zeros((r, r), np.uint32)
                xs0, ys0, xs1, ys1 = max(ox, 0), max(oy, 0), min(ox + r, W), min(oy + r, H)
                cut[ys0 - oy:ys1 - oy, xs0 - ox:xs1 - ox] = before[ys0:ys1, xs0:xs1]
                ang.append(L.orc_find_angle(po.P(cut), r, po.P(fg), fw, fh, steps, 0.0, 0.8, None))
            assert len(set(np.round(ang, 6))) >= 3             # the scan really chose different angles
        assert hs.apply_rule(sa.lib(), "img", hs.OSM_NODE, params, nids) == 0, sa.lib().smr_hip_last_error()
        got = plugin_canvas(sa)

        def symbols(Lc):
            Lc.cd_rule_begin()
            for k in range(n):
                assert Lc.cd_img_place(px[k], py[k], ang[k], float(DPI)) == 0
            Lc.cd_rule_end()

        want = po.cairo_scene([(style, oways)], DPI, W, H, 0xfff8e0, before_render=symbols)
        plain = po.cairo_scene([(style, oways)], DPI, W, H, 0xfff8e0)
        d = po.chan_diff(want, plain)
        sym = (d.max(axis=-1) if d.ndim == 3 else d) > 0
        assert sym.sum() > 1500, mode                           # the symbols really are in the reference picture
        assert within(got, want) >= 0.995, (mode, within(got, want))
        e = po.chan_diff(got, want)
        e = e.max(axis=-1) if e.ndim == 3 else e
        assert (e[sym] <= 2).mean() >= 0.99, (mode, float((e[sym] <= 2).mean()))


def test_img_svg_symbol_and_refusals(chart, tmp_path):
    sa, proj = chart
    from smrender_amd import synth
    if po.cairo() is None:
        pytest.skip("libcairo not loadable")
    nid = _next_id[0]; _next_id[0] += 2
    lat, lon = synth.px_to_geo(proj, np.array([100.0]), np.array([80.0]))
    hs.add_nodes(np.array([nid]), lat, lon)
    # a file that is not there (PNG or JPEG), no file at all: refused like the reference; an SVG as a way's pattern: left to the built-in img
    assert hs.apply_rule(sa.lib(), "img", hs.OSM_NODE, "file=symbol.jpg", np.array([nid])) != 0
    assert hs.apply_rule(sa.lib(), "img", hs.OSM_NODE, "scale=2", np.array([nid])) != 0
    assert hs.apply_rule(sa.lib(), "img", hs.OSM_NODE, f"file={tmp_path / 'missing.png'}", np.array([nid])) != 0
    # an SVG symbol through librsvg: a 24 x 12 px red rectangle, centred on a node at a pixel corner
    svg = tmp_path / "rect.svg"
    svg.write_text('<svg xmlns="http://www.w3.org/2000/svg" width="24px" height="12px" viewBox="0 0 24 12">'
                   '<rect x="0" y="0" width="24" height="12" fill="#ff0000"/></svg>')
    before = plugin_canvas(sa)
    rc = hs.apply_rule(sa.lib(), "img", hs.OSM_NODE, f"file={svg}", np.array([nid]))
    if rc != 0:
        pytest.skip("librsvg not loadable on this box: " + sa.lib().smr_hip_last_error().decode())
    wids, _ = load_ways(sa, proj, po.synth_ways(5, W, H, seed=1))
    assert hs.apply_rule(sa.lib(), "img", hs.OSM_WAY, f"file={svg}", wids) == 2001
    got = plugin_canvas(sa)
    op = po.make_proj(45.0, 15.0, 25000.0, float(W), float(H), DPI)
    px, py, _, _, _ = po.project(op, lat, lon)
    x0, y0 = int(round(px[0])) - 12, int(round(py[0])) - 6
    assert np.all(got[y0 + 2:y0 + 10, x0 + 2:x0 + 22] == 0xffff0000)      # the interior is the symbol's red
    changed = got != before
    assert changed.sum() >= 24 * 12 - 40 and changed.sum() <= 26 * 14
    assert not changed[:y0 - 2].any() and not changed[y0 + 15:].any()


def test_cap_with_given_angles_and_background_boxes(chart):
    """cap with angle= / anglekey= / alignkey= / fillcolor= (cap_coord's else branch, src/smrules_cairo.c:2351-2388, cap_rect
    :2188-2199): no scan, no tags; box and text of every caption go onto the device canvas.  Compared with libcairo recording
    and replaying the same chart.  The reference's own chain works on the device canvas too: a hidden angle=auto rule writes the
    tags, a second rule with anglekey=name:autoangle;alignkey=name:autoalign draws."""
    sa, proj = chart
    from smrender_amd import synth
    from golden_io import within
    cd = po.cairo()
    if cd is None:
        pytest.skip("libcairo not loadable")
    arrs = po.synth_ways(70, W, H, seed=85)
    wids, oways = load_ways(sa, proj, arrs)
    assert hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "color=#4080c0;bcolor=black;bwidth=0.1", wids) == 0
    rng = np.random.default_rng(41)
    n = 8
    cx, cy = rng.uniform(60, W - 60, n), rng.uniform(40, H - 40, n)
    names = [f"Name {k}" for k in range(n)]
    lat, lon = synth.px_to_geo(proj, cx, cy)
    base = _next_id[0]; _next_id[0] += n + 1
    nids = base + np.arange(n)
    sides = ["east", "west", "center", "east", "northwest", "south", "west", "east"]
    dirs = [None if k % 3 == 0 else 23.0 * k for k in range(n)]
    hs.add_nodes(nids, lat, lon, [f"name={names[k]};side={sides[k]}" + (f";dir={dirs[k]}" if dirs[k] is not None else "") for k in range(n)])
    size_mm, col, fillcol = 2.2, 0x102060, 0x40ffffff
    params = f"key=name;size={size_mm};angle=15;anglekey=dir;alignkey=side;color=#102060;fillcolor=#40ffffff;font=serif"
    assert hs.apply_rule(sa.lib(), "cap", hs.OSM_NODE, params, nids) == 0, sa.lib().smr_hip_last_error()
    assert hs.get_tag(hs.OSM_NODE, nids[0], "name:autoangle") is None
    got = plugin_canvas(sa)
    op = po.make_proj(45.0, 15.0, 25000.0, float(W), float(H), DPI)
    px, py, _, _, _ = po.project(op, lat, lon)
    xoff = size_mm * 72 / 25.4 / 2
    POS = {"east": 4, "west": 8, "center": 0, "northwest": 8 | 1, "south": 2}

    def captions(L):
        L.cd_rule_begin()
        for k in range(n):
            a = np.radians(dirs[k] if dirs[k] is not None else 0.0) + np.radians(360 - 15)
            geo = np.zeros(3, np.int32)
            L.cd_caption_geometry(b"serif", size_mm, names[k].encode(), xoff, xoff, 1, W * 72 / DPI, H * 72 / DPI, po.P(geo))
            met = np.zeros(4); L.cd_last_metrics(po.P(met))
            tw, asc, desc, th = met
            hgt = max(th, asc + desc)
            pos = POS[sides[k]]
            oy = -xoff if pos & 3 == 1 else hgt + xoff if pos & 3 == 2 else hgt / 2              # pos_offset (:1478-1508)
            ox = xoff if pos & 12 == 4 else -tw - xoff if pos & 12 == 8 else -tw / 2
            assert L.cd_caption_box(px[k] * 72 / DPI, py[k] * 72 / DPI, a, ox, oy, tw, -hgt, desc, 1.2, fillcol) == 0
            assert L.cd_show_caption(b"serif", size_mm, names[k].encode(), px[k] * 72 / DPI, py[k] * 72 / DPI, a, ox, oy, col) == 0
        L.cd_rule_end()

    style = po.Style(1, 1, 0x4080c0, 0, 0, 0.1, 0)
    want = po.cairo_scene([(style, oways)], DPI, W, H, 0xfff8e0, before_render=captions)
    plain = po.cairo_scene([(style, oways)], DPI, W, H, 0xfff8e0)
    d = po.chan_diff(want, plain)
    ink = (d.max(axis=-1) if d.ndim == 3 else d) > 0
    assert ink.sum() > 4000
    assert within(got, want) >= 0.995, within(got, want)
    e = po.chan_diff(got, want)
    e = e.max(axis=-1) if e.ndim == 3 else e
    assert (e[ink] <= 2).mean() >= 0.97, float((e[ink] <= 2).mean())

    # the reference's chain, both rules on the device canvas
    base = _next_id[0]; _next_id[0] += n + 1
    nids2 = base + np.arange(n)
    hs.add_nodes(nids2, lat, lon, [f"name={t}" for t in names])
    assert hs.apply_rule(sa.lib(), "cap", hs.OSM_NODE, "key=name;size=2;angle=auto;halign=east;hide=1", nids2) == 0
    mid = plugin_canvas(sa)
    assert np.array_equal(mid, got)
    assert hs.get_tag(hs.OSM_NODE, nids2[0], "name:autoangle") is not None
    assert hs.apply_rule(sa.lib(), "cap", hs.OSM_NODE, "key=name;size=2;anglekey=name:autoangle;alignkey=name:autoalign;color=black", nids2) == 0
    assert (plugin_canvas(sa) != mid).sum() > 1000


def _area_axis_angle(lat, lon):
    """cap_way's angle=majoraxis (src/smrules_cairo.c:2431-2441) over area_axis (:1183-1240), farthest_node (:1151-1180) and
    coord_diffp (src/smath.c:34-52), restated for the test: hop to the farthest node until the distance stops growing; the
    last hop's bearing as a math angle, turned to read upright."""
    def diff(i, j):
        dlat = lat[j] - lat[i]
        dlon = (lon[j] - lon[i]) * np.cos(np.radians((lat[i] + lat[j]) / 2))
        b = np.fmod(np.degrees(np.arctan2(dlon, dlat)), 360.0)
        b = b + 360 if b < 0 else b
        c = np.sin(np.radians(lat[i])) * np.sin(np.radians(lat[j])) + np.cos(np.radians(lat[i])) * np.cos(np.radians(lat[j])) * np.cos(np.radians(lon[j] - lon[i]))
        return np.degrees(np.arccos(c)), b
    last, fd, fb = 0, 0.0, 0.0
    ang = 720.0
    while True:
        best, bd, bb = -1, 0.0, 0.0
        for j in range(len(lat)):
            d, b = diff(last, j)
            if d > bd:
                best, bd, bb = j, d, b
        if best == -1:
            break
        if best == 0 or bd <= fd:
            ang = fb
            break
        last, fd, fb = best, bd, bb
    a = np.fmod(90 - ang, 360.0)
    a = a + 360 if a < 0 else a
    return a - 180.0 if 90 < a <= 270 else a


def test_cap_along_the_major_axis_of_closed_ways(chart):
    """cap angle=majoraxis (parse_auto_rot src/smrparse.c:1064-1066, cap_way :2431-2441): the caption of a closed way runs along
    its longest diagonal.  Long thin areas at known headings; the picture is compared with libcairo drawing the same captions at
    the angles of the restated area_axis, and those angles with the headings the areas were made at."""
    sa, proj = chart
    from smrender_amd import synth
    from golden_io import within
    cd = po.cairo()
    if cd is None:
        pytest.skip("libcairo not loadable")
    arrs = po.synth_ways(40, W, H, seed=87)
    wids, oways = load_ways(sa, proj, arrs)
    assert hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "color=#4080c0;bcolor=black;bwidth=0.1", wids) == 0
    rng = np.random.default_rng(77)
    n, m = 6, 16
    heads = [10.0, 75.0, 120.0, 200.0, 265.0, 330.0]           # math angles of the long axis, degrees, y up
    cap_ids, names, cen, want_a = [], [], [], []
    for k in range(n):
        ccx, ccy = 90 + (k % 3) * (W - 180) / 2, 90 + (k // 3) * (H - 180)
        t = np.arange(m) * 2 * np.pi / m + 0.1
        ex, ey = 60 * np.cos(t), 12 * np.sin(t)
        h = np.radians(heads[k])
        x, y = ccx + ex * np.cos(h) - ey * np.sin(h), ccy - (ex * np.sin(h) + ey * np.cos(h))
        lat, lon = synth.px_to_geo(proj, x, y)
        base = _next_id[0]; _next_id[0] += m + 2
        ids = base + np.arange(m)
        hs.add_nodes(ids, lat, lon)
        hs.add_way(base + m, np.append(ids, ids[0]), f"name=Axis {k}")
        cap_ids.append(base + m); names.append(f"Axis {k}")
        clat, clon, _ = hs.poly_area(base + m)
        cen.append((clat, clon))
        a = _area_axis_angle(np.append(lat, lat[0]), np.append(lon, lon[0]))
        want_a.append(a)
        up = (heads[k] + 90) % 180 - 90                          # the heading folded into (-90, 90]
        assert abs((a - up + 90) % 180 - 90) < 8, (k, a, up)     # 16 nodes: the diagonal is within a node's spacing of the axis
    size_mm, col = 2.0, 0x000000
    params = f"key=name;size={size_mm};angle=majoraxis;color=black;halign=center;font=serif"
    before = plugin_canvas(sa)
    assert hs.apply_rule(sa.lib(), "cap", hs.OSM_WAY, params, cap_ids) == 0, sa.lib().smr_hip_last_error()
    got = plugin_canvas(sa)
    assert hs.get_tag(hs.OSM_WAY, cap_ids[0], "name:autoangle") is None
    op = po.make_proj(45.0, 15.0, 25000.0, float(W), float(H), DPI)
    px, py, _, _, _ = po.project(op, np.array([c[0] for c in cen]), np.array([c[1] for c in cen]))
    xoff = size_mm * 72 / 25.4 / 2

    def captions(L):
        L.cd_rule_begin()
        for k in range(n):
            a = np.radians(360 - want_a[k])
            geo = np.zeros(3, np.int32)
            L.cd_caption_geometry(b"serif", size_mm, names[k].encode(), xoff, xoff, 1, W * 72 / DPI, H * 72 / DPI, po.P(geo))
            met = np.zeros(4); L.cd_last_metrics(po.P(met))
            tw, asc, desc, th = met
            hgt = max(th, asc + desc)
            assert L.cd_show_caption(b"serif", size_mm, names[k].encode(), px[k] * 72 / DPI, py[k] * 72 / DPI, a, -tw / 2, hgt / 2, col) == 0
        L.cd_rule_end()

    style = po.Style(1, 1, 0x4080c0, 0, 0, 0.1, 0)
    want = po.cairo_scene([(style, oways)], DPI, W, H, 0xfff8e0, before_render=captions)
    assert (got != before).sum() > 1500
    assert within(got, want) >= 0.995, within(got, want)
    e = po.chan_diff(got, want)
    e = e.max(axis=-1) if e.ndim == 3 else e
    ink = got != before
    assert (e[ink] <= 2).mean() >= 0.97, float((e[ink] <= 2).mean())

    # a node rule with angle=majoraxis turns by a full turn (cap_coord :2356 with cap->angle == 720)
    base = _next_id[0]; _next_id[0] += 2
    lat, lon = synth.px_to_geo(proj, np.array([W / 2.0]), np.array([H / 2.0]))
    hs.add_nodes(np.array([base]), lat, lon, ["name=Turn"])
    assert hs.apply_rule(sa.lib(), "cap", hs.OSM_NODE, "key=name;size=2;angle=majoraxis;color=black;font=serif", np.array([base])) == 0


def test_cap_fontbox_adds_the_captions_boxes_to_the_hosts_data(chart):
    """cap fontbox=1 (cap_coord :2371-2374, cap_geo_rect src/smrules_cairo.c:2114-2176): every caption leaves a closed way of four
    new nodes in the host's data -- the text's box on the page, scaled by bgbox_scale about its middle and NOT rotated -- with
    the object's tags and smrender:fontbox=yes; made through the host's own malloc_way / malloc_node / put_object."""
    sa, proj = chart
    from smrender_amd import synth
    cd = po.cairo()
    if cd is None:
        pytest.skip("libcairo (text extents) not loadable")
    n = 3
    cx, cy = np.array([80.3, 200.7, 310.1]), np.array([60.2, 150.9, 90.4])
    lat, lon = synth.px_to_geo(proj, cx, cy)
    base = _next_id[0]; _next_id[0] += n + 1
    nids = base + np.arange(n)
    names = ["Box A", "Second", "C"]
    hs.add_nodes(nids, lat, lon, [f"name={t};kind=test" for t in names])
    size_mm, scale = 2.5, 1.5
    node0, way0 = hs.next_ids()
    params = f"key=name;size={size_mm};angle=30;color=black;halign=east;valign=north;font=serif;fontbox=1;bgbox_scale={scale};hide=1"
    assert hs.apply_rule(sa.lib(), "cap", hs.OSM_NODE, params, nids) == 0, sa.lib().smr_hip_last_error()
    op = po.make_proj(45.0, 15.0, 25000.0, float(W), float(H), DPI)
    px, py, _, _, _ = po.project(op, lat, lon)
    xoff = size_mm * 72 / 25.4 / 2
    for k in range(n):
        refs = hs.way_refs(way0 - k)
        assert refs is not None and len(refs) == 5 and refs[4] == refs[0], k
        assert list(refs[:4]) == [node0 - 4 * k - i for i in range(4)]
        assert hs.get_tag(hs.OSM_WAY, way0 - k, "smrender:fontbox") == "yes"
        assert hs.get_tag(hs.OSM_WAY, way0 - k, "name") == names[k] and hs.get_tag(hs.OSM_WAY, way0 - k, "kind") == "test"
        assert hs.get_tag(hs.OSM_WAY, way0 - k, "generator") == "smrender"
        assert hs.get_tag(hs.OSM_NODE, refs[2], "smrender:fontbox") == "yes" and hs.get_tag(hs.OSM_NODE, refs[2], "name") == names[k]
        geo = np.zeros(3, np.int32)
        cd.cd_caption_geometry(b"serif", size_mm, names[k].encode(), xoff, xoff, 1, W * 72 / DPI, H * 72 / DPI, po.P(geo))
        met = np.zeros(4); cd.cd_last_metrics(po.P(met))
        tw, asc, desc, th = met
        hgt = max(th, asc + desc)
        # pos_offset (:1478-1508) for east + north, then cap_geo_rect's corners in pt
        X, Y = px[k] * 72 / DPI + xoff, py[k] * 72 / DPI - xoff
        wdt, h = tw, -hgt
        ws, hsc = wdt * scale, h * scale
        x0, y0 = X + (wdt - ws) / 2, Y + (h - hsc) / 2 + desc
        want = [(x0, y0), (x0 + ws, y0), (x0 + ws, y0 + hsc), (x0, y0 + hsc)]
        for i in range(4):
            la, lo, _ = hs.node_pos(refs[i])
            gx, gy, _, _, _ = po.project(op, np.array([la]), np.array([lo]))
            assert abs(gx[0] - want[i][0] * DPI / 72) < 1e-6 and abs(gy[0] - want[i][1] * DPI / 72) < 1e-6, (k, i)


def test_cap_mkarea_adds_the_scans_diagram_to_the_hosts_data(chart):
    """cap angle=auto;mkarea=1 (parse_auto_rot src/smrparse.c:1062, find_angle :1999-2000, dv_mkarea :1772-1801): per caption a closed
    way of num_steps new nodes around the anchor, node i at r * dv_quant[i] in the direction of trial angle i, tagged
    smrender:autorot:angle=<bearing>;<percent>.  dv_quant from the oracle's find_angle on the same cut-out (crop convention of this
    chart fixture)."""
    sa, proj = chart
    from smrender_amd import synth
    cd, L = po.cairo(), po.orc()
    if cd is None:
        pytest.skip("libcairo (text extents) not loadable")
    wids, oways = load_ways(sa, proj, po.synth_ways(90, W, H, seed=91))
    assert hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "color=#4080c0;bcolor=black;bwidth=0.1", wids) == 0
    img = plugin_canvas(sa)
    n = 2
    cx, cy = np.array([120.4, 260.8]), np.array([100.3, 170.6])
    lat, lon = synth.px_to_geo(proj, cx, cy)
    base = _next_id[0]; _next_id[0] += n + 1
    nids = base + np.arange(n)
    names = ["Area One", "Two"]
    hs.add_nodes(nids, lat, lon, [f"name={t}" for t in names])
    size_mm, weight, phase = 2.0, 0.6, 20.0
    node0, way0 = hs.next_ids()
    params = f"key=name;size={size_mm};angle=auto;mkarea=1;color=black;halign=east;font=serif;weight={weight};phase={phase};hide=1"
    assert hs.apply_rule(sa.lib(), "cap", hs.OSM_NODE, params, nids) == 0, sa.lib().smr_hip_last_error()
    op = po.make_proj(45.0, 15.0, 25000.0, float(W), float(H), DPI)
    px, py, _, _, _ = po.project(op, lat, lon)
    xoff = size_mm * 72 / 25.4 / 2
    crnd = lambda v: int(np.sign(v) * np.floor(abs(v) + 0.5))
    nid = node0
    for k in range(n):
        geo = np.zeros(3, np.int32)
        cd.cd_caption_geometry(b"serif", size_mm, names[k].encode(), xoff, xoff, 1, W * 72 / DPI, H * 72 / DPI, po.P(geo))
        pw, ph, pxx = [int(v) for v in geo]
        fw, fh = crnd(pw * DPI / 72), crnd(ph * DPI / 72)
        fg = np.zeros((fh, fw), np.uint32)
        assert cd.cd_plane(pw, ph, pxx, 0, DPI, po.P(fg)) == 0
        r = C.c_int()
        steps = L.orc_num_steps(fw, fh, DPI, C.byref(r)); r = r.value
        ox, oy = crnd(px[k] - r / 2.0), crnd(py[k] - r / 2.0)
        cut = np.zeros((r, r), np.uint32)
        xs0, ys0, xs1, ys1 = max(ox, 0), max(oy, 0), min(ox + r, W), min(oy + r, H)
        cut[ys0 - oy:ys1 - oy, xs0 - ox:xs1 - ox] = img[ys0:ys1, xs0:xs1]
        dv = np.zeros(steps)
        L.orc_find_angle(po.P(cut), r, po.P(fg), fw, fh, steps, phase, weight, po.P(dv))
        # dv_weight (:1813-1821), dv_quantize (:1868-1890)
        ang = np.zeros(steps); a = 0.0
        for i in range(steps):
            ang[i] = a; a += 2 * np.pi / steps
        d = dv * (1.0 - (1.0 - abs(weight)) * (1.0 - np.cos(ang * 2.0 + np.radians(phase))) / 2.0)
        q = (d - d.min()) / (d.max() - d.min())
        refs = hs.way_refs(way0 - k)
        assert refs is not None and len(refs) == steps + 1 and refs[-1] == refs[0], (k, None if refs is None else len(refs), steps)
        assert list(refs[:steps]) == [nid - i for i in range(steps)]
        nid -= steps
        rr = np.hypot(fw, fh) * 72 / DPI              # find_angle's r in pt, added to px as the reference does (its FIXME)
        for i in range(steps):
            la, lo, ntag = hs.node_pos(refs[i])
            assert ntag == 2
            gx, gy, _, _, _ = po.project(op, np.array([la]), np.array([lo]))
            assert abs(gx[0] - (px[k] + rr * q[i] * np.cos(2 * np.pi - ang[i]))) < 1e-6, (k, i)
            assert abs(gy[0] - (py[k] + rr * q[i] * np.sin(2 * np.pi - ang[i]))) < 1e-6, (k, i)
            b = np.fmod(np.degrees(np.pi / 2 - ang[i]), 360.0)
            b = b + 360 if b < 0 else b
            assert hs.get_tag(hs.OSM_NODE, refs[i], "smrender:autorot:angle") == "%.1f;%.1f" % (b, q[i] * 100), (k, i)
        assert q.max() == 1.0 and q.min() == 0.0 and len(np.unique(q)) > steps // 2


def reference_caption_tags(rules, names, px, py, size_mm, col, weight, bg=0xfff8e0):
    """cap_coord the reference's way with the REAL libcairo: the chart's rules recorded (oracle/cairo_drive.c), then per caption, in
    object order, cairo_smr_cut_out's replay at the anchor's sub-pixel offset (cd_cut_out, src/smrules_cairo.c:1511-1533),
    find_angle's metric on it (oracle/caprot.c, pinned against libcairo's rotated paint), the flip, and the caption's rectangle
    filled onto the recording like auto_ctx does (:2376-2382), where the next caption's cut-out sees it.  rules = [(Style, ways)]."""
    cd, L = po.cairo(), po.orc()
    po.cairo_scene(rules, DPI, W, H, bg)                       # leaves the recording in the driver
    xoff = size_mm * 72 / 25.4 / 2
    crnd = lambda v: int(np.sign(v) * np.floor(abs(v) + 0.5))
    out = []
    for k in range(len(names)):
        geo = np.zeros(3, np.int32)
        cd.cd_caption_geometry(b"serif", size_mm, names[k].encode(), xoff, xoff, 1, W * 72 / DPI, H * 72 / DPI, po.P(geo))
        met = np.zeros(4)
        cd.cd_last_metrics(po.P(met))
        pw, ph, pxx = [int(v) for v in geo]
        fw, fh = crnd(pw * DPI / 72), crnd(ph * DPI / 72)
        fg = np.zeros((fh, fw), np.uint32)
        assert cd.cd_plane(pw, ph, pxx, col, DPI, po.P(fg)) == 0
        r = C.c_int()
        steps = L.orc_num_steps(fw, fh, DPI, C.byref(r)); r = r.value
        r_pt = np.hypot(fw, fh) * 72 / DPI
        sub = np.zeros((r + 2) * (r + 2), np.uint32)
        assert cd.cd_cut_out(px[k] * 72 / DPI, py[k] * 72 / DPI, r_pt, float(DPI), po.P(sub), len(sub)) == r
        cut = np.ascontiguousarray(sub[:r * r].reshape(r, r))
        a = L.orc_find_angle(po.P(cut), r, po.P(fg), fw, fh, steps, 0.0, weight, None)
        pos_e = True
        if np.pi / 2 < a < 3 * np.pi / 2:
            a -= np.pi
            pos_e = False
        out.append(("%.1f" % np.degrees(a), "east" if pos_e else "west", a))
        tw, asc, desc, th = met
        hgt = max(th, asc + desc)
        ox_, oy_ = (xoff if pos_e else -tw - xoff), hgt / 2
        wr, hr = tw + xoff, -asc
        ca, sn = np.cos(a), np.sin(a)
        xpt, ypt = px[k] * 72 / DPI, py[k] * 72 / DPI
        pts = [((ca * ux + -sn * uy) + xpt, (sn * ux + ca * uy) + ypt) for ux, uy in ((ox_, oy_), (ox_ + wr, oy_), (ox_ + wr, oy_ + hr), (ox_, oy_ + hr))]
        pts = np.ascontiguousarray(np.array(pts + pts[:1]))
        assert cd.cd_rule_begin() == 0                          # an opaque fill in a group of its own == the same fill without a group
        cd.cd_poly_line(po.P(pts), 5)
        cd.cd_fill(col)
        assert cd.cd_rule_end() == 0
    return out


def test_cap_tags_from_the_references_own_cutouts(chart_replay):
    """the plugin's default: cap's cut-outs are the chart's rules rendered again at each anchor's sub-pixel offset (the plugin keeps
    every drawn rule for that), so the tags are the reference's for anchors anywhere between pixel centres -- here two draw
    rules, then clusters of captions whose rectangles change each other's angles (dependency levels, windows of several sizes
    in one atlas), checked against find_angle on libcairo's own cut-outs, caption after caption"""
    sa, proj = chart_replay
    from smrender_amd import synth
    if po.cairo() is None:
        pytest.skip("libcairo (text extents) not loadable")
    a1, a2 = po.synth_ways(70, W, H, seed=91), po.synth_ways(25, W, H, seed=92)
    w1, o1 = load_ways(sa, proj, a1)
    w2, o2 = load_ways(sa, proj, a2)
    assert hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "color=#4080c0;bcolor=black;bwidth=0.1", w1) == 0
    assert hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "bcolor=#a03020;bwidth=0.3", w2) == 0
    rng = np.random.default_rng(9)
    cx = np.concatenate([rng.uniform(60, 100, 5), rng.uniform(250, 290, 5), rng.uniform(20, W - 20, 6)])
    cy = np.concatenate([rng.uniform(60, 90, 5), rng.uniform(180, 210, 5), rng.uniform(20, H - 20, 6)])
    names = [f"Name {k}" if k % 4 else "A Longer Name %d" % k for k in range(16)]
    lat, lon = synth.px_to_geo(proj, cx, cy)
    n = len(names)
    base = _next_id[0]; _next_id[0] += n + 1
    nids = base + np.arange(n)
    hs.add_nodes(nids, lat, lon, [f"name={t}" for t in names])
    assert hs.apply_rule(sa.lib(), "cap", hs.OSM_NODE, "key=name;size=1.6;angle=auto;color=#203040;halign=east;font=serif;weight=1.0;hide=1",
                         nids) == 0, sa.lib().smr_hip_last_error()
    op = po.make_proj(45.0, 15.0, 25000.0, float(W), float(H), DPI)
    px, py, _, _, _ = po.project(op, lat, lon)
    rules = [(po.Style(1, 1, 0x4080c0, 0, 0, 0.1, 0), o1), (po.Style(0, 1, 0, 0xa03020, 0, 0.3, 0), o2)]
    want = reference_caption_tags(rules, names, px, py, 1.6, 0x203040, 1.0)
    got = [hs.get_tag(hs.OSM_NODE, nids[k], "name:autoangle") for k in range(n)]
    bad = [k for k in range(n) if got[k] != want[k][0] or hs.get_tag(hs.OSM_NODE, nids[k], "name:autoalign") != want[k][1]]
    assert not bad, [(k, got[k], want[k][0]) for k in bad]
    # and the crop convention gives other tags for some of them: the two paths really differ on this chart
    crop, _ = oracle_caption_tags(plugin_canvas(sa), names, px, py, 1.6, 0x203040, 1.0)
    assert sum(crop[k][0] != want[k][0] for k in range(n)) >= 1


def test_cap_on_a_banded_plugin_canvas_with_the_references_cutouts(chart_replay, monkeypatch):
    """a rank of a chart rendered in row bands (SMR_HIP_BAND_Y0/_H) runs the whole rule set over the whole object tree; with the
    default cut-outs (rendered again from the kept rules, not cropped from the canvas) its cap rules need nothing of the other
    bands: the tags are those of the one-canvas run, and its rows, glyphs included, are the chart's"""
    sa, proj = chart_replay
    from smrender_amd import synth
    if po.cairo() is None:
        pytest.skip("libcairo (text extents) not loadable")
    a1 = po.synth_ways(70, W, H, seed=191)
    rng = np.random.default_rng(19)
    cx, cy = rng.uniform(30, W - 30, 12), np.concatenate([rng.uniform(30, H - 30, 8), [118.0, 122.5, 176.0, 181.0]])      # some astride the band's edges
    lat, lon = synth.px_to_geo(proj, cx, cy)
    names = [f"Band {k}" for k in range(12)]
    params = "key=name;size=1.6;angle=auto;color=#203040;halign=east;font=serif;weight=1.0"
    runs = []
    for band in (None, (120, 60)):
        sa.lib().smr_hip_plugin_reset()
        if band:
            monkeypatch.setenv("SMR_HIP_BAND_Y0", str(band[0])); monkeypatch.setenv("SMR_HIP_BAND_H", str(band[1]))
        hs.set_chart(proj, 0xfff8e0)
        w1, _ = load_ways(sa, proj, a1)
        base = _next_id[0]; _next_id[0] += 20
        nids = base + np.arange(12)
        hs.add_nodes(nids, lat, lon, [f"name={t}" for t in names])
        try:
            assert hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "color=#4080c0;bcolor=black;bwidth=0.1", w1) == 0
            assert hs.apply_rule(sa.lib(), "cap", hs.OSM_NODE, params, nids) == 0, sa.lib().smr_hip_last_error()
            tags = [(hs.get_tag(hs.OSM_NODE, i, "name:autoangle"), hs.get_tag(hs.OSM_NODE, i, "name:autoalign")) for i in nids]
            h = band[1] if band else H
            pix = np.empty((h, W), np.uint32)
            assert sa.lib().smr_hip_canvas_read(sa.lib().smr_hip_plugin_canvas(), 0, 0, W, h, pix.ctypes.data) == 0
            runs.append((tags, pix))
        finally:
            if band:
                monkeypatch.undo()
                sa.lib().smr_hip_plugin_reset()
    (t_full, p_full), (t_band, p_band) = runs
    assert all(t[0] is not None for t in t_full) and len(set(t[0] for t in t_full)) > 3
    assert t_band == t_full
    assert np.array_equal(p_band, p_full[120:180])
    assert (p_full[120:180] != 0xfffff8e0).any()


def test_a_way_with_a_node_far_off_the_chart_is_clipped_not_refused(chart):
    """a coastline or ferry line that runs on for a thousand kilometres past a large-scale chart: libsmr_hip's edge arithmetic is
    exact for edges of less than 139 000 px of height, and smr_hip_draw refuses a batch with a node more than 32 768 px beyond
    the chart; the reference draws the way.  The plugin clips such ways to that box before they reach the device (far_clip):
    the rule is drawn, and what is seen of the far ways is the line they are.  The expectation is the oracle's picture of the
    SAME lines with their far ends pulled in to 20 000 px (along the line: the visible part is the same): with the far ends
    where they are libcairo's own scan converter overflows its 64-bit products (an edge of more than ~130 000 px in both
    directions) and paints a horizontal line instead of the sloped one, which the oracle, restating that arithmetic, does too."""
    sa, proj = chart
    from smrender_amd import synth
    # a closed way with one corner 1.5 M px east, an open way that crosses the chart and ends 2 M px north-west, near ways around them
    fx = np.array([100.0, 300.0, 1.5e6, 100.0]); fy = np.array([50.0, 250.0, 150.0, 50.0])
    ox_ = np.array([350.0, 200.0, 60.0, -2.0e6]); oy_ = np.array([280.0, 200.0, 120.0, -1.2e6])
    near = po.synth_ways(30, W, H, seed=5)
    ways_px = [(fx, fy, True), (ox_, oy_, False)]
    base = _next_id[0]; _next_id[0] += 1000
    wids = []
    k = base
    for x, y, closed in ways_px:
        n = len(x) - 1 if closed else len(x)
        lat, lon = synth.px_to_geo(proj, x[:n], y[:n])
        ids = np.arange(k, k + n); k += n
        hs.add_nodes(ids, lat, lon)
        hs.add_way(k, np.append(ids, ids[0]) if closed else ids); wids.append(k); k += 1
    nwids, noways = load_ways(sa, proj, near)
    wids = wids + list(nwids)
    assert hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "color=#4080c0;bcolor=black;bwidth=0.2", wids) == 0, sa.lib().smr_hip_last_error()
    got = plugin_canvas(sa)
    st = po.Style(1, 1, 0x4080c0, 0x000000, 0, 0.2, 0)

    def pulled(ax, ay, bx, by, dist=20000.0):        # the point of the segment a -> b that lies `dist` px from a
        t = dist / np.hypot(bx - ax, by - ay)
        return ax + t * (bx - ax), ay + t * (by - ay)
    # the triangle's far corner becomes two points, one on each of its sides (the same two lines across the chart)
    p1, p2 = pulled(300.0, 250.0, 1.5e6, 150.0), pulled(100.0, 50.0, 1.5e6, 150.0)
    q = pulled(60.0, 120.0, -2.0e6, -1.2e6)
    far = [(np.array([100.0, 300.0, p1[0], p2[0], 100.0]), np.array([50.0, 250.0, p1[1], p2[1], 50.0]), True),
           (np.array([350.0, 200.0, 60.0, q[0]]), np.array([280.0, 200.0, 120.0, q[1]]), False)]
    want = po.oracle_scene([(st, far + noways)], DPI, W, H, 0xfff8e0)
    d = po.chan_diff(got, want)
    assert (got != 0xfffff8e0).sum() > 20000                       # the far triangle covers a good part of the chart
    if os.environ.get("SMR_TEST_DUMP") and (d > 0).sum() > 40:
        np.savez(os.path.join(os.environ["SMR_TEST_DUMP"], "far.npz"), got=got, want=want)
    # the cuts are new end points: the lines keep their direction to within the 24.8 rounding of those (an AA level here and there)
    assert (d > 0).sum() <= 40 and d.max() <= 8, (int((d > 0).sum()), int(d.max()))
