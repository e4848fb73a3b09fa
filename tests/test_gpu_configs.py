"""BASELINE.json's configurations that no other GPU test runs at their stated shape: C1 (A4 at 300 dpi through the action API,
the reference's own test data standing in for the Monaco extract, which is not in the image) and one rank's share of C5
(100 M nodes / 10 M ways on 65 536^2 over 8 GPUs: 12.5 M nodes projected with their clip codes, a band of 65 536 x 8 192 px
with 1.25 M ways).  C2 and C3 at full size: tests/test_gpu_golden.py; C4: bench.py's caption_scan and tests/test_gpu_caprot.py."""
import os
import numpy as np
import pytest
from oracle import pyoracle as po
import hostsim as hs
import c1_scene as c1
from golden_io import draw_style, within

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not hs.available(), reason="oracle/_ref not built")
def test_c1_a4_chart_through_the_action_api(hip, tmp_path):
    """draw@libsmr_hip.so rule after rule as smrender's executor would call them (miniature host on the reference's own
    libsmrender), then hipsave: the PNG's pixels are the oracle's bit for bit and libcairo's own render of the same rules
    (tests/golden/c1_testdata_a4.npz, made by tests/golden/make_c1_fixture.py) within the north star's tolerance"""
    sa = hip
    sa.lib().smr_hip_plugin_reset()
    proj = sa.Projection.from_window(c1.MEAN_LAT, c1.MEAN_LON, c1.SCALE, c1.W_PX, c1.H_PX, c1.DPI)
    hs.set_chart(proj, c1.BG)
    geo, ways = c1.load()
    base = 7000000
    nid = {k: base + 50000 + int(k) for k in geo}
    hs.add_nodes(list(nid.values()), [geo[k][0] for k in geo], [geo[k][1] for k in geo])
    wids = []
    for wid, refs in ways:
        hs.add_way(base + 50000 + wid, np.array([nid[r] for r in refs], np.int64))
        wids.append(base + 50000 + wid)
    f = tmp_path / "c1.png"
    try:
        for params, _ in c1.RULES:
            assert hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, params, wids) == 0, (params, sa.lib().smr_hip_last_error())
        assert hs.apply_rule(sa.lib(), "hipsave", hs.OSM_NODE, f"file={f}", np.zeros(0, np.int64)) == 0, sa.lib().smr_hip_last_error()
        got = np.empty((c1.H, c1.W), np.uint32)
        assert sa.lib().smr_hip_canvas_read(sa.lib().smr_hip_plugin_canvas(), 0, 0, c1.W, c1.H, got.ctypes.data) == 0
    finally:
        sa.lib().smr_hip_plugin_reset()
    assert got.shape == (3508, 2480)
    want = po.oracle_scene([(st, c1.oracle_ways(geo, ways)) for _, st in c1.RULES], c1.DPI, c1.W, c1.H, c1.BG)
    assert np.array_equal(got, want), int((got != want).sum())
    cairo = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c1_testdata_a4.npz"))["image"]
    assert (cairo != cairo[0, 0]).sum() > 2000000                      # a fifth of the page is painted
    assert within(got, cairo, 2) >= 0.9999 and within(got, cairo, 0) >= 0.9999
    assert np.array_equal(po.png_decode(f), po.png_pixels(got))          # and the file holds the canvas


def _c5_rank_share(rank, n_ways, W, HT, BH, seed):
    """ways of ~10 refs, 30 % closed, centred in (and a little beyond) the rank's band"""
    rng = np.random.default_rng(seed)
    n = np.clip(rng.poisson(10, n_ways), 2, 64)
    closed = rng.uniform(size=n_ways) < 0.3
    n = np.where(closed, np.maximum(n, 4), n)
    off = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    cx, cy = rng.uniform(0, W, n_ways), rng.uniform(rank * BH - 40, (rank + 1) * BH + 40, n_ways)
    r = np.exp(rng.uniform(np.log(4), np.log(64), n_ways))
    way = np.repeat(np.arange(n_ways), n)
    k = np.arange(off[-1]) - off[way]
    ang = 2 * np.pi * k / np.where(closed, n - 1, n)[way] + rng.uniform(0, 6.28, n_ways)[way]
    rad = r[way] * rng.uniform(0.7, 1.0, off[-1])
    x, y = cx[way] + rad * np.cos(ang), cy[way] + rad * np.sin(ang)
    last = off[1:] - 1
    ref = np.arange(off[-1], dtype=np.int32)
    ref[last[closed]] = off[:-1][closed]              # a closed way's last ref names its first node
    return x, y, off, ref, cx, cy


def test_c5_one_ranks_share(hip):
    """rank 3 of 8: 12.5 M nodes through smr_hip_project (coordinates within 1e-9 relative, pixel indices and clip codes exact
    against the oracle), then its band of the 65 536^2 chart with 1.25 M ways, fill + border, drawn from the projected
    coordinates: windows at both ends of the band and in its middle equal the oracle's render of the ways that reach them"""
    sa = hip
    from smrender_amd import synth
    W = HT = 65536
    BH, rank, dpi = 8192, 3, 300
    n_ways = 1250000
    x, y, off, ref, cx, cy = _c5_rank_share(rank, n_ways, W, HT, BH, seed=5)
    p = sa.Projection.from_window(45.0, 20.0, 5.6e5, float(W), float(HT), dpi)
    sa.set_projection(p)
    op = po.make_proj(45.0, 20.0, 5.6e5, float(W), float(HT), dpi)
    lat, lon = synth.px_to_geo(p, x, y)
    assert len(lat) > 12400000
    D = sa.DeviceArray.from_numpy
    d_lat, d_lon = D(lat), D(lon)
    d_x, d_y = sa.DeviceArray(lat.shape, np.float64), sa.DeviceArray(lat.shape, np.float64)
    d_ix, d_iy, d_on = sa.DeviceArray(lat.shape, np.int32), sa.DeviceArray(lat.shape, np.int32), sa.DeviceArray(lat.shape, np.uint8)
    sa.project(d_lat, d_lon, d_x, d_y, d_ix, d_iy, d_on)
    sa.sync()
    ox, oy, oix, oiy, oon = po.project(op, lat, lon)
    gx, gy = d_x.numpy(), d_y.numpy()
    assert np.max(np.abs(gx - ox)) <= 1e-9 * W and np.max(np.abs(gy - oy)) <= 1e-9 * HT
    assert np.array_equal(d_ix.numpy(), oix) and np.array_equal(d_iy.numpy(), oiy)
    assert np.array_equal(d_on.numpy(), oon) and 0 < int(oon.sum()) < len(oon)       # ways hang over the chart's sides: both codes occur
    st = po.Style(1, 1, 0x4080c0, 0x000000, 0.1, 25.4 / dpi, 0)
    cv = sa.Canvas(W, HT, dpi, 0xffffff, rank * BH, BH)
    cv.draw(draw_style(sa, st), d_x, d_y, D(off), D(ref), n_ways, len(ref))
    assert sa.lib().smr_hip_draw_status() == 0
    S = 192
    y0 = rank * BH
    for wx, wy in ((0, y0), (W - S, y0 + BH - S), (31000, y0 + 4000), (40000, y0), (12345, y0 + BH - S)):
        got = cv.read(wx, wy - y0, S, S)
        keep = (cx > wx - 150) & (cx < wx + S + 150) & (cy > wy - 150) & (cy < wy + S + 150)
        sx, sy, soff, sref = synth.select_ways(gx, gy, off, ref, keep)
        want = po.oracle_scene([(st, po.ways_from_arrays(sx, sy, soff, sref))], dpi, S, S, 0xffffff, wx, wy, image=(0, 0, W, HT))
        assert len(np.unique(want)) > 3
        assert np.array_equal(got, want), (wx, wy, int((got != want).sum()))
    cv.close()
