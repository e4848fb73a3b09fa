"""Projection restatement (oracle): the reference holds no numeric fixtures for geo2pxf and
src/rdata.c cannot be built stand-alone (needs the autoconf config.h), so it is pinned by the
identities the reference itself relies on."""
import numpy as np
from oracle import pyoracle as po
import ctypes as C


def test_window_constants_and_corner_identities():
    # test/Makefile:15: window 15E34.35:43N44.06:10000 on a 50x50 mm page at 300 dpi
    w = h = 50 * 300 / 25.4
    p = po.make_proj(43 + 44.06 / 60, 15 + 34.35 / 60, 10000.0, w, h, 300)
    # init_bbox_mll: the chart is wc degrees wide, centred on mean_lon (src/smrender.c:285-287)
    assert abs((p.lon0 + p.lon_ru) / 2 - (15 + 34.35 / 60)) < 1e-12
    assert abs(p.lon_ru - p.lon0 - p.wc) < 1e-12
    x = C.c_double()
    y = C.c_double()
    po.orc().orc_geo2pxf(C.byref(p), p.lon0, p.lat_ru, C.byref(x), C.byref(y))       # upper left -> (0,0)
    assert abs(x.value) < 1e-9 and abs(y.value) < 1e-6
    po.orc().orc_geo2pxf(C.byref(p), p.lon_ru, p.lat_ll, C.byref(x), C.byref(y))     # lower right -> (w,h)
    assert abs(x.value - w) < 1e-9 and abs(y.value - h) < 1e-6
    # scale 1:10000 -> page width in nm: mean_lat_len * 60 nm == 50 mm * 10000 / 1852 m
    mean_lat_len = p.wc * np.cos(np.radians(43 + 44.06 / 60))
    assert abs(mean_lat_len * 60 * 1852 - 0.05 * 10000) < 1e-9


def test_roundtrip_and_indices():
    p = po.make_proj(45.0, 15.0, 100000.0, 16384.0, 16384.0, 300)
    rng = np.random.default_rng(3)
    lat, lon = rng.uniform(p.lat_ll, p.lat_ru, 5000), rng.uniform(p.lon0, p.lon_ru, 5000)
    x, y, ix, iy, on = po.project(p, lat, lon)
    assert on.all() and x.min() >= 0 and x.max() <= 16384 and y.min() >= -1e-6 and y.max() <= 16384 + 1e-6
    assert np.array_equal(ix, np.floor(x + 0.5).astype(np.int32))       # C round(), values are positive
    lo, la = C.c_double(), C.c_double()
    for i in range(0, 5000, 97):
        po.orc().orc_pxf2geo(C.byref(p), x[i], y[i], C.byref(lo), C.byref(la))
        assert abs(lo.value - lon[i]) < 1e-10 and abs(la.value - lat[i]) < 1e-10
    _, _, _, _, on2 = po.project(p, np.array([p.lat_ru + 1e-6, 45.0]), np.array([15.0, p.lon0 - 1e-6]))
    assert not on2.any()


def test_formula_against_50_digit_arithmetic():
    """geo2pxf's Mercator branch (src/rdata.c:221-225) and pxf2geo (:68-72) evaluated with mpmath at 50 digits: the oracle's
    double-precision evaluation stays within 1e-9 relative of the exact value of the FORMULA over the world range, so the
    north star's 1e-9 bar is about the implementation, not about the conditioning of tan/asinh near the poles."""
    import mpmath as mp
    mp.mp.dps = 50
    p = po.make_proj(45.0, 20.0, 5.0e6, 32768.0, 32768.0, 300)
    rng = np.random.default_rng(11)
    lat = np.concatenate([rng.uniform(-85.0, 85.0, 300), [-85.05, 85.05, 0.0, p.lat_ll, p.lat_ru]])
    lon = np.concatenate([rng.uniform(-180, 180, 300), [-180.0, 180.0, 0.0, p.lon0, p.lon_ru]])
    x, y, _, _, _ = po.project(p, lat, lon)
    lath, lath_len, lon0, w, h, wc = (mp.mpf(float(v)) for v in (p.lath, p.lath_len, p.lon0, p.w, p.h, p.wc))
    lo, la = C.c_double(), C.c_double()
    for i in range(len(lat)):
        ex = (mp.mpf(float(lon[i])) - lon0) * w / wc
        ey = h * (mp.mpf("0.5") - (mp.asinh(mp.tan(mp.mpf(float(lat[i])) * mp.pi / 180)) - lath) / lath_len)
        assert abs(mp.mpf(float(x[i])) - ex) <= mp.mpf("1e-9") * max(1, abs(ex)), (i, lat[i], lon[i])
        assert abs(mp.mpf(float(y[i])) - ey) <= mp.mpf("1e-9") * max(1, abs(ey)), (i, lat[i], lon[i])
        # the inverse at the exact image point
        po.orc().orc_pxf2geo(C.byref(p), float(ex), float(ey), C.byref(lo), C.byref(la))
        elat = mp.atan(mp.sinh(lath_len * (mp.mpf("0.5") - mp.mpf(float(ey)) / h) + lath)) * 180 / mp.pi
        assert abs(mp.mpf(la.value) - elat) < mp.mpf("1e-9") and abs(lo.value - lon[i]) < 1e-9
