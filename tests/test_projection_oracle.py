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
