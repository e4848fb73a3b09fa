"""Kernel 1 (proj_mercator) through the C-ABI vs the oracle restatement of geo2pxf
(src/rdata.c:221-225), round() indices (:253-254) and is_on_page (:482-487)."""
import numpy as np
import pytest
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
REL_TOL = 1e-9      # north_star: projected coordinates within 1e-9 relative


def _setup(sa, mean_lat, mean_lon, scale, w, h, dpi):
    p = sa.Projection.from_window(mean_lat, mean_lon, scale, w, h, dpi)
    sa.set_projection(p)
    op = po.make_proj(mean_lat, mean_lon, scale, w, h, dpi)
    for f in ("lon0", "w", "h", "wc", "lath", "lath_len", "lat_ll", "lon_ru", "lat_ru"):
        assert getattr(p, f) == getattr(op, f), f     # host restatement of init_bbox_mll is bit-equal
    return p, op


@pytest.mark.parametrize("n", [0, 1, 2, 3, 1000, 100001, 1 << 20])
def test_project_matches_oracle(hip, n):
    sa = hip
    p, op = _setup(sa, 43.9, 15.5, 20000.0, 2480.3, 3507.9, 300)
    rng = np.random.default_rng(n + 1)
    lat = rng.uniform(op.lat_ll - 0.05, op.lat_ru + 0.05, n)
    lon = rng.uniform(op.lon0 - 0.05, op.lon_ru + 0.05, n)
    d_lat, d_lon = sa.DeviceArray.from_numpy(lat), sa.DeviceArray.from_numpy(lon)
    d_x, d_y = sa.DeviceArray((n,), np.float64), sa.DeviceArray((n,), np.float64)
    d_ix, d_iy = sa.DeviceArray((n,), np.int32), sa.DeviceArray((n,), np.int32)
    d_on = sa.DeviceArray((n,), np.uint8)
    sa.project(d_lat, d_lon, d_x, d_y, d_ix, d_iy, d_on)
    sa.sync()
    if n == 0:
        return
    x, y, ix, iy, on = po.project(op, lat, lon)
    gx, gy = d_x.numpy(), d_y.numpy()
    scale = max(op.w, op.h)
    assert np.max(np.abs(gx - x)) <= REL_TOL * scale
    assert np.max(np.abs(gy - y)) <= REL_TOL * scale
    # pixel indices bit-exact (ties at .5 would need |x - k - .5| < 1e-9 px: none in seeded data)
    assert np.array_equal(d_ix.numpy(), ix)
    assert np.array_equal(d_iy.numpy(), iy)
    assert np.array_equal(d_on.numpy(), on)


def test_project_world_range_and_roundtrip(hip):
    sa = hip
    p, op = _setup(sa, 45.0, 20.0, 5.0e6, 32768.0, 32768.0, 300)
    rng = np.random.default_rng(7)
    n = 200000
    lat = rng.uniform(-85.0, 85.0, n)
    lon = rng.uniform(-180, 180, n)
    d_lat, d_lon = sa.DeviceArray.from_numpy(lat), sa.DeviceArray.from_numpy(lon)
    d_x, d_y = sa.DeviceArray((n,), np.float64), sa.DeviceArray((n,), np.float64)
    sa.project(d_lat, d_lon, d_x, d_y)
    x, y, _, _, _ = po.project(op, lat, lon)
    gx, gy = d_x.numpy(), d_y.numpy()
    assert np.max(np.abs(gx - x) / np.maximum(1.0, np.abs(x))) <= REL_TOL
    assert np.max(np.abs(gy - y) / np.maximum(1.0, np.abs(y))) <= REL_TOL
    # pxf2geo(geo2pxf(p)) == p  (src/rdata.c:68-72 is the inverse)
    d_lon2, d_lat2 = sa.DeviceArray((n,), np.float64), sa.DeviceArray((n,), np.float64)
    sa.unproject(d_x, d_y, d_lon2, d_lat2)
    assert np.max(np.abs(d_lon2.numpy() - lon)) < 1e-9
    assert np.max(np.abs(d_lat2.numpy() - lat)) < 1e-9


def test_unproject_matches_oracle(hip):
    """k_unproject (pxf2geo, src/rdata.c:68-72) against the oracle's restatement point by point, not only through the round
    trip: arbitrary pixel positions, on and beside the chart"""
    import ctypes as C
    sa = hip
    p, op = _setup(sa, 43.9, 15.5, 20000.0, 2480.3, 3507.9, 300)
    rng = np.random.default_rng(21)
    n = 50000
    x, y = rng.uniform(-500, 3000, n), rng.uniform(-500, 4000, n)
    d_lon, d_lat = sa.DeviceArray((n,), np.float64), sa.DeviceArray((n,), np.float64)
    sa.unproject(sa.DeviceArray.from_numpy(x), sa.DeviceArray.from_numpy(y), d_lon, d_lat)
    glon, glat = d_lon.numpy(), d_lat.numpy()
    lo, la = C.c_double(), C.c_double()
    L = po.orc()
    for i in range(0, n, 7):
        L.orc_pxf2geo(C.byref(op), x[i], y[i], C.byref(lo), C.byref(la))
        assert abs(glon[i] - lo.value) <= 1e-12 * max(1.0, abs(lo.value)) and abs(glat[i] - la.value) <= 1e-12 * max(1.0, abs(la.value)), i


def test_project_rejects_misaligned_arrays(hip):
    sa = hip
    d = sa.DeviceArray((8,), np.float64)
    rc = sa.lib().smr_hip_project(d.ptr + 8, d.ptr, 2, d.ptr, d.ptr, None, None, None)
    assert rc == -2 and b"aligned" in sa.lib().smr_hip_last_error()
