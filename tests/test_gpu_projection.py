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


@pytest.mark.gpu
def test_poly_area_matches_the_hosts(hip):
    """smr_hip_poly_area against poly_area as the miniature host runs it (oracle/hostsim.c: the reference's loop, src/smfunc.c:
    377-432, over the reference's own libsmrender objects; pinned there by an independent shoelace): area and centroid of
    closed ways as far as the formula defines them (see below), orientation sign exact, open ways area 0 / NaN centroid"""
    import hostsim as hs
    if not hs.available():
        pytest.skip("oracle/_ref not built")
    sa = hip
    from smrender_amd import synth
    proj = sa.Projection.from_window(45.0, 15.0, 25000.0, 400.0, 300.0, 300)
    hs.set_chart(proj, 0xffffff)
    x, y, off, ref = synth.synth_ways(60, 400, 300, seed=17)
    # every third way reversed (clockwise / counterclockwise mix), every fifth opened (its closing ref dropped)
    lat, lon = synth.px_to_geo(proj, x, y)
    refs, offs, wids = [], [0], []
    base = 5_000_000
    hs.add_nodes(base + np.arange(len(x)), lat, lon)
    for w in range(len(off) - 1):
        r = ref[off[w]:off[w + 1]].copy()
        if w % 3 == 0:
            r = r[::-1].copy()
        if w % 5 == 4:
            r = r[:-1]
        refs.append(r); offs.append(offs[-1] + len(r))
        hs.add_way(base + len(x) + w, base + r.astype(np.int64))
        wids.append(base + len(x) + w)
    D = sa.DeviceArray.from_numpy
    n = len(wids)
    d_area, d_clat, d_clon = (sa.DeviceArray((n,), np.float64) for _ in range(3))
    sa.poly_area(D(lat), D(lon), D(np.array(offs, np.int64)), D(np.concatenate(refs).astype(np.int32)), n, d_area, d_clat, d_clon)
    sa.sync()
    area, clat, clon = d_area.numpy(), d_clat.numpy(), d_clon.numpy()
    seen_cw = seen_ccw = 0
    for w in range(n):
        if w % 5 == 4:
            assert area[w] == 0 and np.isnan(clat[w]) and np.isnan(clon[w])
            continue
        hlat, hlon, har = hs.poly_area(wids[w])
        assert np.sign(area[w]) == np.sign(har)
        # the reference's formula cancels about nine digits (terms of ~500 deg^2 sum to ~1e-6 deg^2 for a way of this size), so
        # a last-bit difference between the device's cos and glibc's shows at 1e-7 relative: the reference's own value is only
        # that well defined
        # (an ABSOLUTE error of ~1e-10 square miles whatever the way's size; the centroid divides by the area)
        assert abs(area[w] - har) <= 1e-9 and abs(clat[w] - hlat) <= 4e-9 / abs(har) and abs(clon[w] - hlon) <= 4e-9 / abs(har), w
        seen_cw += har < 0; seen_ccw += har > 0
    assert seen_cw > 5 and seen_ccw > 5


def test_project_keeps_the_references_ieee_results_where_it_can(hip):
    """x = (lon - lon0) * w / wc is a multiply and a division in the reference (src/rdata.c:222): the kernel's reciprocal form
    gives the same IEEE doubles.  y goes through the kernel's own evaluation of asinh(tan(lat)) (one division, no libm call):
    a few ulp of the ordinate at every latitude, the equator (where a difference of sums would cancel), latitudes beyond
    89.9 degrees (the C library's path, as the reference computes them) and the seam between the two included."""
    sa = hip
    p, op = _setup(sa, 10.0, -30.0, 3.0e7, 20000.0, 30000.0, 300)
    rng = np.random.default_rng(11)
    lat = np.concatenate([rng.uniform(-89.99, 89.99, 300001), [0.0, 1e-12, -1e-12, 1e-7, 89.9, -89.9, 89.899999, 89.95, -89.95, 45.0, -45.0],
                          rng.uniform(-1e-6, 1e-6, 1000)])
    lon = rng.uniform(-180, 180, lat.size)
    d_lat, d_lon = sa.DeviceArray.from_numpy(lat), sa.DeviceArray.from_numpy(lon)
    d_x, d_y = sa.DeviceArray(lat.shape, np.float64), sa.DeviceArray(lat.shape, np.float64)
    sa.project(d_lat, d_lon, d_x, d_y)
    x, y, _, _, _ = po.project(op, lat, lon)
    gx, gy = d_x.numpy(), d_y.numpy()
    assert np.array_equal(gx, x)
    # the ordinate m = asinh(tan(lat)): y = h * (0.5 - (m - lath) / lath_len), so an error of e in m is h * e / lath_len px
    m_err = np.abs(gy - y) * op.lath_len / op.h
    m = np.arcsinh(np.tan(np.radians(lat)))
    assert np.max(m_err / np.maximum(np.abs(m), 1.0)) < 4e-15          # towards the poles both evaluations lose digits
    mid = np.abs(lat) < 80
    assert np.max(m_err[mid]) < 1.5e-15
