"""The plugin boundary on a GPU-less host: the reference's own libsmrender (oracle/_ref) is the
host, our layouts are pinned against its headers at build time (oracle/hostsim.c), and the
action entry points fail loudly -- never fall back -- when no GPU is there."""
import pytest
import smrender_amd as sa
from oracle import pyoracle as po
import hostsim as hs

pytestmark = pytest.mark.skipif(not hs.available(), reason="oracle/_ref not built (needs /root/reference at build time)")


def test_layouts_pinned_against_reference_headers():
    assert hs.host().hs_abi_check() == 0


def test_draw_on_node_rule_is_rejected_like_the_reference():
    # act_draw_ini: "'draw' may be applied to ways or relations only" -> 1 (src/smrules_cairo.c:590-594)
    hs.set_chart(po.make_proj(45.0, 15.0, 20000.0, 256.0, 256.0, 300))
    assert hs.apply_rule(sa.lib(), "draw", hs.OSM_NODE, "color=red", []) == 2001


def test_no_gpu_no_render():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    hs.set_chart(po.make_proj(45.0, 15.0, 20000.0, 256.0, 256.0, 300))
    assert hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "color=red", []) == 1001      # _ini < 0: error, nothing drawn
    assert hs.apply_rule(sa.lib(), "cap", hs.OSM_NODE, "key=name;angle=auto", []) == 1001
    assert b"no HIP device" in sa.lib().smr_hip_last_error()


def test_host_poly_area_against_an_independent_shoelace():
    """cap on closed ways (cap_way, src/smrules_cairo.c:2395-2444) takes centroid and area from the executable's poly_area
    (src/smfunc.c:377-432), which hostsim restates: pinned by the shoelace identity in numpy (area of a known square in
    nm^2, orientation sign, centroid of a symmetric polygon)"""
    import numpy as np
    hs.set_chart(po.make_proj(45.0, 15.0, 20000.0, 256.0, 256.0, 300))
    rng = np.random.default_rng(4)
    for k, sgn in enumerate((1, -1)):
        n = 9
        a = np.sort(rng.uniform(0, 2 * np.pi, n)) * sgn
        lat = 45.0 + 0.01 * rng.uniform(0.5, 1.0, n) * np.sin(a)
        lon = 15.0 + 0.01 * rng.uniform(0.5, 1.0, n) * np.cos(a)
        ids = 900000 + 100 * k + np.arange(n)
        hs.add_nodes(ids, lat, lon)
        hs.add_way(900090 + 100 * k, np.append(ids, ids[0]))
        clat, clon, ar = hs.poly_area(900090 + 100 * k)
        la, lo = np.append(lat, lat[0]), np.append(lon, lon[0])
        x = lo * np.cos(np.radians(la))
        f = x[:-1] * la[1:] - x[1:] * la[:-1]
        assert abs(ar - f.sum() * 1800.0) <= 1e-8 * abs(ar) and np.sign(ar) == sgn      # counterclockwise positive
        wlat = ((la[:-1] + la[1:]) * f).sum() / (3 * f.sum())
        # (sums of terms of ~1e3 that cancel to ~1e-4: the order of summation shows at 1e-11)
        assert abs(clat - wlat) < 1e-9 and abs(clon - ((x[:-1] + x[1:]) * f).sum() / (3 * f.sum() * np.cos(np.radians(wlat)))) < 1e-9
        assert lat.min() < clat < lat.max() and lon.min() < clon < lon.max()
    # a square of 1' x 1' at the equator is one square nautical mile
    ids = 900500 + np.arange(4)
    hs.add_nodes(ids, np.array([0, 0, 1 / 60, 1 / 60]), np.array([0, 1 / 60, 1 / 60, 0]))
    hs.add_way(900590, np.append(ids, ids[0]))
    assert abs(hs.poly_area(900590)[2] - 1.0) < 1e-6
    H = hs.host()
    assert abs(H.rdata_square_mm() - (256 * 25.4 / 300) ** 2) < 1e-9 and H.rdata_square_nm() > 0
