"""Live pinning of the oracle against the libcairo of this image (skipped where libcairo is
not loadable).  Larger samples than the committed fixtures."""
import numpy as np
import pytest
from oracle import pyoracle as po

pytestmark = pytest.mark.skipif(po.cairo() is None, reason="libcairo.so.2 not loadable")


def test_version():
    assert po.cairo().cd_version().decode().startswith("1.16")


def test_tor_exact_on_random_polygons():
    cd = po.cairo()
    rng = np.random.default_rng(1)
    edge_px = 0
    for trial in range(150):
        W = int(rng.integers(16, 80))
        n = 3 + trial % 9
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        r = rng.uniform(0.2, 0.48, n) * W
        xy = np.ascontiguousarray(np.stack([W / 2 + r * np.cos(ang), W / 2 + r * np.sin(ang)], 1))
        if trial % 3 == 0:
            rng.shuffle(xy)
        out = np.zeros((W, W), np.uint32)
        cd.cd_direct_fill(po.P(xy), n, 0x000000, 0xffffff, W, W, po.P(out))
        want = (255 - (out & 0xff)).astype(np.uint8)
        p = np.rint(xy * 256).astype(np.int32)
        got = po.tor_coverage(np.concatenate([p, np.roll(p, -1, 0)], 1), W, W)
        assert np.array_equal(got, want), trial
        edge_px += int(((want > 0) & (want < 255)).sum())
    assert edge_px > 10000


@pytest.mark.parametrize("colA,colB,bg", [(0x4080c0, 0xc08040, 0xffffff), (0x4080c0, 0x30c08040, 0xffffff),
                                          (0x204080c0, 0x50c08040, 0x102030), (0x204080c0, 0xc08040, 0x7f102030)])
def test_compositing_exact(colA, colB, bg):
    """two overlapping fills in one group: lerp (opaque / first on a clear layer) vs pixman OVER-with-mask"""
    t1 = (np.array([2.3, 29.1, 9.4, 2.3]), np.array([1.7, 6.2, 28.9, 1.7]), True)
    t2 = (np.array([28.3, 25.1, 3.4, 28.3]), np.array([2.7, 26.2, 12.9, 2.7]), True)
    rules = [(po.Style(1, 0, colA, 0, 0, 0, 0), [t1]), ]
    rules2 = [(po.Style(1, 0, colA, 0, 0, 0, 0), [t1]), (po.Style(1, 0, colB, 0, 0, 0, 0), [t2])]
    for rr in (rules, rules2):
        a = po.oracle_scene(rr, 72, 32, 32, bg)
        b = po.cairo_scene(rr, 72, 32, 32, bg)
        assert np.array_equal(a, b)


def test_c2_like_scene_statistics():
    arrs = po.synth_ways(900, 512, 512, seed=5)
    rules = [(po.Style(1, 1, 0x4080c0, 0x000000, 0, 25.4 / 300, 0), po.ways_from_arrays(*arrs))]
    a = po.oracle_scene(rules, 300, 512, 512, 0xffffff)
    b = po.cairo_scene(rules, 300, 512, 512, 0xffffff)
    d = po.chan_diff(a, b)
    assert (d <= 2).mean() >= 0.995 and (d == 0).mean() >= 0.98


def test_rot_paint_exact():
    cd = po.cairo()
    rng = np.random.default_rng(2)
    for t in range(40):
        dw, dh = int(rng.integers(8, 80)), int(rng.integers(4, 30))
        r = int(round(np.hypot(dw, dh)))
        src = (0xff000000 | rng.integers(0, 1 << 24, (r, r))).astype(np.uint32)
        d0 = (0xff000000 | rng.integers(0, 1 << 24, (dh, dw))).astype(np.uint32)
        n = int(rng.integers(20, 90))
        k = int(rng.integers(0, n))
        a1, a2 = d0.copy(), d0.copy()
        assert cd.cd_rot_paint(po.P(a1), dw, dh, po.P(src), r, r, k * 2 * np.pi / n) == 0
        po.orc().orc_rot_paint(po.P(a2), dw, dh, po.P(src), r, r, k * 2 * np.pi / n)
        assert np.array_equal(a1, a2), t
