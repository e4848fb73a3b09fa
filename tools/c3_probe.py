"""Diagnostic (GPU box): C3-like chart (fractal coastline + islands as one directional curve fill, depth contours as curve
borders): small size against the oracle, full size for timing and capacity."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import smrender_amd as sa
from smrender_amd import synth
from oracle import pyoracle as po
from golden_io import draw_style
sa.init(0); sa.set_profiling(True)
S = po.Style

def draw(cv, st, arrs, **kw):
    x, y, off, ref = arrs
    cv.draw(draw_style(sa, st, **kw), sa.DeviceArray.from_numpy(x), sa.DeviceArray.from_numpy(y), sa.DeviceArray.from_numpy(off.astype(np.int64)),
            sa.DeviceArray.from_numpy(ref.astype(np.int32)), len(off) - 1, len(ref))

def small():
    W, H, dpi = 1024, 1448, 600
    land, rings = synth.synth_coast(20000, 300, 6, W, H, seed=3)
    st_land = S(1, 0, 0xf0e0a0, 0, 0, 0, 0).with_curve()
    st_ring = S(0, 1, 0, 0x4060a0, 0, 0.15, 0).with_curve()
    cv = sa.Canvas(W, H, dpi, 0xc0d8f0)
    draw(cv, st_land, land, directional=1)
    g1 = cv.read()
    draw(cv, st_ring, rings)
    g2 = cv.read(); cv.close()
    o1 = po.oracle_directional(st_land, po.ways_from_arrays(*land), dpi, W, H, 0xc0d8f0)
    d = po.chan_diff(g1, o1)
    print("small land: identical", (d == 0).mean(), "ndiff", int((d > 0).sum()), "max", int(d.max()), np.argwhere(d > 0)[:4].tolist())
    o2 = po.oracle_scene([(st_ring, po.ways_from_arrays(*rings))], dpi, W, H, 0xffffff)
    cvb = sa.Canvas(W, H, dpi, 0xffffff); draw(cvb, st_ring, rings); gb = cvb.read(); cvb.close()
    d = po.chan_diff(gb, o2)
    print("small rings: identical", (d == 0).mean(), "ndiff", int((d > 0).sum()), "max", int(d.max()), np.argwhere(d > 0)[:4].tolist())

def full():
    W, H, dpi = 19866, 28087, 600
    t0 = time.time()
    land, rings = synth.synth_coast(200000, 5000, 50, W, H, seed=1234)
    print("generated in %.1f s: land refs %d, ring refs %d" % (time.time() - t0, len(land[3]), len(rings[3])))
    st_land = S(1, 0, 0xf0e0a0, 0, 0, 0, 0).with_curve()
    st_ring = S(0, 1, 0, 0x4060a0, 0, 0.15, 0).with_curve()
    cv = sa.Canvas(W, H, dpi, 0xc0d8f0)
    for it in range(2):
        cv.clear(); sa.sync(); t0 = time.time()
        draw(cv, st_land, land, directional=1); sa.sync(); t1 = time.time()
        tl = sa.timings()
        draw(cv, st_ring, rings); sa.sync(); t2 = time.time()
        tr = sa.timings()
        print("full: land %.1f ms, rings %.1f ms, %.0f Mpx/s" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, W * H / (t2 - t0) / 1e6))
        print("   land stages", {k: round(v, 1) for k, v in tl.items() if k.endswith("_ms")}, "tile refs", tl["n_tile_refs"])
        print("   ring stages", {k: round(v, 1) for k, v in tr.items() if k.endswith("_ms")}, "tile refs", tr["n_tile_refs"])
    print("status", sa.lib().smr_hip_draw_status())
    win = cv.read(9000, 13000, 256, 256)
    print("window distinct colours", len(np.unique(win)))
    cv.close()

small()
full()
