"""Diagnostic (GPU box): one rank's share of BASELINE config 5 (100 M nodes / 10 M ways on 65 536^2, 8 GPUs): a band of
65 536 x 8 192 px with 1.25 M ways of ~10 refs, 30 % closed; windows against the oracle."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import smrender_amd as sa
from oracle import pyoracle as po
from golden_io import draw_style
sa.init(0); sa.set_profiling(True)
W, HT, BH, dpi = 65536, 65536, 8192, 300
rng = np.random.default_rng(5)
n_ways = 1250000
n = np.clip(rng.poisson(9, n_ways), 2, 64)
closed = rng.uniform(size=n_ways) < 0.3
n = np.where(closed, np.maximum(n, 4), n)
off = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
cx, cy = rng.uniform(0, W, n_ways), rng.uniform(0, BH, n_ways)
r = np.exp(rng.uniform(np.log(4), np.log(64), n_ways))
way = np.repeat(np.arange(n_ways), n)
k = np.arange(off[-1]) - off[way]
ang = 2 * np.pi * k / np.where(closed, n - 1, n)[way] + rng.uniform(0, 6.28, n_ways)[way]
rad = r[way] * rng.uniform(0.7, 1.0, off[-1])
x, y = cx[way] + rad * np.cos(ang), cy[way] + rad * np.sin(ang)
last = off[1:] - 1
x[last[closed]] = x[off[:-1][closed]]; y[last[closed]] = y[off[:-1][closed]]
ref = np.arange(off[-1], dtype=np.int32); ref[last[closed]] = off[:-1][closed]
st = po.Style(1, 1, 0x4080c0, 0x000000, 0.1, 25.4 / dpi, 0)
dx, dy, doff, dref = (sa.DeviceArray.from_numpy(a) for a in (x, y, off, ref))
cv = sa.Canvas(W, HT, dpi, 0xffffff, 0, BH)
for it in range(3):
    cv.clear(); sa.sync(); t0 = time.time()
    cv.draw(draw_style(sa, st), dx, dy, doff, dref, n_ways, len(ref)); sa.sync()
    print("draw %.1f ms" % ((time.time() - t0) * 1e3), {k: round(v, 2) for k, v in sa.timings().items() if k.endswith("_ms")})
print("status", sa.lib().smr_hip_draw_status(), "refs", len(ref))
ways = po.ways_from_arrays(x, y, off, ref)
S = 192
for wx, wy in ((0, 0), (65536 - S, 8192 - S), (31000, 4000)):
    g = cv.read(wx, wy, S, S)
    near = [w for w in ways if w[0].max() > wx - 80 and w[0].min() < wx + S + 80 and w[1].max() > wy - 80 and w[1].min() < wy + S + 80] if False else None
    sel = (cx > wx - 150) & (cx < wx + S + 150) & (cy > wy - 150) & (cy < wy + S + 150)
    sub = [ways[i] for i in np.nonzero(sel)[0]]
    o = po.oracle_scene([(st, sub)], dpi, S, S, 0xffffff, wx, wy, image=(0, 0, W, HT))
    print("window", wx, wy, "ways", len(sub), "identical", bool(np.array_equal(g, o)), int((g != o).sum()))
cv.close()
