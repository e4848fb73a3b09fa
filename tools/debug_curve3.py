"""Diagnostic (GPU box): edge slots of one curve way, GPU vs oracle."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import smrender_amd as sa
from smrender_amd import synth
from oracle import pyoracle as po
from golden_io import draw_style
sa.init(0)
S = po.Style
L = po.orc()
x, y, off, ref = po.synth_ways(3000, 1024, 1024, seed=8)
wi = 2561
r = ref[off[wi]:off[wi + 1]]
wx, wy = x[r], y[r]
dpi = 600
st = S(0, 1, 0, 0, 0, 2 * 25.4 / 300, 0).with_curve()
cv = sa.Canvas(1024, 1024, dpi, 0xffffff)
o1 = np.array([0, len(r)], np.int64); r1 = np.arange(len(r), dtype=np.int32); r1[-1] = 0
cv.draw(draw_style(sa, st), sa.DeviceArray.from_numpy(wx[:-1].copy()), sa.DeviceArray.from_numpy(wy[:-1].copy()),
        sa.DeviceArray.from_numpy(o1), sa.DeviceArray.from_numpy(r1), 1, len(r))
ge = cv.debug_edges()
ge = ge[(ge != 0).any(1)]
pt = np.ascontiguousarray(np.stack([wx * 72 / dpi, wy * 72 / dpi], 1))
start, curves = po.curve_path(pt, True, 0.2)
cap = 64 * (len(curves) + 2)
pts = np.zeros(2 * cap, np.int32); tin = np.zeros(2 * cap, np.int32); tout = np.zeros(2 * cap, np.int32)
m = L.orc_curve_flatten(po.P(start), po.P(curves), len(curves), dpi, None, po.P(pts), po.P(tin), po.P(tout), cap)
oe = np.zeros((8 * m, 4), np.int32)
hw_pt = po.mm2wu(2 * 25.4 / 300, dpi) / 2
ne = L.orc_stroke_edges_tan(po.P(pts), po.P(tin), po.P(tout), m, hw_pt, dpi, po.P(oe))
oe = oe[:ne]
print("gpu edges", len(ge), "oracle edges", len(oe))
n = min(len(ge), len(oe))
bad = np.nonzero((ge[:n] != oe[:n]).any(1))[0]
print("first differing", bad[:10].tolist())
for k in bad[:12]:
    print(k, "gpu", ge[k].tolist(), "oracle", oe[k].tolist())
if len(ge) != len(oe):
    print("tail gpu", ge[n - 2:n + 6].tolist()); print("tail oracle", oe[n - 2:n + 6].tolist())
