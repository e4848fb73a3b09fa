"""Diagnostic (GPU box): windows of the golden cut-out scene pixel by pixel against libcairo's own cut-outs."""
import os, sys, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import smrender_amd as sa
from oracle import pyoracle as po
G = np.load(os.path.join(R, "tests", "golden", "cutout_subpixel.npz"))
W, H, DPI = int(G["W"]), int(G["H"]), int(G["dpi"])
sa.init(0)
L, cd = po.orc(), po.cairo()
x, y, off, ref = po.synth_ways(int(G["n_ways"]), W, H, seed=int(G["seed"]))
sty = po.Style(1, 1, 0x4080c0, 0x000000, 0.0, 25.4 / DPI, 0)
po.cairo_scene([(sty, po.ways_from_arrays(x, y, off, ref))], DPI, W, H, 0xffffff)
D = sa.DeviceArray.from_numpy
st = sa.DrawStyle(1, 1, 0x4080c0, 0x000000, 0.0, 25.4 / DPI, 0, 0, 0, 0, 0, 0)
tag = sys.argv[1] if len(sys.argv) > 1 else "c4"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 120
fw, fh, fx = (int(v) for v in G[tag + "_plane"])
r = C.c_int(); steps = L.orc_num_steps(fw, fh, DPI, C.byref(r)); r = r.value
r_pt = np.hypot(fw, fh) * 72 / DPI
cx, cy = G[tag + "_cx"][:N], G[tag + "_cy"][:N]
at = sa.Canvas.windows(W, H, DPI, 0xffffff, cx, cy, r, float(np.hypot(fw, fh)))
at.draw_windows(st, D(x), D(y), D(off.astype(np.int64)), D(ref.astype(np.int32)), len(off) - 1, len(ref))
img = at.read()
ax, ay = at.window_anchors()
d = sa.DeviceArray((N,), np.float64)
at.caprot(D(ax), D(ay), N, fw, fh, fx, 0, 0.0, 1.0, d)
sa.sync()
ang = d.numpy()
fg = np.zeros((fh, fw), np.uint32); L.orc_make_plane(po.P(fg), fw, fh, fx, 0)
sub = np.zeros((r + 2) * (r + 2), np.uint32)
print("r", r, "r_pt*s", r_pt * DPI / 72, "size_pt used by the library", r * 72 / DPI)
for k in range(N):
    n = cd.cd_cut_out(cx[k] * 72 / DPI, cy[k] * 72 / DPI, r_pt, float(DPI), po.P(sub), len(sub))
    want = sub[:r * r].reshape(r, r)
    got = img[k * at.window_pitch:k * at.window_pitch + r]
    nd = int((got != want).sum())
    a_cut = L.orc_find_angle(po.P(np.ascontiguousarray(got)), r, po.P(fg), fw, fh, steps, 0.0, 1.0, None)
    flag = "" if "%.1f" % np.degrees(ang[k]) == "%.1f" % np.degrees(G[tag + "_ref"][k]) else "  <-- tag differs"
    if nd > 200 or flag:
        ys, xs = np.nonzero(got != want)
        print(k, "px diff", nd, "rows", (ys.min(), ys.max()) if nd else None, "cols", (xs.min(), xs.max()) if nd else None,
              "gpu", round(np.degrees(ang[k]), 2), "oracle on gpu px", round(np.degrees(a_cut), 2), "ref", round(np.degrees(G[tag + "_ref"][k]), 2), flag)
