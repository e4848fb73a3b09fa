"""Diagnostic (GPU box): one style's windows against libcairo's cut-outs, the oracle's shifted render and the chart's crop."""
import os, sys, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import smrender_amd as sa
from oracle import pyoracle as po
from golden_io import draw_style, within
import test_gpu_windows as T
name = sys.argv[1] if len(sys.argv) > 1 else "dashed"
whole = len(sys.argv) > 2
sa.init(0)
L, cd = po.orc(), po.cairo()
L.orc_canvas_set_shift.argtypes = [C.c_void_p, C.c_double, C.c_double]
DPI = 300
x, y, off, ref = po.synth_ways(1500, 900, 700, seed=21)
ways = po.ways_from_arrays(x, y, off, ref)
sty = T._style(name)
rng = np.random.default_rng(3)
n, r = 7, (92 if whole else 93)
cx, cy = rng.uniform(100, 800, n), rng.uniform(100, 600, n)
if whole:
    cx, cy = np.floor(cx), np.floor(cy)
D = sa.DeviceArray.from_numpy
args = (draw_style(sa, sty), D(x), D(y), D(off.astype(np.int64)), D(ref.astype(np.int32)), len(off) - 1, len(ref))
at = sa.Canvas.windows(900, 700, DPI, 0xfff8e0, cx, cy, r)
at.draw_windows(*args)
img = at.read()
cv = sa.Canvas(900, 700, DPI, 0xfff8e0); cv.draw(*args); chart = cv.read()
po.cairo_scene([(sty, ways)], DPI, 900, 700, 0xfff8e0)
r_pt = r * 72 / DPI
sub = np.zeros((r + 2) * (r + 2), np.uint32)
for k in range(n):
    cd.cd_cut_out(cx[k] * 72 / DPI, cy[k] * 72 / DPI, r_pt, float(DPI), po.P(sub), len(sub))
    ref_cut = sub[:r * r].reshape(r, r).copy()
    got = img[k * at.window_pitch:k * at.window_pitch + r]
    c = L.orc_canvas_new(0, 0, r, r, DPI, 0xfff8e0)
    tx, ty = -cx[k] * 72 / DPI + r_pt / 2, -cy[k] * 72 / DPI + r_pt / 2
    L.orc_canvas_set_shift(c, tx, ty)
    L.orc_rule_begin(c)
    for wx, wy, closed in ways:
        wx = np.ascontiguousarray(wx, np.float64); wy = np.ascontiguousarray(wy, np.float64)
        if wx.max() < cx[k] - 200 or wx.min() > cx[k] + 200 or wy.max() < cy[k] - 200 or wy.min() > cy[k] + 200: continue
        L.orc_draw_way(c, C.byref(sty), po.P(wx), po.P(wy), len(wx), int(closed), 0, 0, 0, 0)
    L.orc_rule_end(c)
    orc = np.ctypeslib.as_array(C.cast(L.orc_canvas_pixels(c), C.POINTER(C.c_uint32)), (r, r)).copy()
    L.orc_canvas_free(c)
    line = f"{k}: gpu~cairo {within(got, ref_cut):.4f} oracle~cairo {within(orc, ref_cut):.4f} gpu==oracle {(got == orc).mean():.4f}"
    if whole:
        crop = chart[int(cy[k]) - r // 2:int(cy[k]) + r // 2, int(cx[k]) - r // 2:int(cx[k]) + r // 2]
        line += f" gpu==chart crop {(got == crop).mean():.4f} cairo~crop {within(ref_cut, crop):.4f}"
    print(line)
