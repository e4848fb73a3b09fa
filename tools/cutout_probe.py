#!/usr/bin/env python3
"""Diagnostic (CPU, where libcairo 1.16 is): one caption's cut-out of the golden set (tests/golden/cutout_subpixel.npz) through the
oracle's shifted replay and through libcairo's own (cairo_drive.c: cd_cut_out), pixel by pixel, and then every way of it ALONE
through both, as a fill and as a border: which way, which rows.  This is how the limits of the replay were found.
usage: tools/cutout_probe.py [c4|small] [caption index]      (ORC_STROKE_OUTLINE=1: strokes as libcairo's polygons)"""
import ctypes as C
import os
import sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from oracle import pyoracle as po

_z = np.load(os.path.join(R, "tests", "golden", "cutout_subpixel.npz"))
G = {k: _z[k] for k in _z.files}
W = H = int(G["W"]); DPI = int(G["dpi"])
tag = sys.argv[1] if len(sys.argv) > 1 else "c4"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1692
L, cd = po.orc(), po.cairo()
L.orc_canvas_set_shift.argtypes = [C.c_void_p, C.c_double, C.c_double]
x, y, off, ref = po.synth_ways(int(G["n_ways"]), W, H, seed=int(G["seed"]))
ways = po.ways_from_arrays(x, y, off, ref)
box = np.array([[w[0].min(), w[0].max(), w[1].min(), w[1].max()] for w in ways])
fw, fh, fx = (int(v) for v in G[tag + "_plane"])
r = C.c_int(); L.orc_num_steps(fw, fh, DPI, C.byref(r)); r = r.value
r_pt = np.hypot(fw, fh) * 72 / DPI
cx, cy = float(G[tag + "_cx"][k]), float(G[tag + "_cy"][k])
sub = np.zeros((r + 2) * (r + 2), np.uint32)
FULL = po.Style(1, 1, 0x4080c0, 0x000000, 0.0, 25.4 / DPI, 0)
FILL = po.Style(1, 0, 0x4080c0, 0, 0.0, 0, 0)
BORDER = po.Style(0, 1, 0, 0x000000, 0.0, 2 * 25.4 / DPI, 0)      # (a closed way's border under a fill is twice the border width)


def both(idx, style):
    """(libcairo's cut-out, the oracle's) of the ways idx drawn with style"""
    po.cairo_scene([(style, [ways[i] for i in idx])], DPI, W, H, 0xffffff)
    assert cd.cd_cut_out(cx * 72 / DPI, cy * 72 / DPI, r_pt, float(DPI), po.P(sub), len(sub)) == r
    a = sub[:r * r].reshape(r, r).copy()
    c = L.orc_canvas_new(0, 0, r, r, DPI, 0xffffff)
    L.orc_canvas_set_shift(c, -cx * 72 / DPI + r_pt / 2, -cy * 72 / DPI + r_pt / 2)
    L.orc_rule_begin(c)
    for i in idx:
        wx, wy, closed = ways[i]
        L.orc_draw_way(c, C.byref(style), po.P(np.ascontiguousarray(wx)), po.P(np.ascontiguousarray(wy)), len(wx), int(closed), 0, 0, 0, 0)
    L.orc_rule_end(c)
    b = np.ctypeslib.as_array(C.cast(L.orc_canvas_pixels(c), C.POINTER(C.c_uint32)), (r, r)).copy()
    L.orc_canvas_free(c)
    return a, b


m = r / 2 + 12
sel = [int(i) for i in np.nonzero(~((box[:, 1] < cx - m) | (box[:, 0] > cx + m) | (box[:, 3] < cy - m) | (box[:, 2] > cy + m)))[0]]
a, b = both(sel, FULL)
d = np.argwhere(a != b)
print(f"{tag} caption {k}: {r}-px cut-out, {len(sel)} ways, {len(d)} pixels differ, rows {sorted(set(d[:, 0].tolist()))}")
for nm, style in (("fill", FILL), ("border", BORDER)):
    for i in sel:
        a1, b1 = both([i], style)
        dd = np.argwhere(a1 != b1)
        if len(dd):
            wx, wy, _ = ways[i]
            print(f"  way {i} ({len(wx)} refs) alone, {nm}: {len(dd)} px, rows {sorted(set(dd[:, 0].tolist()))}, "
                  f"extent in the cut-out x {wx.min() - (cx - r / 2):.1f}..{wx.max() - (cx - r / 2):.1f} y {wy.min() - (cy - r / 2):.1f}..{wy.max() - (cy - r / 2):.1f}")
