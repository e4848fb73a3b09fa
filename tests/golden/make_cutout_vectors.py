#!/usr/bin/env python3
"""Generates tests/golden/cutout_subpixel.npz (run here, where libcairo 1.16 is): how the caption scan's cut-out convention
compares with the reference's.

The reference cuts the background out by REPLAYING the recording at a sub-pixel offset (cairo_smr_cut_out,
src/smrules_cairo.c:1511-1533: translate by -x + r/2 pt, whatever fraction of a pixel that is); libsmr_hip crops the
rendered canvas at the rounded centre.  For captions with non-integer anchors on a C2-like chart this
script runs find_angle (oracle/caprot.c, pinned against libcairo's rotated paint) on
  ref  : libcairo's own sub-pixel cut-out of the recording (oracle/cairo_drive.c:cd_cut_out),
  conv : the crop convention on libcairo's full render (isolates the convention),
  gpu  : the crop convention on the oracle's render (= the GPU's pixels bit for bit: the end-to-end difference)
and stores anchors and the three angle vectors."""
import ctypes as C
import os
import sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from oracle import pyoracle as po

W = H = 2048
DPI = 300
N_CAP = 2000
L, cd = po.orc(), po.cairo()
x, y, off, ref = po.synth_ways(15625, W, H, seed=1234)           # C2's density and style
st = po.Style(1, 1, 0x4080c0, 0x000000, 0.0, 25.4 / DPI, 0)
ways = po.ways_from_arrays(x, y, off, ref)
img_orc = po.oracle_scene([(st, ways)], DPI, W, H, 0xffffff)
img_cairo = po.cairo_scene([(st, ways)], DPI, W, H, 0xffffff)       # leaves the recording in the driver for cd_cut_out
crnd = lambda v: int(np.sign(v) * np.floor(abs(v) + 0.5))
out = {"W": W, "H": H, "dpi": DPI, "seed": 1234, "n_ways": 15625}
for tag, (fw, fh, fx) in {"c4": (270, 20, 135), "small": (90, 14, 45)}.items():
    r = C.c_int(); steps = L.orc_num_steps(fw, fh, DPI, C.byref(r)); r = r.value
    r_pt = np.hypot(fw, fh) * 72 / DPI                             # find_angle :1975
    fg = np.zeros((fh, fw), np.uint32); L.orc_make_plane(po.P(fg), fw, fh, fx, 0x000000)
    rng = np.random.default_rng(fw)
    cx, cy = rng.uniform(r, W - r, N_CAP), rng.uniform(r, H - r, N_CAP)
    a = np.zeros((3, N_CAP))
    sub = np.zeros((r + 2) * (r + 2), np.uint32)
    for k in range(N_CAP):
        n = cd.cd_cut_out(cx[k] * 72 / DPI, cy[k] * 72 / DPI, r_pt, float(DPI), po.P(sub), len(sub))
        assert n == r, (n, r)
        cut = np.ascontiguousarray(sub[:r * r].reshape(r, r))
        a[0, k] = L.orc_find_angle(po.P(cut), r, po.P(fg), fw, fh, steps, 0.0, 1.0, None)
        ox, oy = crnd(cx[k] - r / 2.0), crnd(cy[k] - r / 2.0)
        for j, img in ((1, img_cairo), (2, img_orc)):
            cut = np.ascontiguousarray(img[oy:oy + r, ox:ox + r])
            a[j, k] = L.orc_find_angle(po.P(cut), r, po.P(fg), fw, fh, steps, 0.0, 1.0, None)
    t = lambda v: np.array(["%.1f" % (q * 180 / np.pi) for q in v])
    print(tag, "plane", (fw, fh, fx), "steps", steps, "r", r, "| tags differing from the reference's cut-out: convention alone",
          int((t(a[0]) != t(a[1])).sum()), "of", N_CAP, "; end to end (GPU pixels)", int((t(a[0]) != t(a[2])).sum()),
          "; convention vs GPU pixels", int((t(a[1]) != t(a[2])).sum()))
    d = np.abs(a[0] - a[2]); d = np.minimum(d, 2 * np.pi - d) * 180 / np.pi
    print("   end-to-end angle difference where tags differ: median %.1f deg, max %.1f deg" % (np.median(d[d > 0.05]) if (d > 0.05).any() else 0, d.max()))
    out.update({f"{tag}_plane": np.array([fw, fh, fx]), f"{tag}_cx": cx, f"{tag}_cy": cy, f"{tag}_ref": a[0], f"{tag}_conv": a[1], f"{tag}_gpu": a[2]})
np.savez_compressed(os.path.join(R, "tests", "golden", "cutout_subpixel.npz"), **out)
