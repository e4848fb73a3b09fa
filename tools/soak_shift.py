"""Diagnostic (GPU box): one way of a dumped scene on taller / shifted canvases against the oracle."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import smrender_amd as sa
from oracle import pyoracle as po
from golden_io import draw_style
sa.init(0)
d = np.load(sys.argv[1]); w = int(sys.argv[2])
x, y, off, ref = d["x"], d["y"], d["off"], d["ref"]
W, H, dpi = int(d["W"]), int(d["H"]), int(d["dpi"])
st = po.Style.from_saved(d["style"])
r1 = np.ascontiguousarray(ref[off[w]:off[w + 1]]); o1 = np.array([0, len(r1)], np.int64)
for tag, dy, HH in (("as is", 0, H), ("taller", 0, H + 30), ("up 10", -10, H), ("up 1", -1, H), ("down 3 taller", 3, H + 30), ("H+1", 0, H + 1), ("H+2", 0, H + 2)):
    yy = y + dy
    wt = po.oracle_scene([(st, po.ways_from_arrays(x, yy, o1, r1))], dpi, W, HH, 0xffffff)
    c1 = sa.Canvas(W, HH, dpi, 0xffffff)
    c1.draw(draw_style(sa, st), *(sa.DeviceArray.from_numpy(a) for a in (x, yy, o1, r1)), 1, len(r1))
    g1 = c1.read(); c1.close()
    dd = np.argwhere(g1 != wt)
    print(tag, "diff", len(dd), [(int(r), int(c), hex(int(g1[r, c])), hex(int(wt[r, c]))) for r, c in dd[:6]], flush=True)
