"""Diagnostic (GPU box): draws one way of a dumped scene once (for builds with device-side prints)."""
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
st = po.Style.from_saved(d["style"])
r1 = np.ascontiguousarray(ref[off[w]:off[w + 1]]); o1 = np.array([0, len(r1)], np.int64)
cv = sa.Canvas(int(d["W"]), int(d["H"]), int(d["dpi"]), 0xffffff)
cv.draw(draw_style(sa, st), *(sa.DeviceArray.from_numpy(a) for a in (x, y, o1, r1)), 1, len(r1))
img = cv.read(); cv.close()
print("pixel", hex(int(img[117, 128])))
