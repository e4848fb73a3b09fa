"""Diagnostic (GPU box): the edge slots the library generates for one way of a dumped scene (-> gpurun_out/edges_<tag>.npy)
and the way's image alone."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import smrender_amd as sa
from oracle import pyoracle as po
from golden_io import draw_style
sa.init(0)
d = np.load(sys.argv[1]); w = int(sys.argv[2]); tag = sys.argv[3]
x, y, off, ref = d["x"], d["y"], d["off"], d["ref"]
W, H, dpi = int(d["W"]), int(d["H"]), int(d["dpi"])
st = po.Style.from_saved(d["style"])
r1 = np.ascontiguousarray(ref[off[w]:off[w + 1]]); o1 = np.array([0, len(r1)], np.int64)
cv = sa.Canvas(W, H, dpi, 0xffffff)
cv.draw(draw_style(sa, st), *(sa.DeviceArray.from_numpy(a) for a in (x, y, o1, r1)), 1, len(r1))
e = cv.debug_edges(); img = cv.read(); cv.close()
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
np.save(os.path.join(R, "gpurun_out", f"edges_{tag}.npy"), e)
np.save(os.path.join(R, "gpurun_out", f"img_{tag}.npy"), img)
print(tag, "edge slots", len(e), "non-empty", int((e != 0).any(1).sum()), "status", sa.lib().smr_hip_draw_status())
