"""Diagnostic (GPU box): C2 stage times for fill only / border only / both."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import smrender_amd as sa
from smrender_amd import synth
sa.init(0); sa.set_profiling(True)
W = H = 16384
x, y, off, ref = synth.synth_ways(1000000, W, H, seed=1234)
dx, dy, doff, dref = (sa.DeviceArray.from_numpy(a) for a in (x, y, off, ref))
cv = sa.Canvas(W, H, 300, 0xffffff)
for label, fu, bu in (("both", 1, 1), ("fill", 1, 0), ("border", 0, 1), ("closed-border-2bw", 0, 1)):
    bw = 25.4 / 300 * (2 if label.startswith("closed") else 1)
    gs = sa.DrawStyle(fu, bu, 0x4080c0, 0x000000, 0.0, bw, 0, 0, 0, 0, 0, 0)
    for it in range(2):
        cv.clear(); cv.draw(gs, dx, dy, doff, dref, len(off) - 1, len(ref)); sa.sync()
    t = sa.timings()
    print(label, {k: round(v, 2) if isinstance(v, float) else v for k, v in t.items() if k.endswith("_ms") or k in ("n_tile_refs",)})
