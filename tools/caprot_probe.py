"""Diagnostic (GPU box): the C4-shaped caption scan alone (20 000 captions, 270x20 plane, 72 angles) over a C2-like canvas of
4096^2 px, for rocprofv3 --pmc passes.  usage: caprot_probe.py [n_captions]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import smrender_amd as sa
from smrender_amd import synth
n_cap = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
sa.init(0); sa.set_profiling(True)
W = H = 4096
x, y, off, ref = synth.synth_ways(62500, W, H, seed=1234)
cv = sa.Canvas(W, H, 300, 0xffffff)
cv.draw(sa.DrawStyle(1, 1, 0x4080c0, 0x000000, 0.0, 25.4 / 300, 0, 0, 0, 0, 0, 0), *(sa.DeviceArray.from_numpy(a) for a in (x, y, off, ref)), len(off) - 1, len(ref))
rng = np.random.default_rng(4321)
side = int(np.ceil(np.sqrt(n_cap)))
gx, gy = np.meshgrid(np.arange(side), np.arange(side))
cx = (gx.ravel()[:n_cap] + rng.uniform(0.2, 0.8, n_cap)) * W / side
cy = (gy.ravel()[:n_cap] + rng.uniform(0.2, 0.8, n_cap)) * H / side
d_cx, d_cy, d_ang = sa.DeviceArray.from_numpy(cx), sa.DeviceArray.from_numpy(cy), sa.DeviceArray((n_cap,), np.float64)
for it in range(2):
    cv.caprot(d_cx, d_cy, n_cap, 270, 20, 135, 0x000000, 0.0, 1.0, d_ang); sa.sync()
print("caprot_ms", sa.timings()["caprot_ms"], "distinct", len(np.unique(d_ang.numpy())))
