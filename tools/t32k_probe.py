"""Diagnostic (GPU box): stage times of the 32k target (1 M ways / 10 M nodes on 32768^2)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import smrender_amd as sa
from smrender_amd import synth
sa.init(0); sa.set_profiling(True)
S = 32768
x, y, off, ref = synth.synth_ways(1000000, S, S, seed=4242, mean_nodes=10)
d = [sa.DeviceArray.from_numpy(a) for a in (x, y, off.astype(np.int64), ref.astype(np.int32))]
cv = sa.Canvas(S, S, 300, 0xffffff)
st = sa.DrawStyle(1, 1, 0x4080c0, 0x000000, 0.0, 25.4 / 300, 0, 0, 0, 0, 0, 0)
for it in range(3):
    cv.clear(); cv.draw(st, d[0], d[1], d[2], d[3], len(off) - 1, len(ref)); sa.sync()
    t = sa.timings()
    print({k: round(v, 2) if isinstance(v, float) else v for k, v in t.items() if k.endswith("_ms") or k.startswith("n_")})
