"""Diagnostic (GPU box): one small rule drawn again and again; with rocprofv3 --kernel-trace the gaps between the kernels show the
host's share of the pass.  usage: small_rule_trace.py [size] [ways] [reps]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import smrender_amd as sa
from smrender_amd import synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
n_ways = int(sys.argv[2]) if len(sys.argv) > 2 else 10
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
sa.init(0)
cv = sa.Canvas(size, size, 300, 0xffffff)
x, y, off, ref = synth.synth_ways(n_ways, size, size, seed=5)
D = sa.DeviceArray.from_numpy
a = (sa.DrawStyle(1, 1, 0x4080c0, 0, 0.0, 0.1, 0, 0, 0, 0, 0, 0), D(x), D(y), D(off.astype(np.int64)), D(ref.astype(np.int32)), len(off) - 1, len(ref))
for _ in range(3):
    cv.draw(*a)
sa.sync()
t0 = time.perf_counter()
for _ in range(reps):
    cv.draw(*a)
sa.sync()
print(f"{size}^2, {n_ways} ways: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per draw")
