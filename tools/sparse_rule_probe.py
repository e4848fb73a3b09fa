"""Diagnostic (GPU box): the fixed cost of a draw pass -- a rule of few ways on a large canvas."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import smrender_amd as sa
from smrender_amd import synth
sa.init(0)
sa.set_profiling(True)
for size in (4096, 16384, 32768):
    cv = sa.Canvas(size, size, 300, 0xffffff)
    for n_ways in (10, 1000, 100000):
        x, y, off, ref = synth.synth_ways(n_ways, size, size, seed=5)
        D = sa.DeviceArray.from_numpy
        a = (sa.DrawStyle(1, 1, 0x4080c0, 0, 0.0, 0.1, 0, 0, 0, 0, 0, 0), D(x), D(y), D(off.astype(np.int64)), D(ref.astype(np.int32)), len(off) - 1, len(ref))
        cv.draw(*a); sa.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            cv.draw(*a)
        sa.sync()
        dt = (time.perf_counter() - t0) / 5
        t = sa.timings()
        print(f"{size}^2, {n_ways} ways: {dt * 1e3:.2f} ms per draw; stages", {k: round(v, 3) for k, v in t.items() if k.endswith('_ms')})
    cv.close()
