"""Diagnostic (GPU box): where the time of the caption scan through re-rasterised cut-outs goes (C2 chart, C4 captions).
usage: python tools/replay_probe.py [n_atlases] -- wall time per phase, synchronised after each"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import smrender_amd as sa
from smrender_amd import synth
n_at = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sa.init(0)
W = H = 16384
x, y, off, ref = synth.synth_ways(1000000, W, H, seed=1234)
D = sa.DeviceArray.from_numpy
d_x, d_y, d_off, d_ref = D(x), D(y), D(off.astype(np.int64)), D(ref.astype(np.int32))
style = sa.DrawStyle(1, 1, 0x4080c0, 0x000000, 0.0, 25.4 / 300, 0, 0, 0, 0, 0, 0)
fg_w, fg_h, fg_x = 270, 20, 135
n_steps, r_px = sa.caprot_steps(fg_w, fg_h, 300)
per = 65536 // ((r_px + 3) & ~3)
n = per * n_at
rng = np.random.default_rng(4321)
side = int(np.ceil(np.sqrt(20000)))
gx, gy = np.meshgrid(np.arange(side), np.arange(side))
cx = ((gx.ravel()[:n] + rng.uniform(0.2, 0.8, n)) * W / side)
cy = ((gy.ravel()[:n] + rng.uniform(0.2, 0.8, n)) * H / side)
ph = {"create": 0.0, "draw": 0.0, "scan": 0.0, "close": 0.0}
def tick(name, t0):
    sa.sync(); t = time.perf_counter(); ph[name] += t - t0; return t
sa.set_profiling(True)
stages = {}
at = None
for rep in range(2):
    for k in ph: ph[k] = 0.0
    for k0 in range(0, n, per):
        k1 = k0 + per
        sa.sync(); t = time.perf_counter()
        if at is None: at = sa.Canvas.windows(W, H, 300, 0xffffff, cx[k0:k1], cy[k0:k1], r_px, float(np.hypot(fg_w, fg_h)))
        else: at.reset_windows(cx[k0:k1], cy[k0:k1], r_px, float(np.hypot(fg_w, fg_h)))
        t = tick("create", t)
        at.draw_windows(style, d_x, d_y, d_off, d_ref, len(off) - 1, len(ref))
        t = tick("draw", t)
        tm = sa.timings()
        for k in ("edges_ms", "prims_ms", "cover_ms", "bin_ms", "raster_ms", "total_ms"): stages[k] = stages.get(k, 0.0) + tm[k]
        ax, ay = at.window_anchors()
        d2 = sa.DeviceArray((per,), np.float64)
        at.caprot(D(ax), D(ay), per, fg_w, fg_h, fg_x, 0x000000, 0.0, 1.0, d2)
        t = tick("scan", t)
        a = d2.numpy()
        t = tick("close", t)
    print("rep", rep, {k: round(v * 1e3 / n_at, 2) for k, v in ph.items()}, "ms per atlas of", per, "windows;", round(n / sum(ph.values())), "captions/s")
print("draw pass stages per atlas (last rep x2):", {k: round(v / (2 * n_at), 2) for k, v in stages.items()}, "prims", tm["n_prims"], "tile_refs", tm["n_tile_refs"])
