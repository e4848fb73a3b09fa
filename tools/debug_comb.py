"""Diagnostic (GPU box): comb scenes against the oracle, to tell the second raster pass from the row-list pruning."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import smrender_amd as sa
from oracle import pyoracle as po
from test_gpu_raster import _comb, gpu_scene, oracle_of, STYLES, report
sa.init(0)
for teeth, pitch, width, x0 in ((10, 2.37, 1.1, 66.2), (26, 2.37, 1.1, 66.2), (26, 6.0, 1.1, 6.2), (14, 2.37, 1.1, 66.2), (16, 2.37, 1.1, 66.2), (26, 2.37, 1.1, 70.2)):
    for style in ("fill+border", "border", "hairline"):
        W, H = 260, 64
        comb = _comb(teeth, x0, pitch, width, 7.3, 41.8, W)
        rules = [(STYLES[style], comb)]
        want = oracle_of(rules, 300, W, H, 0xffffff)
        got = gpu_scene(sa, rules, 300, W, H, 0xffffff)
        d = po.chan_diff(got, want)
        print(teeth, pitch, x0, style, "status", sa.lib().smr_hip_draw_status(), "diff px", int((d > 0).sum()), "max", int(d.max()), np.argwhere(d > 0)[:6].tolist())
