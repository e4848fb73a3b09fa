#!/usr/bin/env python3
"""where the GPU's polygon-mode rendering of explicit Bezier ways differs from the oracle's (diagnostic)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa
import smrender_amd as sa
from smrender_amd import synth
from oracle import pyoracle as po
from test_gpu_raster import WAVY_STYLES, draw_style

sa.init(0)
W, H, dpi = 320, 256, 600
st = WAVY_STYLES["bordered"].with_knots()
ways = synth.synth_wavy_knots(24, W, H, seed=5, wave_px=(12.0, 60.0), closed=False)
sel = [int(a) for a in sys.argv[1:]] or list(range(len(ways)))
for only in [None] + sel:
    ww = ways if only is None else [ways[only]]
    want = po.oracle_scene([(st, ww)], dpi, W, H, 0xffffff)
    x = np.concatenate([w[0] for w in ww]); y = np.concatenate([w[1] for w in ww])
    off = np.concatenate([[0], np.cumsum([len(w[0]) for w in ww])]).astype(np.int64)
    ref = np.arange(len(x), dtype=np.int32)
    dx, dy, doff, dref = (sa.DeviceArray.from_numpy(a) for a in (x, y, off, ref))
    cv = sa.Canvas(W, H, dpi, 0xffffff)
    cv.draw(draw_style(sa, st), dx, dy, doff, dref, len(ww), len(x))
    got = cv.read(); cv.close()
    d = np.argwhere(got != want)
    if only is None or len(d):
        print("way", only, "knots", [len(w[0]) for w in ww][:3], "diff px", len(d), d[:12].tolist(),
              [(hex(got[r, c]), hex(want[r, c])) for r, c in d[:6]], flush=True)
