#!/usr/bin/env python3
"""tests/golden/c1_testdata_a4.npz: BASELINE config 1's plumbing case (tests/c1_scene.py) through the REAL libcairo 1.16 with the
reference's call sequence (oracle/cairo_drive.c), run here where libcairo is.  The image (2480 x 3508 ARGB32, mostly
background) compresses to a few hundred KB."""
import os
import sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
from oracle import pyoracle as po  # noqa: E402
import c1_scene as c1  # noqa: E402

geo, ways = c1.load()
ow = c1.oracle_ways(geo, ways)
img = po.cairo_scene([(st, ow) for _, st in c1.RULES], c1.DPI, c1.W, c1.H, c1.BG, 0, 0, c1.W_PX, c1.H_PX)
orc = po.oracle_scene([(st, ow) for _, st in c1.RULES], c1.DPI, c1.W, c1.H, c1.BG)
d = po.chan_diff(img, orc)
print("libcairo", po.cairo().cd_version().decode(), img.shape, "painted px", int((img != img[0, 0]).sum()),
      "| oracle: within 2:", float((d <= 2).mean()), "identical:", float((d == 0).mean()), "differing px", int((d > 0).sum()))
np.savez_compressed(os.path.join(HERE, "c1_testdata_a4.npz"), image=img, W=c1.W, H=c1.H, dpi=c1.DPI)
