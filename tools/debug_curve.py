"""Diagnostic (GPU box): curve=1 scenes, GPU vs oracle."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import smrender_amd as sa
from oracle import pyoracle as po
from golden_io import draw_style
from test_gpu_raster import gpu_scene, oracle_of, _open_ways, STYLES
sa.init(0)
S = po.Style
def run(label, st, arrs, dpi=300, W=256, H=200):
    try:
        g = gpu_scene(sa, [(st, arrs)], dpi, W, H, 0xffffff)
    except Exception as e:
        print(label, "ERROR", e); return
    o = oracle_of([(st, arrs)], dpi, W, H, 0xffffff)
    d = po.chan_diff(g, o)
    print(f"{label}: identical {(d == 0).mean():.6f} ndiff {(d > 0).sum()} max {d.max()} first {np.argwhere(d > 0)[:4].tolist()}")
closed_in = po.synth_ways(12, 200, 150, seed=5, rmin=10.0, rmax=40.0)
closed_in = (closed_in[0] * 0.6 + 50, closed_in[1] * 0.6 + 40, closed_in[2], closed_in[3])
closed_x = po.synth_ways(40, 256, 200, seed=6, rmin=10.0, rmax=60.0)
ow = _open_ways(30, 256, 200, seed=7)
run("fill closed inside", S(1, 0, 0x4080c0, 0, 0, 0, 0).with_curve(), closed_in)
run("border closed inside", S(0, 1, 0, 0x000000, 0, 0.3, 0).with_curve(), closed_in)
run("fill+border closed inside", S(1, 1, 0x4080c0, 0x000000, 0, 0.2, 0).with_curve(), closed_in)
run("fill closed crossing", S(1, 0, 0x4080c0, 0, 0, 0, 0).with_curve(), closed_x)
run("border closed crossing", S(0, 1, 0, 0x000000, 0, 0.3, 0).with_curve(), closed_x)
run("fill+border closed crossing", S(1, 1, 0x4080c0, 0x000000, 0, 0.2, 0).with_curve(), closed_x)
run("hairline closed crossing", S(0, 1, 0, 0x000000, 0, 0.0, 0).with_curve(), closed_x)
run("border open", S(0, 1, 0, 0x204060, 0, 0.3, 0).with_curve(), ow)
run("fill+border open f=.35", S(1, 1, 0x30c0d8f0, 0x000000, 0.4, 0.15, 0).with_curve(0.35), ow)
run("border open 600", S(0, 1, 0, 0x204060, 0, 0.3, 0).with_curve(), ow, dpi=600)
big = po.synth_ways(3000, 1024, 1024, seed=8)
run("fill+border 3000 ways 1024", S(1, 1, 0x4080c0, 0x000000, 0, 25.4 / 300, 0).with_curve(), big, W=1024, H=1024)
