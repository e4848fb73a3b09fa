"""BASELINE.md 3.6, run HERE (the container the survey measured the reference in): the CPU restatement of the reference's draw
path (oracle/cairo_drive.c: smrender's Cairo call sequence over the real libcairo) on the survey's input shape -- 20 000
closed 32-gon ways, fill + border, A4 at 300 dpi (2480 x 3508 px) -- next to the survey's 5.6 s / 1.55 Mpx/s for the
reference's replay + rasterize + PNG encode.  usage: python tools/cpu_anchor.py"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from oracle import pyoracle as po
W, H, DPI = 2480, 3508, 300
rng = np.random.default_rng(1)
n, k = 20000, 32
# the survey's set: 32-gons; radius so that the 20 000 ways cover the sheet about as densely as a chart does
cx, cy = rng.uniform(0, W, n), rng.uniform(0, H, n)
r = np.exp(rng.uniform(np.log(4), np.log(64), n))
a = np.linspace(0, 2 * np.pi, k, endpoint=False)
ways = [(np.append(cx[i] + r[i] * np.cos(a), cx[i] + r[i]), np.append(cy[i] + r[i] * np.sin(a), cy[i]), True) for i in range(n)]
st = po.Style(1, 1, 0x4080c0, 0x000000, 0.0, 0.085, 0)
t0 = time.perf_counter()
img = po.cairo_scene([(st, ways)], DPI, W, H, 0xffffff)
t = time.perf_counter() - t0
t0 = time.perf_counter()
assert po.cairo().cd_save_png(float(DPI), W, H, b"/tmp/cpu_anchor.png") == 0
tp = time.perf_counter() - t0
print(f"replay + rasterize + PNG encode alone (save_main_image): {tp:.2f} s = {W * H / 1e6 / tp:.2f} Mpx/s")
print(f"libcairo {po.cairo().cd_version().decode()}: record + replay + rasterize {t:.2f} s = {W * H / 1e6 / t:.2f} Mpx/s "
      f"(survey, reference binary incl. PNG encode: 5.6 s = 1.55 Mpx/s)")
