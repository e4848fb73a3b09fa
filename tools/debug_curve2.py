"""Diagnostic (GPU box): isolate the ways behind a curve mismatch."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import smrender_amd as sa
from smrender_amd import synth
from oracle import pyoracle as po
from test_gpu_raster import gpu_scene, oracle_of
sa.init(0)
S = po.Style
x, y, off, ref = po.synth_ways(3000, 1024, 1024, seed=8)
x0, y0, x1, y1 = synth.way_bbox(x, y, off, ref)
px, py = 76.5, 591.5
keep = (x0 - 6 < px) & (x1 + 6 > px) & (y0 - 6 < py) & (y1 + 6 > py)
print("ways near", np.nonzero(keep)[0].tolist())
for wi in np.nonzero(keep)[0]:
    k = np.zeros(len(off) - 1, bool); k[wi] = True
    arrs = synth.select_ways(x, y, off, ref, k)
    for label, st in (("fill", S(1, 0, 0x4080c0, 0, 0, 0, 0)), ("border2bw", S(0, 1, 0, 0, 0, 2 * 25.4 / 300, 0)), ("both", S(1, 1, 0x4080c0, 0, 0, 25.4 / 300, 0))):
        st = st.with_curve()
        g = gpu_scene(sa, [(st, arrs)], 600, 1024, 1024, 0xffffff)
        o = oracle_of([(st, arrs)], 600, 1024, 1024, 0xffffff)
        d = po.chan_diff(g, o)
        print(wi, label, "ndiff", int((d > 0).sum()), "max", int(d.max()), np.argwhere(d > 0)[:4].tolist())
    r = ref[off[wi]:off[wi + 1]]
    np.save(os.path.join(R, "gpurun_out", f"way_{wi}.npy"), np.stack([x[r], y[r]]))
