import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import smrender_amd as sa
from oracle import pyoracle as po
import test_gpu_golden as tg
sa.init(0)
for name in tg.SCENES:
    meta, rules, want = tg.load_scene(name)
    got = tg.gpu_render(sa, meta, rules)
    d = po.chan_diff(got, want)
    ys, xs = np.nonzero(d > 2)
    print(name, "bad px", len(ys), "max", d.max(), "rows", (ys.min(), ys.max()) if len(ys) else None, "cols", (xs.min(), xs.max()) if len(xs) else None)
    if len(ys):
        print("  sample", list(zip(ys[:8].tolist(), xs[:8].tolist())), [hex(int(got[y, x])) for y, x in zip(ys[:4], xs[:4])], [hex(int(want[y, x])) for y, x in zip(ys[:4], xs[:4])])
