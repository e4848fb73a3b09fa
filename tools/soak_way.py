"""Diagnostic (GPU box): one way of a scene dumped by tools/soak.py, drawn in variants (fill only, border only, black on white,
prefixes and suffixes of the way) against the oracle, to narrow a difference down to a primitive."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import smrender_amd as sa
from oracle import pyoracle as po
from golden_io import draw_style
sa.init(0)
d = np.load(sys.argv[1]); w = int(sys.argv[2])
x, y, off, ref = d["x"], d["y"], d["off"], d["ref"]
W, H, dpi = int(d["W"]), int(d["H"]), int(d["dpi"])
st0 = po.Style.from_saved(d["style"])
r_all = np.ascontiguousarray(ref[off[w]:off[w + 1]])


def run(tag, st, r1):
    o1 = np.array([0, len(r1)], np.int64)
    wy = po.ways_from_arrays(x, y, o1, r1)
    wt = po.oracle_scene([(st, wy)], dpi, W, H, 0xffffff)
    c1 = sa.Canvas(W, H, dpi, 0xffffff)
    c1.draw(draw_style(sa, st), *(sa.DeviceArray.from_numpy(a) for a in (x, y, o1, np.ascontiguousarray(r1))), 1, len(r1))
    g1 = c1.read(); c1.close()
    dd = np.argwhere(g1 != wt)
    print(tag, "diff", len(dd), [(int(r), int(c), hex(int(g1[r, c])), hex(int(wt[r, c]))) for r, c in dd[:8]], flush=True)
    return len(dd)


def variant(**kw):
    s = po.Style.from_buffer_copy(bytes(st0))
    for k, v in kw.items():
        setattr(s, k, v)
    return s


run("as is", st0, r_all)
run("fill only", variant(border_used=0), r_all)
run("border only", variant(fill_used=0), r_all)
run("fill only black", variant(border_used=0, fill_col=0), r_all)
run("border only black", variant(fill_used=0, border_col=0), r_all)
run("no curve", variant(curve=0), r_all)
for k in range(2, len(r_all)):
    run(f"prefix {k}", st0, r_all[:k])
for k in range(1, len(r_all) - 1):
    run(f"suffix {k}", st0, r_all[k:])
