"""Diagnostic (GPU box): the dash fixture rule by rule and way by way, GPU vs oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import smrender_amd as sa
from oracle import pyoracle as po
from golden_io import load_scene, draw_style

sa.init(0)
meta, rules, want = load_scene("dashes_300")
W, H, dpi, bg = meta["W"], meta["H"], meta["dpi"], meta["bg"]

def gpu(st, x, y, off, ref):
    cv = sa.Canvas(W, H, dpi, bg)
    cv.draw(draw_style(sa, st), sa.DeviceArray.from_numpy(x), sa.DeviceArray.from_numpy(y),
            sa.DeviceArray.from_numpy(off.astype(np.int64)), sa.DeviceArray.from_numpy(ref.astype(np.int32)), len(off) - 1, len(ref))
    out = cv.read(); cv.close(); return out

for ri, (st, (x, y, off, ref)) in enumerate(rules):
    try:
        g = gpu(st, x, y, off, ref)
    except Exception as e:
        print("rule", ri, "ERROR", e); continue
    o = po.oracle_scene([(st, po.ways_from_arrays(x, y, off, ref))], dpi, W, H, bg)
    nd = int((g != o).sum())
    print(f"rule {ri}: bstyle {st.border_style} fstyle {st.fill_style} fill_used {st.fill_used} diff px {nd}, gpu nonbg {(g != g[0,0]).sum()} oracle nonbg {(o != o[0,0]).sum()}")
    if nd == 0: continue
    for w in range(len(off) - 1):
        r = ref[off[w]:off[w + 1]]
        xs, ys = x[r], y[r]
        o1 = np.array([0, len(r)], np.int64); r1 = np.arange(len(r), dtype=np.int32)
        try:
            g1 = gpu(st, xs, ys, o1, r1)
        except Exception as e:
            print("  way", w, "ERROR", e); continue
        oo = po.oracle_scene([(st, po.ways_from_arrays(xs, ys, o1, r1))], dpi, W, H, bg)
        d = g1 != oo
        if d.any():
            yy, xx = np.nonzero(d)
            print(f"  way {w}: n {len(r)} diff {int(d.sum())} bbox x {xx.min()}..{xx.max()} y {yy.min()}..{yy.max()} gpu nonbg {(g1 != g1[0,0]).sum()} oracle nonbg {(oo != oo[0,0]).sum()}")
            k = 0
            for py, px in zip(yy[:6], xx[:6]):
                print(f"     ({px},{py}) gpu {g1[py,px]:08x} oracle {oo[py,px]:08x}")

print("---- crop, rule 3 way 7")
st, (x, y, off, ref) = rules[3]
r = ref[off[7]:off[8]]
xs, ys = x[r], y[r]
o1 = np.array([0, len(r)], np.int64); r1 = np.arange(len(r), dtype=np.int32)
g1 = gpu(st, xs, ys, o1, r1)
oo = po.oracle_scene([(st, po.ways_from_arrays(xs, ys, o1, r1))], dpi, W, H, bg)
for name, im in (("gpu", g1), ("oracle", oo)):
    print(name)
    for yy in range(142, 158):
        print(yy, " ".join("%02x" % (im[yy, xx] & 0xff) for xx in range(176, 194)))
