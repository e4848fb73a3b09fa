"""Diagnostic (GPU box): the plugin path on a chart of many rules.  The miniature host (oracle/hostsim.c + the reference's own
libsmrender) applies N draw rules of M ways each through act_draw_ini/_main/_fini, i.e. bxtree lookups per ref in _main,
upload + projection + draw in _fini; the same batches drawn straight through the C-ABI from resident device arrays are the
comparison.  usage: plugin_bench.py [rules] [ways per rule]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import smrender_amd as sa
from smrender_amd import synth
from oracle import pyoracle as po
import hostsim as hs
n_rules = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n_ways = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
W = H = 4096
DPI = 300
sa.init(0)
proj = sa.Projection.from_window(45.0, 15.0, 25000.0, float(W), float(H), DPI)
sa.set_projection(proj)
hs.set_chart(proj)
sa.lib().smr_hip_plugin_reset()
rules = []
nid = 1
for r in range(n_rules):
    x, y, off, ref = synth.synth_ways(n_ways, W, H, seed=100 + r)
    lat, lon = synth.px_to_geo(proj, x, y)
    base = nid; nid += len(x) + n_ways + 1
    hs.add_nodes(base + np.arange(len(x)), lat, lon)
    wids = []
    for w in range(n_ways):
        wid = base + len(x) + w
        hs.add_way(wid, base + ref[off[w]:off[w + 1]].astype(np.int64))
        wids.append(wid)
    rules.append((wids, (x, y, off.astype(np.int64), ref.astype(np.int32))))
col = lambda r: "#%06x" % ((r * 2654435761) & 0xffffff)
for rep in range(2):
    sa.lib().smr_hip_plugin_reset()
    t0 = time.perf_counter()
    for r, (wids, _) in enumerate(rules):
        assert hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, f"color={col(r)};bcolor=black;bwidth=0.1", wids) == 0
    sa.sync()
    t_plugin = time.perf_counter() - t0
    import ctypes as C
    st8 = (C.c_double * 8)()
    sa.lib().smr_hip_plugin_stats(st8, 8, 1)
    print("plugin seconds: _main (host get_object) %.4f, uploads + projection %.4f, smr_hip_draw %.4f, wait %.4f" % (st8[0], st8[2], st8[3], st8[4]))
dev = [tuple(sa.DeviceArray.from_numpy(a) for a in arrs) for _, arrs in rules]
cv = sa.Canvas(W, H, DPI, 0xffffff)
for rep in range(2):
    cv.clear(); sa.sync()
    t0 = time.perf_counter()
    for r, d in enumerate(dev):
        st = sa.DrawStyle(1, 1, (r * 2654435761) & 0xffffff, 0, 0.0, 0.1, 0, 0, 0, 0, 0, 0)
        cv.draw(st, d[0], d[1], d[2], d[3], n_ways, len(rules[r][1][3]))
    sa.sync()
    t_abi = time.perf_counter() - t0
refs = sum(len(a[3]) for _, a in rules)
print(f"{n_rules} rules x {n_ways} ways ({refs} refs) on {W}x{H}: plugin path {t_plugin * 1e3:.1f} ms ({t_plugin / n_rules * 1e3:.2f} ms per rule, "
      f"{refs / t_plugin / 1e6:.2f} M refs/s), C-ABI from resident arrays {t_abi * 1e3:.1f} ms ({t_abi / n_rules * 1e3:.2f} ms per rule)")
# ---- a caption rule on that chart: 2 000 captions, angle=auto, the reference's cut-outs (default) and crops of the canvas ----
if po.cairo() is not None:
    n_cap = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    rng = np.random.default_rng(7)
    cx, cy = rng.uniform(100, W - 100, n_cap), rng.uniform(100, H - 100, n_cap)
    lat, lon = synth.px_to_geo(proj, cx, cy)
    for mode in ("replay", "crop"):
        os.environ["SMR_HIP_CUTOUT"] = mode
        sa.lib().smr_hip_plugin_reset()
        for r, (wids, _) in enumerate(rules):
            assert hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, f"color={col(r)};bcolor=black;bwidth=0.1", wids) == 0
        sa.sync()
        nids = nid + np.arange(n_cap); nid += n_cap + 1
        hs.add_nodes(nids, lat, lon, [f"name=Name {k % 37}" for k in range(n_cap)])
        t0 = time.perf_counter()
        assert hs.apply_rule(sa.lib(), "cap", hs.OSM_NODE, "key=name;size=2;angle=auto;color=black;halign=east;hide=1", nids) == 0
        dt = time.perf_counter() - t0
        print(f"cap rule, {n_cap} captions after {n_rules} draw rules, SMR_HIP_CUTOUT={mode}: {dt:.3f} s ({n_cap / dt:.0f} captions/s)")
