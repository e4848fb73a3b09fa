"""Diagnostic (GPU box): the chart's KAP file from the device canvas on the bench's C2 chart, against the
reference's way on one host core: its own colour reduction (src/median_cut.c compiled where it lies, oracle/_ref) plus the
restated nearest-colour search and row coder (oracle/kap.c) on a BOUNDED window of the same pixels.
usage: kap_bench.py [size] [ways] [cpu_window]"""
import os, sys, time, tempfile, json, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import smrender_amd as sa
from smrender_amd import synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ways = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
cpu_win = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
sa.init(0)
x, y, off, ref = synth.synth_ways(ways, size, size, seed=1234)
D = sa.DeviceArray.from_numpy
cv = sa.Canvas(size, size, 300, 0xfff8e0)
cv.draw(sa.DrawStyle(1, 1, 0x4080c0, 0x000000, 0.0, 0.1, 0, 0, 0, 0, 0, 0), D(x), D(y), D(off.astype(np.int64)), D(ref.astype(np.int32)),
        len(off) - 1, len(ref))
sa.sync()
tmp = tempfile.mkdtemp(prefix="smr_kap_", dir="/tmp")
f = os.path.join(tmp, "chart.kap")
best = None
for rep in range(2):
    t0 = time.perf_counter()
    st = cv.write_kap(f, "VER/2.0\r\n")
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
out = {"chart": f"{size}x{size} px, {ways} closed ways fill+border",
       "gpu": {"seconds": round(best, 4), "Mpx_per_s": round(size * size / best / 1e6, 1), "colors": st.n_colors, "depth": st.depth,
               "reduce_ms": round(st.reduce_ms, 1), "index_ms": round(st.index_ms, 1), "rows_ms": round(st.rows_ms, 1), "write_ms": round(st.write_ms, 1),
               "points_sorted_M": round(st.sorted_points / 1e6, 1), "MB_file": round(st.file_bytes / 1e6, 2)}}
from oracle import pyoracle as po
win = cv.read(0, 0, cpu_win, cpu_win)
refm = po.ref_median()
pts = po.argb_points(win)
t0 = time.perf_counter()
po.median_cut(pts, 127, ref=refm) if refm is not None else po.median_cut(pts, 127)
t_cut = time.perf_counter() - t0
t0 = time.perf_counter()
body, pal = po.kap_body(win, 9)
t_all = time.perf_counter() - t0
out["cpu_reference"] = {"window": f"{cpu_win}x{cpu_win} px of the same chart", "cores": 1,
                        "median_cut_seconds": round(t_cut, 2), "median_cut_kind": "reference (src/median_cut.c)" if refm is not None else "port",
                        "whole_body_seconds_port": round(t_all, 2), "Mpx_per_s": round(cpu_win * cpu_win / t_all / 1e6, 2)}
print(json.dumps(out))
shutil.rmtree(tmp, ignore_errors=True)
