"""Diagnostic (GPU box): the chart's PNG and its tiles from the device canvas on the bench's C2 chart
(8192 x 8192 px, 1 M ways with border), against libcairo's own PNG writer on the same pixels (one thread: what
save_main_image, src/smrules_cairo.c:409, spends after the replay).  The GPU's own share (k_png_opaque + k_png_rows) is in
the rocprof kernel statistics of this program.  usage: png_bench.py [size] [ways]"""
import os, sys, time, tempfile, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import smrender_amd as sa
from smrender_amd import synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ways = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
sa.init(0)
x, y, off, ref = synth.synth_ways(ways, size, size, seed=1234)
D = sa.DeviceArray.from_numpy
cv = sa.Canvas(size, size, 300, 0xfff8e0)
cv.draw(sa.DrawStyle(1, 1, 0x4080c0, 0x000000, 0.0, 0.1, 0, 0, 0, 0, 0, 0), D(x), D(y), D(off.astype(np.int64)), D(ref.astype(np.int32)),
        len(off) - 1, len(ref))
sa.sync()
out = {"chart": f"{size}x{size} px, {ways} closed ways fill+border", "runs": []}
tmp = tempfile.mkdtemp(prefix="smr_png_", dir="/tmp")
for level, threads in [(100, 16), (-1, 16), (1, 16), (-1, 1)]:
    f = os.path.join(tmp, "chart.png")
    best = None
    for rep in range(2):
        t0 = time.perf_counter()
        st = cv.write_png(f, level=level, threads=threads)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out["runs"].append({"what": "write_png", "level": level, "threads": threads, "seconds": round(best, 4), "Mpx_per_s": round(size * size / best / 1e6, 1),
                        "deflate_ms": round(st.deflate_ms, 1), "MB_filtered": round(st.bytes_filtered / 1e6, 1), "MB_file": round(st.bytes_written / 1e6, 2),
                        "rgb": bool(st.n_opaque)})
n = size // 256
for level in (100, -1):
    t0 = time.perf_counter()
    st = cv.write_tiles(os.path.join(tmp, f"z{level}"), 0, 0, n, n, threads=16, level=level)
    dt = time.perf_counter() - t0
    out["runs"].append({"what": "write_tiles", "level": level, "tiles": n * n, "threads": 16, "seconds": round(dt, 4), "tiles_per_s": round(n * n / dt),
                        "Mpx_per_s": round(size * size / dt / 1e6, 1), "deflate_ms": round(st.deflate_ms, 1), "MB_files": round(st.bytes_written / 1e6, 2)})
# the reference's way: libcairo's writer (libpng, one thread) on the same pixels -- read back first, which the reference does not need
img = cv.read()
sys.path.insert(0, os.path.join(R, "tests"))
from oracle import pyoracle as po
L = po.cairo()
if L is not None:
    g = os.path.join(tmp, "cairo.png")
    t0 = time.perf_counter()
    assert L.cd_write_png_data(po.P(img), size, size, g.encode()) == 0
    dt = time.perf_counter() - t0
    out["runs"].append({"what": "libcairo write_to_png (CPU, 1 thread)", "seconds": round(dt, 3), "Mpx_per_s": round(size * size / dt / 1e6, 1),
                        "MB_file": round(os.path.getsize(g) / 1e6, 2)})
print(json.dumps(out))
import shutil
shutil.rmtree(tmp, ignore_errors=True)
