"""The N>1 path on CPU: two gloo ranks, one band each.  Checks the band split, the way hand-out
(every way reaches every band it can touch; bands rendered by the oracle stitch bit-identically to the
single-canvas render), the halo exchange and the max-over-ranks timing reduction."""
import os
import subprocess
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, os.environ["SMR_ROOT"])
from smrender_amd import dist, synth
from oracle import pyoracle as po
g = dist.Group("gloo")
assert g.world == 2
W, H, dpi = 256, 203, 300
x, y, off, ref = synth.synth_ways(120, W, H, seed=5)
st = po.Style(1, 1, 0x4080c0, 0x000000, 0.0, 0.3, 0)
bands = dist.bands(H, g.world)
assert sum(h for _, h in bands) == H and all(y0 % dist.TILE_H == 0 for y0, _ in bands)
y0, h = bands[g.rank]
keep = dist.ways_for_band(y, off, ref, y0, h, dist.stroke_margin_px(0.0, 0.3, dpi))
sx, sy, soff, sref = synth.select_ways(x, y, off, ref, keep)
band = po.oracle_scene([(st, po.ways_from_arrays(sx, sy, soff, sref))], dpi, W, h, 0xffffff, 0, y0)
full = po.oracle_scene([(st, po.ways_from_arrays(x, y, off, ref))], dpi, W, H, 0xffffff)
assert np.array_equal(band, full[y0:y0 + h]), "band %d differs from the full render" % g.rank
assert 0 < keep.sum() < len(keep)          # the hand-out really drops far ways
halo = 9
above, below = g.exchange_halo(band[:halo], band[-halo:], halo)
if g.rank == 0:
    assert above is None and np.array_equal(below, full[y0 + h:y0 + h + halo])
else:
    assert below is None and np.array_equal(above, full[y0 - halo:y0])
assert g.max(1.0 + g.rank) == 2.0 and g.sum(1.0) == 2.0
g.barrier()
g.close()
print("rank", g.rank, "ok")
'''


CAP_WORKER = r'''
import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.environ["SMR_ROOT"])
from smrender_amd import dist, synth
from oracle import pyoracle as po
g = dist.Group("gloo")
L = po.orc()
W, H, dpi = 256, int(os.environ["SMR_TEST_H"]), 300
x, y, off, ref = synth.synth_ways(140, W, H, seed=11)
st = po.Style(1, 1, 0x4080c0, 0x000000, 0.0, 0.3, 0)
full = po.oracle_scene([(st, po.ways_from_arrays(x, y, off, ref))], dpi, W, H, 0xffffff)
bands = dist.bands(H, g.world)
y0, h = bands[g.rank]
# captions of ONE rule: clusters astride every band boundary (their rectangles and cut-outs cross it), a few elsewhere
rng = np.random.default_rng(3)
cx, cy = [rng.uniform(10, W - 10, 10)], [rng.uniform(5, H - 5, 10)]
for by, bh in bands[1:]:
    if bh > 0:
        cx.append(rng.uniform(60, 120, 6)); cy.append(rng.uniform(by - 9, by + 9, 6))
cx, cy = np.concatenate(cx), np.clip(np.concatenate(cy), 0, H - 0.01)
n = len(cx)
fw, fh, fx, col = 40, 8, 20, 0x102030
r = C.c_int(); steps = L.orc_num_steps(fw, fh, dpi, C.byref(r)); r = r.value
fg = np.zeros((fh, fw), np.uint32); L.orc_make_plane(po.P(fg), fw, fh, fx, col)
crnd = lambda v: int(np.sign(v) * np.floor(abs(v) + 0.5))

def find(pix, row_first, k):
    """find_angle of caption k on pixel rows [row_first, row_first + len(pix)) of the chart (rows outside: clear, as outside the chart)"""
    ox, oy = crnd(cx[k] - r / 2.0), crnd(cy[k] - r / 2.0)
    cut = np.zeros((r, r), np.uint32)
    xs0, ys0, xs1, ys1 = max(ox, 0), max(oy, row_first), min(ox + r, W), min(oy + r, row_first + pix.shape[0])
    if xs1 > xs0 and ys1 > ys0:
        cut[ys0 - oy:ys1 - oy, xs0 - ox:xs1 - ox] = pix[ys0 - row_first:ys1 - row_first, xs0:xs1]
    return L.orc_find_angle(po.P(cut), r, po.P(fg), fw, fh, steps, 0.0, 1.0, None)

def rect_pt(k, a):
    """the rectangle a finished caption leaves (src/smrules_cairo.c:2376-2382): text box at the anchor, rotated, after the flip (:2284)"""
    if np.pi / 2 < a < 3 * np.pi / 2:
        a -= np.pi
    ca, sa = np.cos(a), np.sin(a)
    pts = []
    for ux, uy in ((0, 0), (fx, 0), (fx, -fh), (0, -fh)):
        pts += [(cx[k] + ca * ux - sa * uy) * 72 / dpi, (cy[k] + sa * ux + ca * uy) * 72 / dpi]
    return np.array(pts)

def canvas(pix, row_first):
    oc = L.orc_canvas_new(0, row_first, W, pix.shape[0], dpi, 0xffffff)
    view = np.ctypeslib.as_array(C.cast(L.orc_canvas_pixels_rw(oc), C.POINTER(C.c_uint32)), pix.shape)
    view[:] = pix
    return oc, view

# the reference's way: one caption after the other on one canvas
oc, view = canvas(full, 0)
want = np.zeros(n)
for k in range(n):
    want[k] = find(view, 0, k)
    L.orc_direct_fill_pt(oc, po.P(rect_pt(k, want[k])), 4, col)
L.orc_canvas_free(oc)

# banded: own band + halo strips from the neighbours, levels of independent captions, rectangles of every level on every rank
halo = r // 2 + 2
got = None
if h > 0:
    mine = full[y0:y0 + h]
    above, below = g.exchange_halo(mine[:halo], mine[-halo:], halo)
else:
    above, below = g.exchange_halo(np.zeros((0, W), np.uint32), np.zeros((0, W), np.uint32), halo)
level, n_levels = dist.caption_levels(cx, cy, np.full(n, float(r)))
if h > 0:
    parts = [p for p in (above, mine, below) if p is not None and len(p)]
    ext = np.concatenate(parts)
    row_first = y0 - (0 if above is None else len(above))
    oc, view = canvas(ext, row_first)
    scan = lambda idx: np.array([find(view, row_first, int(k)) for k in idx])
    def paint(idx, ang):
        for k, a in zip(idx, ang):
            L.orc_direct_fill_pt(oc, po.P(rect_pt(int(k), float(a))), 4, col)
else:
    scan = lambda idx: np.zeros(len(idx))
    paint = lambda idx, ang: None
got = dist.scan_captions_banded(g, cy, bands, level, n_levels, scan, paint)
assert n_levels >= 3, n_levels                      # the clusters really depend on each other
assert np.array_equal(got, want), (np.nonzero(got != want)[0].tolist(), g.rank)
# and the dependency matters in this scene: without the rectangles some angles differ
indep = np.array([find(full, 0, k) for k in range(n)])
assert (indep != want).sum() >= 2
g.barrier(); g.close()
print("rank", g.rank, "ok")
'''


WIN_WORKER = r'''
import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.environ["SMR_ROOT"])
from smrender_amd import dist, synth
from oracle import pyoracle as po
g = dist.Group("gloo")
L = po.orc()
L.orc_canvas_set_shift.argtypes = [C.c_void_p, C.c_double, C.c_double]
W, H, dpi = 256, int(os.environ["SMR_TEST_H"]), 300
x, y, off, ref = synth.synth_ways(140, W, H, seed=11)
st = po.Style(1, 1, 0x4080c0, 0x000000, 0.0, 0.3, 0)
bands = dist.bands(H, g.world)
y0, h = bands[g.rank]
rng = np.random.default_rng(3)
cx, cy = [rng.uniform(30, W - 30, 10)], [rng.uniform(30, H - 30, 10)]
for by, bh in bands[1:]:
    if bh > 0 and 40 < by < H - 40:
        cx.append(rng.uniform(60, 120, 6)); cy.append(rng.uniform(by - 9, by + 9, 6))
cx, cy = np.concatenate(cx), np.concatenate(cy)
n = len(cx)
fw, fh, fx, col = 40, 8, 20, 0x102030
r = C.c_int(); steps = L.orc_num_steps(fw, fh, dpi, C.byref(r)); r = r.value
r_pt = np.hypot(fw, fh) * 72 / dpi
fg = np.zeros((fh, fw), np.uint32); L.orc_make_plane(po.P(fg), fw, fh, fx, col)

def rect_pt(k, a):
    if np.pi / 2 < a < 3 * np.pi / 2:
        a -= np.pi
    ca, sa = np.cos(a), np.sin(a)
    pts = []
    for ux, uy in ((0, 0), (fx, 0), (fx, -fh), (0, -fh)):
        pts += [(cx[k] + ca * ux - sa * uy) * 72 / dpi, (cy[k] + sa * ux + ca * uy) * 72 / dpi]
    return np.array(pts)

def find(ways, rects, k):
    """find_angle on the reference's cut-out of caption k: the ways and the rectangles so far rendered again with the 24.8 shift
    of a replay translated by (-x + r/2, -y + r/2) pt (cairo_smr_cut_out; oracle: orc_canvas_set_shift)"""
    c = L.orc_canvas_new(0, 0, r, r, dpi, 0xffffff)
    L.orc_canvas_set_shift(c, -cx[k] * 72 / dpi + r_pt / 2, -cy[k] * 72 / dpi + r_pt / 2)
    L.orc_rule_begin(c)
    for wx, wy, closed in ways:
        wx = np.ascontiguousarray(wx, np.float64); wy = np.ascontiguousarray(wy, np.float64)
        L.orc_draw_way(c, C.byref(st), po.P(wx), po.P(wy), len(wx), int(closed), 0, 0, 0, 0)
    L.orc_rule_end(c)
    for pts in rects:
        L.orc_direct_fill_pt(c, po.P(pts), 4, col)
    cut = np.ctypeslib.as_array(C.cast(L.orc_canvas_pixels(c), C.POINTER(C.c_uint32)), (r, r)).copy()
    L.orc_canvas_free(c)
    return L.orc_find_angle(po.P(cut), r, po.P(fg), fw, fh, steps, 0.0, 1.0, None)

# the reference's way: one caption after the other, every way of the chart
all_ways = po.ways_from_arrays(x, y, off, ref)
want, rects = np.zeros(n), []
for k in range(n):
    want[k] = find(all_ways, rects, k)
    rects.append(rect_pt(k, want[k]))

# banded: a rank holds only the ways within a window's reach of its band, and scans the captions anchored in its band
margin = dist.window_margin_px(r, dist.stroke_margin_px(0.0, 0.3, dpi))
keep = dist.ways_for_band(y, off, ref, y0, h, margin) if h > 0 else np.zeros(len(off) - 1, bool)
mine = po.ways_from_arrays(*synth.select_ways(x, y, off, ref, keep))
level, n_levels = dist.caption_levels(cx, cy, np.full(n, float(r)))
seen = []
scan = lambda idx: np.array([find(mine, seen, int(k)) for k in idx])
def paint(idx, ang):
    for k, a in zip(idx, ang):
        seen.append(rect_pt(int(k), float(a)))
got = dist.scan_captions_banded(g, cy, bands, level, n_levels, scan, paint)
assert n_levels >= 3, n_levels
assert np.array_equal(got, want), (np.nonzero(got != want)[0].tolist(), g.rank)
if g.world > 1 and h > 0:
    assert keep.sum() < len(keep)                      # the rank really holds a part of the chart only
g.barrier(); g.close()
print("rank", g.rank, "ok")
'''


EMPTY_BAND_WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, os.environ["SMR_ROOT"])
from smrender_amd import dist
g = dist.Group("gloo")
W, H, halo = 64, 12, 6
full = (np.arange(W * H, dtype=np.uint32) * 2654435761 % 16777216).reshape(H, W)
bands = dist.bands(H, g.world)                      # 4 ranks, 3 tile rows: the last rank's band is empty
assert bands[-1][1] == 0 and all(h > 0 for _, h in bands[:-1])
y0, h = bands[g.rank]
mine = full[y0:y0 + h]
above, below = g.exchange_halo(mine[:halo], mine[max(0, h - halo):], halo)
if h == 0:
    assert above is None and below is None          # nothing posted to or by a rank without rows
else:
    lo, hi = bands[g.rank - 1] if g.rank > 0 else (0, 0), bands[g.rank + 1] if g.rank + 1 < g.world else (0, 0)
    if lo[1] > 0:
        assert np.array_equal(above, full[max(lo[0], y0 - halo):y0])
    else:
        assert above is None
    if hi[1] > 0:
        assert np.array_equal(below, full[y0 + h:min(y0 + h + halo, hi[0] + hi[1])])
    else:
        assert below is None
g.barrier(); g.close()
print("rank", g.rank, "ok")
'''


def _run(tmp_path, text, world, port, extra_env=None):
    script = tmp_path / "worker.py"
    script.write_text(text)
    env = dict(os.environ, SMR_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("ok") == world


def test_same_rule_captions_across_band_boundaries_2_ranks(tmp_path):
    """SURVEY 8e(2), src/smrules_cairo.c:2376-2382: captions of one rule near a band boundary see each other's rectangles
    whichever rank scans them: the banded scan (levels, angles summed over the ranks, rectangles painted by every rank)
    gives the angles of the reference's one-after-the-other scan on one canvas"""
    _run(tmp_path, CAP_WORKER, 2, 29519, {"SMR_TEST_H": "208"})


def test_same_rule_captions_8_ranks_ragged_height(tmp_path):
    """8 ranks on a chart whose height is no multiple of 32: short last band, halo strips of unequal length"""
    _run(tmp_path, CAP_WORKER, 8, 29521, {"SMR_TEST_H": "203"})


def test_captions_on_re_rasterised_cutouts_need_no_halo(tmp_path):
    """the scan on the reference's own cut-outs (windows rendered again at each anchor's sub-pixel offset) over row
    bands: no pixel exchange at all -- a rank holds the ways within dist.window_margin_px of its band, scans the captions
    anchored in its band, and the levels' angles are summed over the ranks as before; 3 ranks give the angles of the
    one-after-the-other scan with every way of the chart"""
    _run(tmp_path, WIN_WORKER, 3, 29523, {"SMR_TEST_H": "300"})


def test_halo_exchange_with_an_empty_band(tmp_path):
    """more ranks than tile rows: the last rank has no rows, posts nothing and is sent nothing (Group.exchange_halo pairs
    neighbours only if both have rows, like exchange_halo_device); the others get their neighbours' strips, shorter than
    the halo where a band is"""
    _run(tmp_path, EMPTY_BAND_WORKER, 4, 29525)


def test_two_gloo_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, SMR_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("ok") == 2


def test_bands_cover_every_row():
    from smrender_amd import dist
    for H in (1, 4, 203, 16384, 28087):
        for world in (1, 2, 3, 8):
            b = dist.bands(H, world)
            assert len(b) == world and b[0][0] == 0 and sum(h for _, h in b) == H
            assert all(b[i][0] + b[i][1] == b[i + 1][0] or b[i + 1][1] == 0 for i in range(world - 1))
