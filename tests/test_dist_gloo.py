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
