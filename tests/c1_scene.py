"""BASELINE config 1 as far as it can be had here (the Monaco extract is not in the image, no network): the reference's own test
data (tests/golden/testdata.osm = its test/testdata.osm) at 1:20000 on A4 at 300 dpi (210 x 297 mm: 2480.3 x 3507.9 px, the
canvas round()s to 2480 x 3508), drawn with a sequence of shipped `draw` parameter strings (tests/golden/draw_params.txt lists
them all), every rule applied to every way in bxtree order.  Shared by the generator of the libcairo fixture
(tests/golden/make_c1_fixture.py) and by the GPU test (tests/test_gpu_configs.py)."""
import os
import xml.etree.ElementTree as ET
import numpy as np
from oracle import pyoracle as po

DPI = 300
W_PX, H_PX = 210 / 25.4 * DPI, 297 / 25.4 * DPI          # rd->w, rd->h: doubles (src/smrender.c:358-460)
W, H = int(round(W_PX)), int(round(H_PX))
MEAN_LAT, MEAN_LON, SCALE = 43.0 + 44.06 / 60, 15.0 + 34.35 / 60, 20000.0
BG = 0xffffff
RULES = [   # (parameters as the rulesets give them, the oracle's style)
    ("color=#c0d8f0;bcolor=#204060;bwidth=0.2", po.Style(1, 1, 0xc0d8f0, 0x204060, 0, 0.2, 0)),
    ("color=#e0e0e0;bcolor=#0000ff;curve=1", po.Style(1, 1, 0xe0e0e0, 0x0000ff, 0, 0, 0).with_curve(0.2)),
    ("bcolor=#40FF00FF;bwidth=0.5;bstyle=dashed", po.Style(0, 1, 0, 0x40ff00ff, 0, 0.5, 0).with_dash(border_style=1)),
    ("bcolor=#ff00ff;bstyle=pipe;bwidth=0.2", po.Style(0, 1, 0, 0xff00ff, 0, 0.2, 0).with_dash(border_style=4)),
    ("color=#00600000;width=0.3;curve=1", po.Style(1, 0, 0x00600000, 0, 0.3, 0, 0).with_curve(0.2)),
    ("bcolor=#30000000;bwidth=0;bstyle=dotted", po.Style(0, 1, 0, 0x30000000, 0, 0, 0).with_dash(border_style=2)),
]


def load():
    """nodes {id: (lat, lon)} and ways [(id, [node ids])] in bxtree order (negative ids ascending)"""
    t = ET.parse(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "testdata.osm")).getroot()
    geo = {n.get("id"): (float(n.get("lat")), float(n.get("lon"))) for n in t.findall("node")}
    ways = sorted(((int(w.get("id")), [nd.get("ref") for nd in w.findall("nd")]) for w in t.findall("way")))
    return geo, ways


def oracle_ways(geo, ways):
    op = po.make_proj(MEAN_LAT, MEAN_LON, SCALE, W_PX, H_PX, DPI)
    out = []
    for _, refs in ways:
        la, lo = np.array([geo[r][0] for r in refs]), np.array([geo[r][1] for r in refs])
        ox, oy, _, _, _ = po.project(op, la, lo)
        out.append((ox, oy, len(refs) >= 4 and refs[0] == refs[-1]))
    return out
