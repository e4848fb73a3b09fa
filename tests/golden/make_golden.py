#!/usr/bin/env python3
"""Generates the golden fixtures in this directory by driving the REAL libcairo 1.16.0 +
pixman 0.40.0 of this image with the reference's call sequence (oracle/cairo_drive.c, which
cites the smrender lines it restates).  The reference's own tests hold no expected outputs
(test/Makefile only renders PNGs), and its Cairo path cannot be built here under the rules
(autoconf-generated config.h), so these vectors -- outputs of the third-party arithmetic the
reference delegates to, reached through its own call sites -- are what pins the oracle.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz (needs libcairo)
    python tests/golden/make_golden.py hairpins_300      # only the named scenes
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as po  # noqa: E402
from smrender_amd import synth  # noqa: E402


def open_ways(n_ways, W, H, seed):
    rng = np.random.default_rng(seed)
    xs, ys, off, ref = [], [], [0], []
    for _ in range(n_ways):
        n = int(rng.integers(2, 14))
        step = rng.uniform(2, 30)
        xs.append(rng.uniform(0, W) + np.cumsum(rng.normal(0, step, n)))
        ys.append(rng.uniform(0, H) + np.cumsum(rng.normal(0, step, n)))
        ref.extend(range(off[-1], off[-1] + n))
        off.append(off[-1] + n)
    return np.concatenate(xs), np.concatenate(ys), np.array(off, np.int64), np.array(ref, np.int32)


def hairpin_ways(n_ways, W, H, seg_px, margin, seed):
    """sharp turns and segments shorter than the stroke is wide, kept `margin` px inside the window"""
    rng = np.random.default_rng(seed)
    xs, ys, off, ref = [], [], [0], []
    while len(off) - 1 < n_ways:
        n = int(rng.integers(3, 9))
        x = [rng.uniform(margin, W - margin)]; y = [rng.uniform(margin, H - margin)]
        ang = rng.uniform(0, 2 * np.pi)
        for _ in range(n - 1):
            ang += rng.choice([rng.uniform(2.6, 3.6), rng.uniform(-0.5, 0.5), rng.uniform(1.5, 2.5)])
            L = rng.uniform(0.3, 3.0) * seg_px
            x.append(x[-1] + L * np.cos(ang)); y.append(y[-1] + L * np.sin(ang))
        x, y = np.array(x), np.array(y)
        if x.min() < margin or y.min() < margin or x.max() > W - margin or y.max() > H - margin:
            continue
        xs.append(x); ys.append(y); ref.extend(range(off[-1], off[-1] + n)); off.append(off[-1] + n)
    return np.concatenate(xs), np.concatenate(ys), np.array(off, np.int64), np.array(ref, np.int32)


def star_ways(n_ways, W, H, seed, margin=52):
    """closed ways that cross themselves: {n/k} star polygons (pentagrams, heptagrams ...) and random closed walks, so that
    the two fill rules give different pictures (winding numbers 2 and more, and regions wound both ways)"""
    rng = np.random.default_rng(seed)
    xs, ys, off, ref = [], [], [0], []
    for i in range(n_ways):
        # kept inside the page: where two edges of one polygon cross each other BEYOND the page's side, libcairo has projected both
        # onto the side before it looks for crossings, and the restatement has not (known deviations)
        cx, cy, r = rng.uniform(margin, W - margin), rng.uniform(margin, H - margin), rng.uniform(8, 50)
        if i % 3 == 2:
            n = int(rng.integers(5, 11))
            x, y = cx + rng.uniform(-r, r, n), cy + rng.uniform(-r, r, n)
        else:
            n = int(rng.choice([5, 7, 8, 9, 11])); k = 2 if n < 8 else 4 if n == 9 else 3      # n and k coprime: one closed star, no coincident edges
            t = rng.uniform(0, 2 * np.pi) + 2 * np.pi * k * np.arange(n) / n * rng.choice([-1, 1])
            rr = r * rng.uniform(0.8, 1.0, n)
            x, y = cx + rr * np.cos(t), cy + rr * np.sin(t)
        xs.append(x); ys.append(y)
        ref.extend(list(range(off[-1], off[-1] + n)) + [off[-1]])
        off.append(off[-1] + n)
    # node arrays hold n points per way, the refs n + 1 (the closing ref is the first node again)
    x, y = np.concatenate(xs), np.concatenate(ys)
    roff = np.concatenate([[0], np.cumsum(np.diff(off) + 1)]).astype(np.int64)
    return x, y, roff, np.array(ref, np.int32)


def style_tuple(st):
    return np.array([st.fill_used, st.border_used, st.fill_col, st.border_col, st.ignore_open,
                     st.border_style, st.fill_style, st.border_ndash, st.fill_ndash, st.curve, st.evenodd], np.int64), \
        np.array([st.fill_width_mm, st.border_width_mm] + list(st.border_dash) + list(st.fill_dash) + [st.curve_fact])


def main():
    cd = po.cairo()
    assert cd is not None, "libcairo not loadable"
    print("libcairo", cd.cd_version().decode())
    rng = np.random.default_rng(20261110)

    # 1. scan-converter KATs: single polygons filled black on white straight onto an image
    polys, alphas = [], []
    for t in range(48 if not sys.argv[1:] else 0):
        W = int(rng.integers(12, 48))
        n = 3 + t % 9
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        r = rng.uniform(0.15, 0.48, n) * W
        xy = np.stack([W / 2 + r * np.cos(ang), W / 2 + r * np.sin(ang)], 1)
        if t % 3 == 0:
            rng.shuffle(xy)                      # self-intersecting -> exercises the winding rule
        out = np.zeros((W, W), np.uint32)
        cd.cd_direct_fill(po.P(np.ascontiguousarray(xy)), n, 0x000000, 0xffffff, W, W, po.P(out))
        polys.append(xy)
        alphas.append((255 - (out & 0xff)).astype(np.uint8))
    if not sys.argv[1:]:
      np.savez_compressed(os.path.join(HERE, "tor_kat.npz"), n=len(polys),
                        **{f"xy{i}": p for i, p in enumerate(polys)}, **{f"a{i}": a for i, a in enumerate(alphas)})

    # 1b. the same under CAIRO_FILL_RULE_EVEN_ODD (every polygon shuffled: they cross themselves), and the premultiplied pixel of
    # translucent colours (cairo_set_source_rgba -> _cairo_color_double_to_short: one triangle over black, a pixel well inside)
    if not sys.argv[1:] or "kat_evenodd" in sys.argv[1:]:
        r2 = np.random.default_rng(20261111)
        polys, alphas = [], []
        cd.cd_set_direct_fill_rule(1)
        for t in range(48):
            W = int(r2.integers(12, 48))
            n = 4 + t % 9
            xy = np.stack([r2.uniform(0.05, 0.95, n) * W, r2.uniform(0.05, 0.95, n) * W], 1)
            out = np.zeros((W, W), np.uint32)
            cd.cd_direct_fill(po.P(np.ascontiguousarray(xy)), n, 0x000000, 0xffffff, W, W, po.P(out))
            polys.append(xy); alphas.append((255 - (out & 0xff)).astype(np.uint8))
        cd.cd_set_direct_fill_rule(0)
        np.savez_compressed(os.path.join(HERE, "tor_kat_evenodd.npz"), n=len(polys),
                            **{f"xy{i}": p for i, p in enumerate(polys)}, **{f"a{i}": a for i, a in enumerate(alphas)})
        cols = [0x40c04020, 0x4076d93f, 0x4036cf4c] + [int(c) for c in r2.integers(0, 1 << 31, 1500)]
        tri = (np.array([2.0, 60.0, 20.0, 2.0]), np.array([2.0, 10.0, 60.0, 2.0]), True)
        pix = [int(po.cairo_scene([(po.Style(1, 0, c, 0, 0, 0, 0), [tri])], 72, 64, 64, 0x000000)[24, 24]) & 0xffffff for c in cols]
        np.savez_compressed(os.path.join(HERE, "color_pixels.npz"), col=np.array(cols, np.uint32), rgb_over_black=np.array(pix, np.uint32))
        print("kat_evenodd, color_pixels")

    # 2. whole scenes through the reference call sequence (recording surface in pt, group per rule, replay)
    scenes = {}
    c2 = synth.synth_ways(160, 256, 256, seed=77)
    ow = open_ways(50, 256, 200, seed=78)
    scenes["c2_fill_border_300"] = dict(dpi=300, W=256, H=256, bg=0xffffff,
                                        rules=[(po.Style(1, 1, 0x4080c0, 0x000000, 0.0, 25.4 / 300, 0), c2)])
    scenes["open_border_300"] = dict(dpi=300, W=256, H=200, bg=0xffffff,
                                     rules=[(po.Style(0, 1, 0, 0x204060, 0.0, 0.3, 0), ow)])
    # the page is larger than the window: at 600 dpi libcairo clips to the recording surface's extents in whole
    # pt scaled by the 24.8-quantised 600/72, i.e. to y = 199.97 px for a 200-px page, and dims the page's last
    # row by that fraction -- a page-edge effect this fixture keeps out of the window (deviations)
    scenes["layers_translucent_600"] = dict(dpi=600, W=256, H=200, bg=0x7f336699, full_w=300, full_h=240, rules=[
        (po.Style(1, 1, 0x4080c0, 0x000000, 0.0, 25.4 / 300, 0), synth.synth_ways(60, 256, 200, seed=79)),
        (po.Style(1, 1, 0x304080c0, 0x50102030, 0.4, 0.2, 0), ow),
        (po.Style(0, 1, 0, 0x000000, 0.0, 0.0, 0), synth.synth_ways(30, 256, 200, seed=80))])
    # a window far from the origin: pins the 24.8 quantisation of the replay scale
    far = synth.synth_ways(40, 128, 128, seed=81)
    far = (far[0] + 15000.0, far[1] + 9000.0, far[2], far[3])
    scenes["far_window_300"] = dict(dpi=300, W=128, H=128, bg=0xffffff, x0=15000, y0=9000,
                                    rules=[(po.Style(1, 1, 0xc04020, 0x000000, 0.0, 0.1, 0), far)])
    # dash styles (cairo_smr_dash, src/smrules_cairo.c:898-923): one rule per style
    dw = open_ways(14, 256, 200, seed=82)
    dc = synth.synth_ways(10, 256, 200, seed=83, rmin=10.0, rmax=60.0)
    S = po.Style
    scenes["dashes_300"] = dict(dpi=300, W=256, H=200, bg=0xffffff, rules=[
        (S(0, 1, 0, 0x204060, 0, 0.25, 0).with_dash(border_style=S.DASHED), dw),
        (S(1, 1, 0x30c0d8f0, 0x000000, 0.3, 0.1, 0).with_dash(border_style=S.DOTTED), dc),
        (S(0, 1, 0, 0x802010, 0, 0.3, 0).with_dash(border_style=S.ROUNDDOT), open_ways(8, 256, 200, seed=84)),
        (S(0, 1, 0, 0x103050, 0, 0.15, 0).with_dash(border_style=S.PIPE), open_ways(8, 256, 200, seed=85)),
        (S(1, 1, 0x4080c0, 0x50000000, 0.4, 0.1, 0).with_dash(border_style=S.DASHED, fill_style=S.DASHED,
                                                              border_dash=(5, 2, 1, 2)), open_ways(8, 256, 200, seed=86))])
    # curve=1 (cairo_smr_poly_curve, src/smrules_cairo.c:697-749): closed ways crossing the page edge, open ways
    scenes["curves_300"] = dict(dpi=300, W=256, H=200, bg=0xffffff, rules=[
        (S(1, 1, 0x4080c0, 0x000000, 0, 0.2, 0).with_curve(), synth.synth_ways(14, 256, 200, seed=87, rmin=10.0, rmax=60.0)),
        (S(0, 1, 0, 0x204060, 0, 0.3, 0).with_curve(), open_ways(10, 256, 200, seed=88)),
        (S(1, 1, 0x30c0d8f0, 0x000000, 0.4, 0.15, 0).with_curve(0.35), open_ways(8, 256, 200, seed=89))])
    # hairpins: joins whose inner corner pokes through a neighbouring segment shorter than the stroke is wide (the stroker's
    # polygon winds against itself there); a wide border, a narrow one, the fill stroke of open ways
    scenes["hairpins_300"] = dict(dpi=300, W=400, H=300, bg=0xffffff, rules=[
        (S(0, 1, 0, 0x203040, 0, 1.2, 0), hairpin_ways(14, 400, 300, 8, 75, seed=90)),
        (S(0, 1, 0, 0x802010, 0, 0.3, 0), hairpin_ways(20, 400, 300, 3, 20, seed=91)),
        (S(1, 0, 0x6030a060, 0, 0.8, 0, 0), hairpin_ways(14, 400, 300, 6, 50, seed=92))])
    # the even-odd rule (cairo_set_fill_rule; the reference keeps the default, so this pins the rasterizer's second rule only):
    # stars with a border, translucent stars over them, and the same ways once more under the winding rule for contrast
    sw = star_ways(40, 256, 200, seed=93)
    scenes["evenodd_300"] = dict(dpi=300, W=256, H=200, bg=0xffffff, rules=[
        (S(1, 1, 0x4080c0, 0x000000, 0, 0.1, 0).with_evenodd(), sw),
        (S(1, 0, 0x40c04020, 0, 0, 0, 0).with_evenodd(), star_ways(25, 256, 200, seed=94)),
        (S(1, 0, 0x6020a040, 0, 0, 0, 0), star_ways(10, 256, 200, seed=95)),
        (S(1, 1, 0x30c0d8f0, 0x203040, 0, 0.15, 0).with_curve().with_evenodd(), star_ways(8, 256, 200, seed=96, margin=75))])
    only = sys.argv[1:]
    for name, sc in scenes.items():
        if only and name not in only:
            continue
        rules = [(st, po.ways_from_arrays(*arrs)) for st, arrs in sc["rules"]]
        img = po.cairo_scene(rules, sc["dpi"], sc["W"], sc["H"], sc["bg"], sc.get("x0", 0), sc.get("y0", 0),
                             sc.get("full_w"), sc.get("full_h"))
        d = dict(dpi=sc["dpi"], W=sc["W"], H=sc["H"], bg=sc["bg"], win_x0=sc.get("x0", 0), win_y0=sc.get("y0", 0),
                 n_rules=len(sc["rules"]), image=img)
        for i, (st, (x, y, off, ref)) in enumerate(sc["rules"]):
            si, sf = style_tuple(st)
            d.update({f"si{i}": si, f"sf{i}": sf, f"x{i}": x, f"y{i}": y, f"off{i}": off, f"ref{i}": ref})
        np.savez_compressed(os.path.join(HERE, f"scene_{name}.npz"), **d)
        print(name, img.shape)

    # 3. rotated bilinear OVER paint (cairo_smr_diff, src/smrules_cairo.c:1746-1759)
    if sys.argv[1:]:
        return
    d = {}
    for t in range(12):
        dw, dh = int(rng.integers(8, 60)), int(rng.integers(4, 24))
        r = int(round(np.hypot(dw, dh)))
        src = (0xff000000 | rng.integers(0, 1 << 24, (r, r))).astype(np.uint32)
        if t % 4 == 1:
            a = rng.integers(0, 256, (r, r))
            c = rng.integers(0, 256, (r, r, 3)) * a[..., None] // 255
            src = ((a << 24) | (c[..., 0] << 16) | (c[..., 1] << 8) | c[..., 2]).astype(np.uint32)
        dst = (0xff000000 | rng.integers(0, 1 << 24, (dh, dw))).astype(np.uint32)
        n = int(rng.integers(20, 90))
        k = int(rng.integers(0, n))
        out = dst.copy()
        assert cd.cd_rot_paint(po.P(out), dw, dh, po.P(src), r, r, k * 2 * np.pi / n) == 0
        d.update({f"src{t}": src, f"dst{t}": dst, f"out{t}": out, f"ang{t}": np.array([k, n])})
    np.savez_compressed(os.path.join(HERE, "rot_paint.npz"), n=12, **d)
    print("done")


if __name__ == "__main__":
    main()
