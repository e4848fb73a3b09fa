"""Run on the GPU box: the caption scan's windows (smr_hip_draw_windows) against the oracle's shifted renders
(orc_canvas_set_shift) over random charts, styles, window sizes and anchors.  tools/soak.py does the same for the chart's own
canvas; this one is for what only windows have: the limits of the image libcairo replays into, per window, and the row decisions
beyond them.  usage: soak_windows.py SEED ITERATIONS; one line per iteration that differs, one summary line at the end."""
import ctypes as C
import os
import sys
import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import pyoracle as po
import smrender_amd as sa
from golden_io import draw_style

SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ITER = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ATLAS = len(sys.argv) > 3 and sys.argv[3] == "atlas"        # the oracle renders window k at the atlas's rows k * pitch, as the device does
PROBE = len(sys.argv) > 3 and sys.argv[3] == "probe"        # take the first differing window apart way by way, then stop
sa.init(0)
L = po.orc()
L.orc_canvas_set_shift.argtypes = [C.c_void_p, C.c_double, C.c_double]
L.orc_canvas_move_rows.argtypes = [C.c_void_p, C.c_int]
D = sa.DeviceArray.from_numpy
rng = np.random.default_rng(SEED)


def style(kind, rng):
    fc, bc = int(rng.integers(0, 1 << 24)), int(rng.integers(0, 1 << 24))
    if kind == "fill":
        return po.Style(1, 0, fc, 0, 0, 0, 0)
    if kind in ("pieces", "polygon"):
        return po.Style(1, 1, fc, bc, float(rng.choice([0, 0.3, 0.5])), float(rng.uniform(0.08, 0.4)), 0)
    if kind == "translucent":
        return po.Style(1, 1, fc | int(rng.integers(0x20, 0xe0)) << 24, bc | int(rng.integers(0x20, 0xe0)) << 24, 0, float(rng.uniform(0.1, 0.3)), 0)
    if kind == "dashed":
        return po.Style(0, 1, 0, bc, 0, float(rng.uniform(0.15, 0.4)), 0).with_dash(border_style=1)
    return po.Style(0, 1, 0, bc, float(rng.uniform(0.2, 0.6)), float(rng.uniform(0.1, 0.3)), 0)      # "open": strokes only


bad = tot = 0
for it in range(ITER):
    kind = ["fill", "pieces", "polygon", "translucent", "dashed", "open"][it % 6]
    dpi = int(rng.choice([300, 300, 127, 600]))
    Wc, Hc = int(rng.integers(500, 1400)), int(rng.integers(400, 1100))
    x, y, off, ref = po.synth_ways(int(rng.integers(300, 2500)), Wc, Hc, seed=int(rng.integers(1 << 30)))
    x2, y2, off2, ref2 = po.synth_ways(int(rng.integers(100, 800)), Wc, Hc, seed=int(rng.integers(1 << 30)))
    keep = np.ones(len(ref2), bool); keep[off2[1:] - 1] = False
    ref2, off2 = ref2[keep], off2 - np.arange(len(off2))
    ways = po.ways_from_arrays(x, y, off, ref) + po.ways_from_arrays(x2, y2, off2, ref2)
    X, Y = np.concatenate([x, x2]), np.concatenate([y, y2])
    OFF = np.concatenate([off, off2[1:] + off[-1]]).astype(np.int64)
    REF = np.concatenate([ref, ref2 + len(x)]).astype(np.int32)
    box = np.array([[w[0].min(), w[0].max(), w[1].min(), w[1].max()] for w in ways])
    sty = style(kind, rng)
    for k in ("SMR_HIP_STROKE", "ORC_STROKE_OUTLINE"):
        os.environ.pop(k, None)
    if kind == "polygon":
        os.environ["SMR_HIP_STROKE"] = "polygon"; os.environ["ORC_STROKE_OUTLINE"] = "1"
    n, r = int(rng.integers(4, 12)), int(rng.integers(31, 280))
    if os.environ.get("SOAK_R_ALIGN"):                                     # window sizes (and so the atlas pitch) in multiples of this
        q = int(os.environ["SOAK_R_ALIGN"]); r = max(q, (r + q - 1) // q * q)
    cx, cy = rng.uniform(-10, Wc + 10, n), rng.uniform(-10, Hc + 10, n)
    if it % 4 == 3:                                                        # whole and half pixels too
        cx, cy = np.round(cx * 2) / 2, np.round(cy * 2) / 2
    bg = 0xfff8e0
    at = sa.Canvas.windows(Wc, Hc, dpi, bg, cx, cy, r)
    at.draw_windows(draw_style(sa, sty), D(X), D(Y), D(OFF), D(REF), len(OFF) - 1, len(REF))
    img = at.read()
    status = sa.lib().smr_hip_draw_status()
    pitch = at.window_pitch
    at.close()
    r_pt = r * 72 / dpi
    reach = 60
    diffs = []
    for k in range(n):
        c = L.orc_canvas_new(0, 0, r, r, dpi, bg)
        tx, ty = -cx[k] * 72 / dpi + r_pt / 2, -cy[k] * 72 / dpi + r_pt / 2
        L.orc_canvas_set_shift(c, tx, ty)
        if ATLAS:
            L.orc_canvas_move_rows(c, k * pitch)
        L.orc_rule_begin(c)
        m = r / 2 + reach
        for i in np.nonzero(~((box[:, 1] < cx[k] - m) | (box[:, 0] > cx[k] + m) | (box[:, 3] < cy[k] - m) | (box[:, 2] > cy[k] + m)))[0]:
            wx, wy, closed = ways[i]
            L.orc_draw_way(c, C.byref(sty), po.P(np.ascontiguousarray(wx, np.float64)), po.P(np.ascontiguousarray(wy, np.float64)), len(wx), int(closed), 0, 0, 0, 0)
        L.orc_rule_end(c)
        want = np.ctypeslib.as_array(C.cast(L.orc_canvas_pixels(c), C.POINTER(C.c_uint32)), (r, r)).copy()
        L.orc_canvas_free(c)
        got = img[k * pitch:k * pitch + r]
        x0, y0 = -tx * dpi / 72, -ty * dpi / 72
        xs, ys = np.arange(r) + x0, np.arange(r) + y0
        inside = np.zeros((r, r), bool)
        inside[np.ix_((ys >= 1) & (ys + 1 <= Hc - 1), (xs >= 1) & (xs + 1 <= Wc - 1))] = True
        nd = int((got[inside] != want[inside]).sum())
        tot += 1
        if nd:
            d = po.chan_diff(got, want); d = d.max(axis=-1) if d.ndim == 3 else d
            diffs.append((k, nd, int(d[inside].max()), np.argwhere((d > 0) & inside)[:3].tolist()))
    if diffs and PROBE:
        k, _, _, where = diffs[0]
        py, px = where[0]
        gx, gy = cx[k] - r / 2 + px, cy[k] - r / 2 + py
        print(f"probe: it {it} {kind} dpi {dpi} r {r} window {k} anchor ({cx[k]!r}, {cy[k]!r}) pixel ({px}, {py}) style", [getattr(sty, f[0]) for f in sty._fields_][:8])
        def both(sel, allwin=False):
            """the window with the ways `sel` only, on the GPU and through the oracle"""
            xs_, ys_, of_, rf_ = [], [], [0], []
            for i in sel:
                wx, wy, closed = ways[i]
                m = len(wx) - (1 if closed else 0)
                base = sum(len(a) for a in xs_)
                xs_.append(np.asarray(wx[:m], np.float64)); ys_.append(np.asarray(wy[:m], np.float64))
                rf_.append(np.concatenate([np.arange(m), [0]] if closed else [np.arange(m)]) + base)
                of_.append(of_[-1] + len(rf_[-1]))
            at = sa.Canvas.windows(Wc, Hc, dpi, bg, cx, cy, r) if allwin else sa.Canvas.windows(Wc, Hc, dpi, bg, cx[k:k + 1], cy[k:k + 1], r)
            at.draw_windows(draw_style(sa, sty), D(np.concatenate(xs_)), D(np.concatenate(ys_)), D(np.array(of_, np.int64)), D(np.concatenate(rf_).astype(np.int32)), len(sel), of_[-1])
            g1 = at.read()[k * pitch:k * pitch + r].copy() if allwin else at.read()[:r].copy()
            at.close()
            c = L.orc_canvas_new(0, 0, r, r, dpi, bg)
            L.orc_canvas_set_shift(c, -cx[k] * 72 / dpi + r_pt / 2, -cy[k] * 72 / dpi + r_pt / 2)
            L.orc_rule_begin(c)
            for i in sel:
                wx, wy, closed = ways[i]
                L.orc_draw_way(c, C.byref(sty), po.P(np.ascontiguousarray(wx, np.float64)), po.P(np.ascontiguousarray(wy, np.float64)), len(wx), int(closed), 0, 0, 0, 0)
            L.orc_rule_end(c)
            w1 = np.ctypeslib.as_array(C.cast(L.orc_canvas_pixels(c), C.POINTER(C.c_uint32)), (r, r)).copy()
            L.orc_canvas_free(c)
            return g1, w1
        sel = [int(i) for i in np.nonzero(~((box[:, 1] < gx - reach) | (box[:, 0] > gx + reach) | (box[:, 3] < gy - reach) | (box[:, 2] > gy + reach)))[0]]
        g1, w1 = both(sel)
        print("probe:", len(sel), "ways near the pixel; together they differ at", np.argwhere(g1 != w1)[:6].tolist())
        g2, w2 = both(sel, True)
        print("probe: the same ways, all windows:", np.argwhere(g2 != w2)[:6].tolist())
        g2, w2 = both(list(range(len(ways))))
        print("probe: all ways, this window alone:", np.argwhere(g2 != w2)[:6].tolist())
        g2, w2 = both(list(range(len(ways))), True)
        print("probe: all ways, all windows:", np.argwhere(g2 != w2)[:6].tolist())
        if g2[py, px] != w2[py, px] and g1[py, px] == w1[py, px]:
            # only the whole job differs: which windows and which far ways does it take?  (a far way cannot change the right value)
            want_px = w2[py, px]
            def gpu_px(wsel, wins):
                xs_, ys_, of_, rf_ = [], [], [0], []
                base = 0
                for i in wsel:
                    wx, wy, closed = ways[i]
                    m = len(wx) - (1 if closed else 0)
                    xs_.append(np.asarray(wx[:m], np.float64)); ys_.append(np.asarray(wy[:m], np.float64))
                    rf_.append(np.concatenate([np.arange(m), [0]] if closed else [np.arange(m)]) + base); base += m
                    of_.append(of_[-1] + len(rf_[-1]))
                at = sa.Canvas.windows(Wc, Hc, dpi, bg, cx[wins], cy[wins], r)
                at.draw_windows(draw_style(sa, sty), D(np.concatenate(xs_)), D(np.concatenate(ys_)), D(np.array(of_, np.int64)), D(np.concatenate(rf_).astype(np.int32)), len(wsel), of_[-1])
                kk = list(wins).index(k)
                v = at.read()[kk * pitch + py, px]; at.close()
                return v
            wins = list(range(n)); wsel = list(range(len(ways)))
            for j in range(n):
                if j != k and gpu_px(wsel, [w_ for w_ in wins if w_ != j]) != want_px:
                    wins = [w_ for w_ in wins if w_ != j]
            print("probe: windows needed:", wins, "anchors", [(float(cx[j]), float(cy[j])) for j in wins])
            near = set(sel)
            step = 1024
            while step >= 1:
                pos = 0
                while pos < len(wsel):
                    chunk = [i for i in wsel[pos:pos + step] if i not in near]
                    t = [i for i in wsel if i not in set(chunk)]
                    if chunk and gpu_px(t, wins) != want_px:
                        wsel = t
                    else:
                        pos += step
                step //= 2
            far = [i for i in wsel if i not in near]
            print("probe: far ways needed:", far, "of", len(ways), "ways; value gpu %08x oracle %08x" % (gpu_px(wsel, wins), want_px))
            for i in far:
                wx, wy, closed = ways[i]
                print("probe: far way", i, "closed", int(closed), "box", box[i].tolist(), "nodes", len(wx))
            for env in ("SMR_HIP_COVER", "SMR_HIP_LINEAR", "SMR_HIP_SPLIT", "SMR_HIP_BLEND", "SMR_HIP_OVERLAP"):
                os.environ[env] = "0"
                print("probe:", env + "=0:", "still differs" if gpu_px(wsel, wins) != want_px else "equal")
                del os.environ[env]
            # and the near ways it takes (the right value changes with them: the oracle again per trial)
            def want_of(ws):
                c = L.orc_canvas_new(0, 0, r, r, dpi, bg)
                L.orc_canvas_set_shift(c, -cx[k] * 72 / dpi + r_pt / 2, -cy[k] * 72 / dpi + r_pt / 2)
                L.orc_rule_begin(c)
                for i in ws:
                    wx, wy, closed = ways[i]
                    L.orc_draw_way(c, C.byref(sty), po.P(np.ascontiguousarray(wx, np.float64)), po.P(np.ascontiguousarray(wy, np.float64)), len(wx), int(closed), 0, 0, 0, 0)
                L.orc_rule_end(c)
                v = np.ctypeslib.as_array(C.cast(L.orc_canvas_pixels(c), C.POINTER(C.c_uint32)), (r, r))[py, px]
                L.orc_canvas_free(c)
                return v
            for i in [j for j in wsel if j in near]:
                t = [j for j in wsel if j != i]
                if gpu_px(t, wins) != want_of(t):
                    wsel = t
            print("probe: ways needed in all:", wsel, "gpu %08x oracle %08x" % (gpu_px(wsel, wins), want_of(wsel)))
            for i in wsel:
                wx, wy, closed = ways[i]
                print("probe: way", i, "closed", int(closed)); print("probe: x =", np.asarray(wx).tolist()); print("probe: y =", np.asarray(wy).tolist())
        if g1[py, px] != w1[py, px]:
            for i in list(sel):
                t = [j for j in sel if j != i]
                if t:
                    g1, w1 = both(t)
                    if g1[py, px] != w1[py, px]:
                        sel = t
            g1, w1 = both(sel)
            print("probe: the ways that are needed:", sel, "gpu %08x oracle %08x" % (g1[py, px], w1[py, px]))
            for i in sel:
                wx, wy, closed = ways[i]
                print("probe: way", i, "closed", int(closed)); print("probe: x =", np.asarray(wx).tolist()); print("probe: y =", np.asarray(wy).tolist())
        sys.exit(0)
    if diffs:
        bad += len(diffs)
        print(f"seed {SEED} it {it} {kind} dpi {dpi} r {r} chart {Wc}x{Hc}: status {status}: windows differing (k, px, max level, where): {diffs}", flush=True)
    elif it % 10 == 9:
        print(f"seed {SEED} it {it}: clean so far ({tot} windows)", flush=True)
print(f"soak_windows seed {SEED}: {tot} windows in {ITER} charts, {bad} differ from the oracle")
