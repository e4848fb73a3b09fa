"""Diagnostic (GPU box): random scenes around the thresholds of the 32-bit edge paths (edges of 8..40 px, i.e. on both sides of
the 16-px limit; widths from hairline to 3 px; 150..900 dpi; closed, open, curves) against the oracle."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import smrender_amd as sa
from oracle import pyoracle as po
from golden_io import draw_style
sa.init(0)
SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(SEED)
bad = 0
DASH = len(sys.argv) > 3 and "dash" in sys.argv[3]
BIG = len(sys.argv) > 3 and "big" in sys.argv[3]
DIR = len(sys.argv) > 3 and "dir" in sys.argv[3]
degraded = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    W, H = int(rng.integers(100, 700)), int(rng.integers(80, 500))
    dpi = int(rng.choice([150, 300, 450, 600, 900]))
    n_ways = int(rng.integers(5, 120))
    seg = rng.uniform(3, 45)
    xs, ys, off, ref = [], [], [0], []
    for w in range(n_ways):
        n = int(rng.integers(2, 20))
        if BIG and rng.uniform() < 0.25:
            n = int(rng.integers(60, 400))             # fills and strokes of several items
        closed = rng.uniform() < 0.5 and n >= 4
        if closed:
            r = min(seg * n / 6.3, 0.6 * max(W, H)) * rng.uniform(0.5, 1.5)
            a = np.sort(rng.uniform(0, 2 * np.pi, n - 1)) * rng.choice([-1, 1])
            cx, cy = rng.uniform(-20, W + 20), rng.uniform(-20, H + 20)
            x, y = cx + r * rng.uniform(0.6, 1.0, n - 1) * np.cos(a), cy + r * rng.uniform(0.6, 1.0, n - 1) * np.sin(a)
            base = len(np.concatenate(xs)) if xs else 0
            xs.append(x); ys.append(y)
            ref += list(range(base, base + n - 1)) + [base]
        else:
            x = rng.uniform(-20, W + 20) + np.cumsum(rng.normal(0, seg, n)); y = rng.uniform(-20, H + 20) + np.cumsum(rng.normal(0, seg, n))
            base = len(np.concatenate(xs)) if xs else 0
            xs.append(x); ys.append(y)
            ref += list(range(base, base + n))
        off.append(len(ref))
    x, y = np.concatenate(xs), np.concatenate(ys)
    if rng.uniform() < 0.3:                      # snap some coordinates to pixel / half-pixel positions
        x = np.round(x * 2) / 2; y = np.round(y * 2) / 2
    off, ref = np.array(off, np.int64), np.array(ref, np.int32)
    fu, bu = [(1, 1), (1, 0), (0, 1)][int(rng.integers(0, 3))]
    fc = int(rng.integers(0, 1 << 24)) | (int(rng.integers(0, 0x60)) << 24 if rng.uniform() < 0.4 else 0)
    bc = int(rng.integers(0, 1 << 24)) | (int(rng.integers(0, 0x60)) << 24 if rng.uniform() < 0.4 else 0)
    st = po.Style(fu, bu, fc, bc, float(rng.choice([0, 0.1, 0.3, 0.8])), float(rng.choice([0, 0.08, 0.2, 0.5])), 0)
    kind = rng.uniform()
    if kind < 0.3:
        st = st.with_curve(float(rng.choice([0.2, 0.35])))
    if (0.15 < kind < 0.5) and DASH:
        # dashed / dotted / rounddot / pipe borders and fill strokes of open ways (round caps are sticky: cairo_smr_dash :909-921)
        S = po.Style
        bs = int(rng.choice([S.DASHED, S.DOTTED, S.ROUNDDOT, S.PIPE, 0])) if bu else 0
        fs = int(rng.choice([0, S.DASHED, S.DOTTED, S.ROUNDDOT, S.PIPE])) if fu else 0
        bd = tuple(float(v) for v in rng.uniform(1, 8, 2)) if rng.uniform() < 0.5 else None
        st = st.with_dash(border_style=bs, fill_style=fs, border_dash=bd)
    if rng.uniform() < 0.2:
        st = st.with_evenodd()
    ways = po.ways_from_arrays(x, y, off, ref)
    directional = DIR and st.curve == 0 and st.fill_used and st.border_style == 0 and st.fill_style == 0 and rng.uniform() < 0.6
    want = po.oracle_directional(st, ways, dpi, W, H, 0xffffff) if directional else po.oracle_scene([(st, ways)], dpi, W, H, 0xffffff)
    cv = sa.Canvas(W, H, dpi, 0xffffff)
    try:
        cv.draw(draw_style(sa, st, directional=1) if directional else draw_style(sa, st),
                *(sa.DeviceArray.from_numpy(a) for a in (x, y, off, ref)), len(off) - 1, len(ref))
        got = cv.read()
        nd = int((got != want).sum())
        if nd and sa.lib().smr_hip_draw_status() & 1:
            # the documented degrade: > 64 row-spanning edges of one primitive, reported by the library
            d = po.chan_diff(got, want)
            print("scene", it, "flagged DEGRADED_ROWS: differing px", nd, "in rows", sorted(set(np.argwhere(got != want)[:, 0].tolist())), "max", int(d.max()))
            degraded += 1
            if d.max() <= 4:
                nd = 0
    except sa.SmrHipError as e:
        nd = -1; print("  error:", e)
    cv.close()
    if nd != 0:
        bad += 1
        print("scene", it, "W,H,dpi", W, H, dpi, "ways", n_ways, "seg", round(seg, 1), "style", fu, bu, st.curve, "directional", directional, "DIFF", nd)
        os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
        np.savez(os.path.join(R, "gpurun_out", f"soak_fail_s{SEED}_{it}.npz"), x=x, y=y, off=off, ref=ref, W=W, H=H, dpi=dpi, style=bytes(st),
                 got=got if nd > 0 else np.zeros(1), want=want)
        if directional:
            continue
        # which ways: one at a time
        for w in range(len(off) - 1):
            o1 = np.array([0, off[w + 1] - off[w]], np.int64); r1 = ref[off[w]:off[w + 1]]
            wy = po.ways_from_arrays(x, y, o1, r1)
            wt = po.oracle_scene([(st, wy)], dpi, W, H, 0xffffff)
            c1 = sa.Canvas(W, H, dpi, 0xffffff)
            try:
                c1.draw(draw_style(sa, st), *(sa.DeviceArray.from_numpy(a) for a in (x, y, o1, np.ascontiguousarray(r1))), 1, len(r1))
            except sa.SmrHipError as e:
                print("   way", w, "error:", e); c1.close(); continue
            g1 = c1.read(); c1.close()
            if not np.array_equal(g1, wt):
                print("   way", w, "refs", len(r1), "closed", bool(r1[0] == r1[-1]), "diff px", int((g1 != wt).sum()), np.argwhere(g1 != wt)[:5].tolist())
print("scenes with differences:", bad, "| scenes with reported sub-sampled rows (within 4 levels):", degraded)
