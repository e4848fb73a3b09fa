"""Loader for the committed fixtures in tests/golden (data only; see make_golden.py)."""
import os
import numpy as np
from oracle import pyoracle as po

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCENES = ["c2_fill_border_300", "open_border_300", "layers_translucent_600", "far_window_300", "dashes_300", "curves_300", "hairpins_300",
          "evenodd_300"]


def load_scene(name):
    z = np.load(os.path.join(GOLDEN, f"scene_{name}.npz"))
    rules = []
    for i in range(int(z["n_rules"])):
        si, sf = z[f"si{i}"], z[f"sf{i}"]
        st = po.Style(int(si[0]), int(si[1]), int(si[2]), int(si[3]), float(sf[0]), float(sf[1]), int(si[4]))
        if len(si) > 5:
            st.border_style, st.fill_style, st.border_ndash, st.fill_ndash = (int(v) for v in si[5:9])
            for k in range(4):
                st.border_dash[k], st.fill_dash[k] = float(sf[2 + k]), float(sf[6 + k])
        if len(si) > 9 and int(si[9]):
            st.curve, st.curve_fact = int(si[9]), float(sf[10])
        if len(si) > 10:
            st.evenodd = int(si[10])
        rules.append((st, (z[f"x{i}"], z[f"y{i}"], z[f"off{i}"], z[f"ref{i}"])))
    meta = {k: int(z[k]) for k in ("dpi", "W", "H", "bg")}
    meta["x0"], meta["y0"] = int(z["win_x0"]), int(z["win_y0"])
    # the page the scene was rendered on (make_golden.py: full_w / full_h; default: it ends with the window): the oracle cuts
    # polygon edges at the PAGE's sides as libcairo's replay does, so it has to know them
    fw, fh = {"layers_translucent_600": (300, 240)}.get(name, (meta["x0"] + meta["W"], meta["y0"] + meta["H"]))
    meta["page"] = (0, 0, fw, fh)
    return meta, rules, z["image"]


# Measured agreement of the oracle with libcairo per fixture, as (share of pixels within +-2, share identical).
# Solid fills, dashes and ordinary strokes are bit-identical; what is left are joins that libcairo's contour
# stroker resolves by winding cancellation (turns sharper than ~150 degrees between segments shorter than the
# line width: the random walks of open_border / layers_translucent are full of them), see DESIGN.md deviations.
BARS = {"c2_fill_border_300": (1.0, 0.99995), "open_border_300": (0.999, 0.998), "layers_translucent_600": (0.9993, 0.996),
        "far_window_300": (1.0, 0.999), "dashes_300": (1.0, 0.9999), "curves_300": (0.9995, 0.998), "hairpins_300": (0.9998, 0.9994),
        "evenodd_300": (1.0, 0.9999)}


def draw_style(sa, st, **kw):
    """oracle Style -> smr_draw_style_t (ctypes mirror)"""
    gs = sa.DrawStyle(st.fill_used, st.border_used, st.fill_col, st.border_col, st.fill_width_mm, st.border_width_mm,
                      st.ignore_open, st.fill_style, st.border_style, 0, kw.get("directional", 0), kw.get("direct", 0))
    if st.curve == 2:
        gs.set_knots()
    elif st.curve:
        gs.set_curve(st.curve_fact)
    gs.set_evenodd(st.evenodd)
    return gs.set_dashes([st.border_dash[i] for i in range(st.border_ndash)], [st.fill_dash[i] for i in range(st.fill_ndash)])


def within(a, b, tol=2):
    """fraction of pixels whose every channel differs by <= tol (north_star: >= 99.5 % at tol 2)"""
    return float((po.chan_diff(a, b) <= tol).mean())
