"""The C-ABI library loads on a GPU-less host and exports every symbol include/smr_hip.h declares;
compute entry points fail loudly without a device (no CPU fallback)."""
import os
import re
import numpy as np
import smrender_amd as sa
from smrender_amd import hip as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "smr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(smr_hip_\w+|act_\w+_(?:ini|main|fini))\s*\(", src)))


def test_every_declared_symbol_is_exported():
    L = sa.lib()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/smr_hip.h but not exported by libsmr_hip.so"
    assert set(H.EXPORTS) <= set(syms)


def test_struct_layouts_match_header():
    assert sa.Projection.dpi.offset == 72 and np.dtype(np.float64).itemsize * 9 == 72
    assert sa.DrawStyle.fill_width_mm.offset == 16 and sa.DrawStyle.ignore_open.offset == 32


def test_host_side_projection_constants_need_no_gpu():
    p = sa.Projection.from_window(45.0, 15.0, 100000.0, 16384.0, 16384.0, 300)
    from oracle import pyoracle as po
    q = po.make_proj(45.0, 15.0, 100000.0, 16384.0, 16384.0, 300)
    for f in ("lon0", "w", "h", "wc", "lath", "lath_len", "lat_ll", "lon_ru", "lat_ru"):
        assert getattr(p, f) == getattr(q, f)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        return
    L = sa.lib()
    assert L.smr_hip_init(0) == -1
    assert L.smr_hip_project(None, None, 4, None, None, None, None, None) == -1
    assert b"no CPU fallback" in L.smr_hip_last_error()
