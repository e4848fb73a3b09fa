"""smr_hip_caprot_noalloc (include/smr_hip.h): the caption scan for callers that cannot allocate.  Its tables live in
a static device array of the library instead of the canvas's work buffers; the results must equal smr_hip_caprot's
bit for bit (chosen angles and the raw per-angle differences), on two calls in a row, the second with a smaller input
(fewer captions AND a smaller plane, so fewer trial angles: the array keeps the first call's longer tables)."""
import ctypes as C
import numpy as np
import pytest
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _scan(sa, cv, fn, cx, cy, fg_w, fg_h, fg_x, col, phase, weight, free=None):
    n, _ = sa.caprot_steps(fg_w, fg_h, cv.dpi)
    d_ang = sa.DeviceArray((len(cx),), np.float64)
    d_dv = sa.DeviceArray((len(cx), n), np.float64)
    # poison the outputs: a call that wrote nothing must not pass
    sa.lib().smr_hip_memcpy_h2d(d_ang.ptr, np.full(len(cx), -7.0).ctypes.data, 8 * len(cx))
    d_cx, d_cy = sa.DeviceArray.from_numpy(cx), sa.DeviceArray.from_numpy(cy)
    sa.sync()
    if free is not None:
        import torch
        before = torch.cuda.mem_get_info(0)[0]
    fn(d_cx, d_cy, len(cx), fg_w, fg_h, fg_x, col, phase, weight, d_ang, d_dv)
    sa.sync()
    if free is not None:
        free.append((before, torch.cuda.mem_get_info(0)[0]))
    return d_ang.numpy(), d_dv.numpy()


def test_noalloc_equals_caprot_on_two_calls_in_a_row(hip):
    sa = hip
    W, H = 420, 300
    st = sa.DrawStyle(1, 1, 0x4080c0, 0x000000, 0.0, 25.4 / 300, 0, 0, 0, 0, 0, 0)
    x, y, off, ref = po.synth_ways(int(W * H / 900), W, H, seed=W + H)
    cv = sa.Canvas(W, H, 300, 0xffffff)
    cv.draw(st, sa.DeviceArray.from_numpy(x), sa.DeviceArray.from_numpy(y), sa.DeviceArray.from_numpy(off),
            sa.DeviceArray.from_numpy(ref), len(off) - 1, len(ref))
    rng = np.random.default_rng(11)
    calls = [
        (64, rng.uniform(-5, W + 5, 64), rng.uniform(-5, H + 5, 64), 90, 14, 45, 0x000000, 0.0, 1.0),
        (17, rng.uniform(-5, W + 5, 17), rng.uniform(-5, H + 5, 17), 48, 10, 24, 0x203040, 90.0, -0.5),      # smaller: 17 captions, 13 angles
    ]
    want = [_scan(sa, cv, cv.caprot, cx, cy, *rest) for _, cx, cy, *rest in calls]
    free = []
    got = [_scan(sa, cv, cv.caprot_noalloc, cx, cy, *rest, free=free) for _, cx, cy, *rest in calls]
    # "allocates nothing": the device's free memory (hipMemGetInfo) is the same before and after each call
    assert len(free) == 2 and all(a == b for a, b in free), free
    assert sa.caprot_steps(48, 10, 300)[0] < sa.caprot_steps(90, 14, 300)[0]
    for (wa, wd), (ga, gd) in zip(want, got):
        assert np.array_equal(ga, wa)
        assert np.array_equal(gd, wd)
        assert len(np.unique(ga)) > 3
    cv.close()


def test_noalloc_rejects_more_angles_than_the_static_tables_hold(hip):
    sa = hip
    cv = sa.Canvas(64, 64, 9000, 0xffffff)           # 120 x 8 px at 9000 dpi is a tiny plane in mm ...
    d = sa.DeviceArray((2,), np.float64)
    cv.caprot_noalloc(d, d, 0, 120, 8, 0, 0, 0.0, 1.0, d)      # ... and an empty batch is fine
    cv.close()
    cv = sa.Canvas(64, 64, 10, 0xffffff)             # at 10 dpi the same plane is 305 mm wide: 958 angles fit, 200 px do not
    with pytest.raises(sa.SmrHipError) as e:
        cv.caprot_noalloc(d, d, 1, 200, 8, 0, 0, 0.0, 1.0, d)
    assert e.value.code == -6 and "static tables" in str(e.value)
    cv.close()
