"""Kernel 3 (caption auto-rotation scan, one wavefront per caption) through the C-ABI vs the
oracle restatement of find_angle (src/smrules_cairo.c:1959-2019) run on the same canvas pixels.
Chosen angles must be bit-exact; the raw per-angle differences agree to 1e-12 (the kernel sums
the exact integer |2125 dR + 7154 dG + 721 dB|, the reference adds doubles sequentially)."""
import ctypes as C
import numpy as np
import pytest
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def oracle_angles(img, cx, cy, fg_w, fg_h, fg_x, col, dpi, phase, weight):
    L = po.orc()
    r = C.c_int()
    n = L.orc_num_steps(fg_w, fg_h, dpi, C.byref(r))
    r = r.value
    fg = np.zeros((fg_h, fg_w), np.uint32)
    L.orc_make_plane(po.P(fg), fg_w, fg_h, fg_x, col)
    H, W = img.shape
    angles, dvs = [], []
    for x, y in zip(cx, cy):
        cround = lambda v: int(np.sign(v) * np.floor(abs(v) + 0.5))      # C round(): half away from zero
        ox, oy = cround(x - r / 2.0), cround(y - r / 2.0)
        cut = np.zeros((r, r), np.uint32)
        xs0, ys0 = max(ox, 0), max(oy, 0)
        xs1, ys1 = min(ox + r, W), min(oy + r, H)
        if xs1 > xs0 and ys1 > ys0:
            cut[ys0 - oy:ys1 - oy, xs0 - ox:xs1 - ox] = img[ys0:ys1, xs0:xs1]
        dv = np.zeros(n)
        angles.append(L.orc_find_angle(po.P(cut), r, po.P(fg), fg_w, fg_h, n, phase, weight, po.P(dv)))
        dvs.append(dv)
    return np.array(angles), np.array(dvs), n, r


def make_canvas(sa, W, H, bg, style):
    arrs = po.synth_ways(int(W * H / 900), W, H, seed=W + H)
    cv = sa.Canvas(W, H, 300, bg)
    x, y, off, ref = arrs
    cv.draw(style, sa.DeviceArray.from_numpy(x), sa.DeviceArray.from_numpy(y), sa.DeviceArray.from_numpy(off),
            sa.DeviceArray.from_numpy(ref), len(off) - 1, len(ref))
    return cv


@pytest.mark.parametrize("fg_w,fg_h,fg_x,col,phase,weight,bg", [
    (90, 14, 45, 0x000000, 0.0, 1.0, 0xffffff),      # E-aligned caption: left half of the plane is empty
    (64, 20, 0, 0x203040, 0.0, 0.7, 0xffffff),       # centred, weighted ellipse (test/rules/0180.osm: weight=0.7)
    (48, 10, 24, 0xffffff, 90.0, -0.5, 0xc0d0e0),    # negative weight inverts the metric (:1817-1818)
    (40, 12, 20, 0x000000, 0.0, 1.0, 0x10ffffff),    # background alpha byte 0xdf: (A & 0x7f) = 95 <= 101 -> "transparent" by the 0x7f-mask quirk (:1692)
    (270, 20, 135, 0x000000, 0.0, 1.0, 0xffffff),    # BASELINE.json C4 at its own shape: 270x20 plane, 271-px cut-out, 72 angles (45 blocks, 4 wavefronts)
    (320, 44, 0, 0x102030, 0.0, 0.7, 0xffffff),      # 220 blocks of 16x4 px: 7 wavefronts per caption, 32 blocks each
    (2100, 1, 0, 0x000000, 0.0, 1.0, 0xffffff),      # one line of 132 blocks: 5 wavefronts, the last blocks ragged
])
def test_angles_match_oracle(hip, fg_w, fg_h, fg_x, col, phase, weight, bg):
    sa = hip
    W, H = 420, 300
    st = sa.DrawStyle(1, 1, 0x4080c0, 0x000000, 0.0, 25.4 / 300, 0, 0, 0, 0, 0, 0)
    cv = make_canvas(sa, W, H, bg, st)
    img = cv.read()
    rng = np.random.default_rng(fg_w)
    n_cap = 48 if fg_w < 300 else 10
    cx = rng.uniform(-5, W + 5, n_cap)
    cy = rng.uniform(-5, H + 5, n_cap)
    cx[:4] = [0.0, W, W / 2 + 0.5, 17.25]
    cy[:4] = [0.0, H, H / 2 + 0.5, 290.75]
    want, want_dv, n, r = oracle_angles(img, cx, cy, fg_w, fg_h, fg_x, col, 300, phase, weight)
    assert sa.caprot_steps(fg_w, fg_h, 300) == (n, r)
    assert (fg_w, fg_h) != (270, 20) or (n, r) == (72, 271)
    d_ang = sa.DeviceArray((n_cap,), np.float64)
    d_dv = sa.DeviceArray((n_cap, n), np.float64)
    cv.caprot(sa.DeviceArray.from_numpy(cx), sa.DeviceArray.from_numpy(cy), n_cap, fg_w, fg_h, fg_x, col, phase, weight,
              d_ang, d_dv)
    sa.sync()
    got, got_dv = d_ang.numpy(), d_dv.numpy()
    assert np.max(np.abs(got_dv - want_dv)) < 1e-12
    assert np.array_equal(got, want), np.argwhere(got != want).ravel().tolist()
    assert len(np.unique(got)) > 3          # the scan really discriminates between positions
    cv.close()


def test_bw_stretch_integer_form_equals_double_chain():
    """the kernel's integer bw-stretch vs the reference's rotate/scale/rotate chain (src/smrules_cairo.c:1598-1619)"""
    rng = np.random.default_rng(0)
    cols = np.concatenate([rng.integers(0, 1 << 24, 200000), np.array([0, 0xffffff, 0xff0000, 0x00ff00, 0x0000ff, 0x808080])])
    L = po.orc()
    for c in cols[:20000].tolist() + cols[-6:].tolist():
        R, G, B = (c >> 16) & 255, (c >> 8) & 255, c & 255
        S = R + G + B
        want = L.orc_bw_stretch(c) & 0xffffff
        got = (((2 * (S + 12 * R) + 15) // 30) << 16) | (((2 * (S + 12 * G) + 15) // 30) << 8) | ((2 * (S + 12 * B) + 15) // 30)
        assert got == want, hex(c)


def test_caprot_rejects_oversized_plane(hip):
    sa = hip
    cv = sa.Canvas(64, 64, 300, 0xffffff)
    d = sa.DeviceArray((2,), np.float64)
    with pytest.raises(sa.SmrHipError) as e:
        cv.caprot(d, d, 1, 400, 80, 0, 0, 0.0, 1.0, d)
    assert e.value.code == -6
    cv.close()


def test_bands_with_halo_match_full_canvas(hip):
    """multi-GPU layout: a rank scans the captions whose centre lies in its band and reads the
    neighbours' rows from the halo strips; the angles equal the single-canvas ones bit for bit"""
    sa = hip
    W, H = 300, 240
    st = sa.DrawStyle(1, 1, 0x4080c0, 0x000000, 0.0, 25.4 / 300, 0, 0, 0, 0, 0, 0)
    arrs = po.synth_ways(int(W * H / 900), W, H, seed=7)
    x, y, off, ref = arrs
    dev = [sa.DeviceArray.from_numpy(a) for a in (x, y, off, ref)]
    full = sa.Canvas(W, H, 300, 0xffffff)
    full.draw(st, dev[0], dev[1], dev[2], dev[3], len(off) - 1, len(ref))
    img = full.read()
    rng = np.random.default_rng(3)
    n_cap = 60
    cx, cy = rng.uniform(0, W, n_cap), rng.uniform(0, H - 0.01, n_cap)
    fg_w, fg_h, fg_x = 70, 12, 35
    n, r = sa.caprot_steps(fg_w, fg_h, 300)
    d_all = sa.DeviceArray((n_cap,), np.float64)
    full.caprot(sa.DeviceArray.from_numpy(cx), sa.DeviceArray.from_numpy(cy), n_cap, fg_w, fg_h, fg_x, 0, 0.0, 1.0, d_all)
    sa.sync()
    want = d_all.numpy()
    from smrender_amd import dist
    halo = r // 2 + 2
    got = np.full(n_cap, np.nan)
    for y0, h in dist.bands(H, 3):
        cv = sa.Canvas(W, H, 300, 0xffffff, y0, h)
        cv.draw(st, dev[0], dev[1], dev[2], dev[3], len(off) - 1, len(ref))
        mine = np.nonzero((np.floor(cy) >= y0) & (np.floor(cy) < y0 + h))[0]
        d = sa.DeviceArray((max(len(mine), 1),), np.float64)
        if y0 > 0 or y0 + h < H:
            with pytest.raises(sa.SmrHipError):          # no halo handed over: refused, not guessed
                cv.caprot(sa.DeviceArray.from_numpy(cx[mine]), sa.DeviceArray.from_numpy(cy[mine]), len(mine), fg_w, fg_h, fg_x, 0, 0.0, 1.0, d)
        a0, b1 = max(0, y0 - halo), min(H, y0 + h + halo)
        cv.set_halo(img[a0:y0] if y0 > a0 else None, img[y0 + h:b1] if b1 > y0 + h else None)
        cv.caprot(sa.DeviceArray.from_numpy(cx[mine]), sa.DeviceArray.from_numpy(cy[mine]), len(mine), fg_w, fg_h, fg_x, 0, 0.0, 1.0, d)
        sa.sync()
        got[mine] = d.numpy()[:len(mine)]
        cv.close()
    full.close()
    assert np.array_equal(got, want)


def test_device_halo_and_extended_background_match_full_canvas(hip):
    """the device-to-device form of the halo hand-over (smr_hip_canvas_set_halo_device: what a send/recv over RCCL delivers) and
    the banded caption rule's private background (smr_hip_canvas_extend: band + halo rows as one canvas onto which the
    rectangles of finished captions are drawn, src/smrules_cairo.c:2376-2382): angles equal the single-canvas ones, before
    and after a rectangle that straddles the band boundary"""
    sa = hip
    W, H = 300, 240
    st = sa.DrawStyle(1, 1, 0x4080c0, 0x000000, 0.0, 25.4 / 300, 0, 0, 0, 0, 0, 0)
    x, y, off, ref = po.synth_ways(int(W * H / 900), W, H, seed=17)
    dev = [sa.DeviceArray.from_numpy(a) for a in (x, y, off, ref)]
    full = sa.Canvas(W, H, 300, 0xffffff)
    full.draw(st, dev[0], dev[1], dev[2], dev[3], len(off) - 1, len(ref))
    img = full.read()
    rng = np.random.default_rng(5)
    n_cap = 50
    cx, cy = rng.uniform(0, W, n_cap), rng.uniform(0, H - 0.01, n_cap)
    fg_w, fg_h, fg_x = 70, 12, 35
    n, r = sa.caprot_steps(fg_w, fg_h, 300)
    halo = r // 2 + 2
    from smrender_amd import dist
    bl = dist.bands(H, 2)
    # a rectangle across the boundary between the two bands, drawn without a group layer like auto_ctx's fills
    by = bl[1][0]
    rx = np.array([100.0, 180.0, 180.0, 100.0]); ry = np.array([by - 6.0, by - 6.0, by + 7.0, by + 7.0])
    roff = np.array([0, 5], np.int64); rref = np.array([0, 1, 2, 3, 0], np.int32)
    rst = sa.DrawStyle(1, 0, 0x203040, 0, 0.0, 0.0, 0, 0, 0, 0, 0, 1)
    rdev = [sa.DeviceArray.from_numpy(a) for a in (rx, ry, roff, rref)]

    def scan(cv, idx):
        d = sa.DeviceArray((max(len(idx), 1),), np.float64)
        cv.caprot(sa.DeviceArray.from_numpy(cx[idx]), sa.DeviceArray.from_numpy(cy[idx]), len(idx), fg_w, fg_h, fg_x, 0, 0.0, 1.0, d)
        sa.sync()
        return d.numpy()[:len(idx)]

    everyone = np.arange(n_cap)
    want0 = scan(full, everyone)
    bg = full.clone()
    bg.draw(rst, rdev[0], rdev[1], rdev[2], rdev[3], 1, 5)
    want1 = scan(bg, everyone)
    assert (want0 != want1).sum() >= 1          # the rectangle is seen by somebody
    got0, got1 = np.full(n_cap, np.nan), np.full(n_cap, np.nan)
    for y0, h in bl:
        cv = sa.Canvas(W, H, 300, 0xffffff, y0, h)
        cv.draw(st, dev[0], dev[1], dev[2], dev[3], len(off) - 1, len(ref))
        a0, b1 = max(0, y0 - halo), min(H, y0 + h + halo)
        d_above = sa.DeviceArray.from_numpy(np.ascontiguousarray(img[a0:y0])) if y0 > a0 else None
        d_below = sa.DeviceArray.from_numpy(np.ascontiguousarray(img[y0 + h:b1])) if b1 > y0 + h else None
        cv.set_halo_device(d_above.ptr if d_above else None, y0 - a0, d_below.ptr if d_below else None, b1 - y0 - h)
        mine = np.nonzero((np.floor(cy) >= y0) & (np.floor(cy) < y0 + h))[0]
        got0[mine] = scan(cv, mine)
        # the strips a rank sends are views of its band's device memory (no copy): what the RCCL send reads
        view = cv.device_rows(h - 5, 5)
        assert view.is_cuda and np.array_equal(view.cpu().numpy().view(np.uint32), img[y0 + h - 5:y0 + h])
        ext = cv.extend()
        assert (ext.band_y0, ext.band_h) == (a0, b1 - a0)
        assert np.array_equal(ext.read(), img[a0:b1])
        ext.draw(rst, rdev[0], rdev[1], rdev[2], rdev[3], 1, 5)
        got1[mine] = scan(ext, mine)
        ext.close(); cv.close()
    full.close(); bg.close()
    assert np.array_equal(got0, want0)
    assert np.array_equal(got1, want1)
