"""The KAP checker itself (CPU): oracle/kap.c's colour reduction against the reference's own src/median_cut.c compiled here
(oracle/_ref/libref_median.so), its nearest-colour search likewise, and the BSB row coder through its decoder.  These carry
tests/test_gpu_kap.py."""
import ctypes as C
import numpy as np
import pytest
from oracle import pyoracle as po


def chart_like(rng, h, w):
    """few flat colours, anti-aliased edges between them, a gradient: what a rendered chart holds"""
    base = np.array([0xfff8e0, 0x4080c0, 0x204060, 0x80c080, 0x000000, 0xe0e0e0], np.uint32)
    img = base[rng.integers(0, len(base), (h // 16 + 1, w // 16 + 1))].repeat(16, 0).repeat(16, 1)[:h, :w].copy()
    # blends along tile borders
    for sh in (1, 2, 3):
        a = img[:, :-sh].astype(np.uint64); b = img[:, sh:].astype(np.uint64)
        mix = sum(((((a >> s) & 0xff) * (4 - sh) + ((b >> s) & 0xff) * sh) // 4) << s for s in (0, 8, 16)).astype(np.uint32)
        m = rng.random((h, w - sh)) < 0.15
        img[:, :-sh][m] = mix[m]
    img[h // 3:h // 3 + 8, :] = (np.arange(w, dtype=np.uint32) * 255 // w) * 0x010101
    return img | 0xff000000


@pytest.mark.parametrize("case", ["chart", "noise", "two_colours", "one_colour", "ties"])
def test_median_cut_equals_the_references_own(case):
    ref = po.ref_median()
    if ref is None:
        pytest.skip("oracle/_ref/libref_median.so not built (no reference here)")
    rng = np.random.default_rng(3)
    if case == "chart":
        img = chart_like(rng, 120, 200)
    elif case == "noise":
        img = rng.integers(0, 1 << 24, (90, 110)).astype(np.uint32)
    elif case == "two_colours":
        img = np.where(rng.random((40, 50)) < 0.3, 0x102030, 0xf0e0d0).astype(np.uint32)
    elif case == "one_colour":
        img = np.full((10, 13), 0x808080, np.uint32)
    else:       # many points that tie along the cut dimension: the order of equals decides the boxes
        img = (rng.integers(0, 4, (64, 64)) * 0x400000 + rng.integers(0, 256, (64, 64)) * 0x100 + rng.integers(0, 3, (64, 64)) * 0x7f).astype(np.uint32)
    pts = po.argb_points(img)
    for desired in (127, 5, 1):
        a0, l0, p0 = po.median_cut(pts, desired)
        a1, l1, p1 = po.median_cut(pts, desired, ref=ref)
        assert len(a0) == len(a1) and np.array_equal(l0, l1) and np.array_equal(a0, a1), (case, desired)
        assert np.array_equal(p0, p1), "the points must end up in the same order"
    # nearest colour, the first of equals
    ref.ref_nearest_block.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    avg = np.ascontiguousarray(a0 if len(a0) > 1 else np.array([[1, 2, 3], [1, 2, 3], [9, 9, 9]], np.uint8))
    for p in np.concatenate([pts[:200], avg[:3]]):
        p = np.ascontiguousarray(p)
        assert po._kap().orc_nearest(po.P(avg), len(avg), po.P(p)) == ref.ref_nearest_block(po.P(avg), len(avg), po.P(p))


def test_bsb_rows_decode_to_what_went_in():
    rng = np.random.default_rng(4)
    for depth in range(1, 8):
        ncol = min(127, (1 << depth) - 1) if depth > 1 else 1
        for w, line in ((1, 0), (7, 1), (300, 127), (300, 128), (5000, 16383), (70, 16384), (65535, 65534)):
            runs = rng.integers(1, 400 if w > 1000 else 9, w)
            idx = np.repeat(rng.integers(1, ncol + 1, w), runs)[:w].astype(np.uint8)
            if w == 65535:
                idx[:] = idx[0]                   # one run over the whole row: the longest count there is
            row = po.bsb_row(idx, depth, line)
            assert row[-1] == 0 and len(row) <= 2 * w + 8
            n, got_line, got = po.bsb_decode_row(row, w, depth)
            assert n == len(row) and got_line == line and np.array_equal(got, idx), (depth, w, line)
    # known bytes: line 0 is 0x80 0x00; depth 7 leaves no room for a count in the colour byte
    assert list(po.bsb_row(np.array([5, 5, 5], np.uint8), 3, 0)) == [0x80, 0x00, (5 << 4) | 2, 0]
    assert list(po.bsb_row(np.array([5, 5, 5], np.uint8), 7, 1)) == [0x01, 0x85, 0x02, 0]
    assert list(po.bsb_row(np.array([1] * 300, np.uint8), 3, 200)) == [0x81, 200 & 0x7f, 0x80 | (1 << 4) | (299 >> 7), 299 & 0x7f, 0]


def test_kap_body_layout():
    rng = np.random.default_rng(5)
    img = chart_like(rng, 48, 80)
    body, pal = po.kap_body(img, 1000)
    depth = max(1, int(np.ceil(np.log2(len(pal)))))
    head = f"OST/1\r\nIFM/{depth}\r\n".encode() + b"".join(
        f"RGB/{i + 1},{(c >> 16) & 255},{(c >> 8) & 255},{c & 255}\r\n".encode() for i, c in enumerate(pal)) + b"\x1a\x00" + bytes([depth])
    assert bytes(body[:len(head)]) == head
    offs = np.frombuffer(bytes(body[-4 * 49:]), ">i4")
    assert offs[0] == 1000 + len(head) and offs[-1] == 1000 + len(body) - 4 * 49 and np.all(np.diff(offs) > 0)
    # every row decodes, and its indices name the nearest palette colour
    pts = po.argb_points(img).reshape(48, 80, 3)
    palpts = po.argb_points(pal)
    for y in (0, 17, 47):
        a, b = offs[y] - 1000, offs[y + 1] - 1000
        n, line, idx = po.bsb_decode_row(body[a:b], 80, depth)
        assert n == b - a and line == y
        d = np.abs(pts[y][:, None, :].astype(int) - palpts[None, :, :].astype(int)).sum(-1)
        assert np.array_equal(d[np.arange(80), idx - 1], d.min(1))
