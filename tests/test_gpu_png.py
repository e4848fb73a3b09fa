"""PNG files and tiles written from the device canvas (smr_hip_canvas_write_png / _write_tiles): the
decoded pixels must be those of libcairo's PNG writer for the same ARGB32 data -- checked against the real writer
(oracle/cairo_drive.c: cd_write_png_data, what save_main_image src/smrules_cairo.c:409 and save_image :467 call) and
against its restatement oracle.png_pixels, which tests/test_png_oracle.py pins against the same writer."""
import os
import numpy as np
import pytest
from oracle import pyoracle as po
from golden_io import draw_style

pytestmark = pytest.mark.gpu


def scene(sa, W, H, bg, n_ways=120, seed=5, translucent=False):
    cv = sa.Canvas(W, H, 300, bg)
    D = sa.DeviceArray.from_numpy
    for k, st in enumerate([po.Style(1, 1, 0x304080c0 if translucent else 0x4080c0, 0x000000, 0, 0.15, 0),
                            po.Style(0, 1, 0, 0x50a02030 if translucent else 0xa02030, 0, 0.4, 0)]):
        x, y, off, ref = po.synth_ways(n_ways, W, H, seed=seed + k)
        cv.draw(draw_style(sa, st), D(x), D(y), D(off.astype(np.int64)), D(ref.astype(np.int32)), len(off) - 1, len(ref))
    return cv


@pytest.mark.parametrize("bg,translucent", [(0xfff8e0, False), (0x60f0e0d0, True), (0xff000000, False)])
def test_png_decodes_to_libcairos_pixels(hip, tmp_path, bg, translucent):
    sa = hip
    W, H = 701, 517                                  # a width that leaves a ragged last group of four pixels
    cv = scene(sa, W, H, bg, translucent=translucent)
    img = cv.read()
    f = tmp_path / "gpu.png"
    st = cv.write_png(f)
    got, info = po.png_decode(f, with_info=True)
    want = po.png_pixels(img)
    assert got.shape == want.shape and got.shape[2] == (3 if st.n_opaque else 4)
    assert bool(st.n_opaque) == bool(np.all(img >> 24 == 255))
    assert np.array_equal(got, want), int((got != want).sum())
    assert st.bytes_written == os.path.getsize(f) and st.bytes_filtered == H * (1 + W * got.shape[2])
    assert (info["filters"] > 0).sum() >= 2          # the filter choice is alive (flat rows and busy rows differ)
    L = po.cairo()
    if L is not None:                                # the real writer on the same pixels
        g = tmp_path / "cairo.png"
        assert L.cd_write_png_data(po.P(img), W, H, str(g).encode()) == 0
        assert np.array_equal(got, po.png_decode(g))
        assert st.bytes_written < 1.6 * os.path.getsize(g)          # and the file is not grossly larger than libpng's
    # a sub-rectangle, one thread, fastest level
    cv.write_png(f, rect=(33, 41, 257, 130), level=1, threads=1)
    assert np.array_equal(po.png_decode(f), po.png_pixels(img[41:171, 33:290]))
    cv.close()


def test_png_in_several_strips_and_chunks(hip, tmp_path):
    """8192 px wide: 64 MB of filtered rows are 2047 image rows, so 2300 rows leave the device in two strips, each
    deflated as up to 16 independent chunks; the file is still ONE valid zlib stream (Adler-32 combined)"""
    sa = hip
    W, H = 8192, 2300
    cv = scene(sa, W, H, 0xfff8e0, n_ways=4000)
    img = cv.read()
    f = tmp_path / "big.png"
    st = cv.write_png(f, threads=16)
    got, info = po.png_decode(f, with_info=True)
    assert np.array_equal(got, po.png_pixels(img))
    assert info["idat_chunks"] >= 18 and st.n_opaque == 1
    cv.close()


@pytest.mark.parametrize("level", [-1, 100])          # zlib on the host; deflate on the device (SMR_PNG_DEVICE_DEFLATE)
def test_tiles_are_the_canvas_cut_on_the_tile_grid(hip, tmp_path, level):
    sa = hip
    W, H = 1100, 900
    cv = scene(sa, W, H, 0x60f0e0d0, n_ways=60, translucent=True)
    # make some tiles fully opaque: an opaque image painted over part of the canvas
    cv.blit([(0, 300, np.full((320, 600), 0xff206040, np.uint32))])
    img = cv.read()
    d = tmp_path / "tiles" / "12"
    os.makedirs(tmp_path / "tiles")
    st = cv.write_tiles(d, 40, 60, 4, 3, tile_px=256, tx0=2200, ty0=1400, threads=3, level=level)
    assert st.n_files == 12
    n_rgb = 0
    for j in range(3):
        for i in range(4):
            f = d / str(2200 + i) / f"{1400 + j}.png"
            want = po.png_pixels(img[60 + 256 * j:60 + 256 * (j + 1), 40 + 256 * i:40 + 256 * (i + 1)])
            got = po.png_decode(f)
            assert got.shape == want.shape, (i, j)
            assert np.array_equal(got, want), (i, j)
            n_rgb += got.shape[2] == 3
    assert n_rgb == st.n_opaque and 0 < n_rgb < 12            # per-tile colour type, as cairo decides per image
    assert st.bytes_written == sum(os.path.getsize(d / str(2200 + i) / f"{1400 + j}.png") for i in range(4) for j in range(3))
    cv.close()


@pytest.mark.parametrize("level", [-1, 100])
def test_tiles_in_several_batches(hip, tmp_path, level):
    """more tiles than one 64 MB staging buffer holds (255 tiles of 256 px): 18 x 16 = 288 tiles go in two batches"""
    sa = hip
    W, H = 18 * 256, 16 * 256
    cv = scene(sa, W, H, 0xfff8e0, n_ways=3000)
    img = cv.read()
    st = cv.write_tiles(tmp_path / "z", 0, 0, 18, 16, level=level)
    assert st.n_files == 288
    rng = np.random.default_rng(1)
    for i, j in [(0, 0), (17, 15), (14, 14), (3, 15)] + [tuple(rng.integers(0, 16, 2)) for _ in range(12)]:
        got = po.png_decode(tmp_path / "z" / str(i) / f"{j}.png")
        assert np.array_equal(got, po.png_pixels(img[256 * j:256 * (j + 1), 256 * i:256 * (i + 1)])), (i, j)
    cv.close()


def test_banded_canvas_writes_its_own_rows(hip, tmp_path):
    sa = hip
    W, H = 300, 400
    full = scene(sa, W, H, 0xfff8e0)
    img = full.read(); full.close()
    cv = sa.Canvas(W, H, 300, 0xfff8e0, band_y0=128, band_h=150)
    D = sa.DeviceArray.from_numpy
    for k, st in enumerate([po.Style(1, 1, 0x4080c0, 0x000000, 0, 0.15, 0), po.Style(0, 1, 0, 0xa02030, 0, 0.4, 0)]):
        x, y, off, ref = po.synth_ways(120, W, H, seed=5 + k)
        cv.draw(draw_style(sa, st), D(x), D(y), D(off.astype(np.int64)), D(ref.astype(np.int32)), len(off) - 1, len(ref))
    for level in (-1, 100):
        cv.write_png(tmp_path / "band.png", level=level)
        assert np.array_equal(po.png_decode(tmp_path / "band.png"), po.png_pixels(img[128:278]))
    # tiles of an odd size (a row is not a whole number of segments, a tile not a whole number of workgroups)
    cv.write_tiles(tmp_path / "t", 7, 3, 3, 1, tile_px=97, level=100)
    for i in range(3):
        assert np.array_equal(po.png_decode(tmp_path / "t" / str(i) / "0.png"), po.png_pixels(img[131:228, 7 + 97 * i:7 + 97 * (i + 1)]))
    cv.close()


def test_bad_requests_are_refused(hip, tmp_path):
    sa = hip
    cv = sa.Canvas(300, 200, 300, 0xffffff)
    for rect in [(0, 0, 301, 10), (-1, 0, 10, 10), (0, 190, 10, 11), (0, 0, 0, 10)]:
        with pytest.raises(sa.SmrHipError):
            cv.write_png(tmp_path / "x.png", rect=rect)
    with pytest.raises(sa.SmrHipError):
        cv.write_png(tmp_path / "x.png", level=10)
    with pytest.raises(sa.SmrHipError):
        cv.write_png(tmp_path / "no_such_dir" / "x.png")
    with pytest.raises(sa.SmrHipError):
        cv.write_tiles(tmp_path / "t", 0, 0, 2, 1, tile_px=256)          # 512 px do not fit into 300
    assert not os.path.exists(tmp_path / "x.png")
    cv.close()


DEVICE = 100          # SMR_PNG_DEVICE_DEFLATE


@pytest.mark.parametrize("bg,translucent", [(0xfff8e0, False), (0x60f0e0d0, True)])
def test_png_deflated_on_the_device(hip, tmp_path, bg, translucent):
    """level SMR_PNG_DEVICE_DEFLATE: the zlib stream comes from the device (one dynamic-Huffman block per strip over literals and
    runs); zlib itself inflates it here (its Adler-32 check included) and the pixels are libcairo's"""
    sa = hip
    W, H = 701, 517
    cv = scene(sa, W, H, bg, translucent=translucent)
    # flat stretches (runs of 258 and more), and noise (every byte value, codes of many lengths)
    rng = np.random.default_rng(3)
    noise = rng.integers(0, 1 << 32, (90, 300), dtype=np.uint64).astype(np.uint32)
    noise = (noise | 0xff000000) if not translucent else noise
    if translucent:            # premultiplied: no colour above its alpha
        a = noise >> 24
        ch = [np.minimum((noise >> s) & 255, a) for s in (16, 8, 0)]
        noise = (a << 24 | ch[0] << 16 | ch[1] << 8 | ch[2]).astype(np.uint32)
    cv.blit([(350, 20, noise), (10, 400, np.full((100, 640), 0xff336699, np.uint32))])
    img = cv.read()
    f = tmp_path / "dev.png"
    st = cv.write_png(f, level=DEVICE)
    got = po.png_decode(f)
    assert np.array_equal(got, po.png_pixels(img)), int((got != po.png_pixels(img)).sum())
    assert st.bytes_written == os.path.getsize(f) and st.bytes_filtered == H * (1 + W * got.shape[2])
    g = tmp_path / "zlib.png"
    stz = cv.write_png(g)
    assert st.bytes_written < 3 * stz.bytes_written
    # a sub-rectangle; a single row; a single pixel
    for rect in ((33, 41, 257, 130), (0, 77, 701, 1), (5, 5, 1, 1)):
        cv.write_png(f, rect=rect, level=DEVICE)
        x, y, w, h = rect
        assert np.array_equal(po.png_decode(f), po.png_pixels(img[y:y + h, x:x + w])), rect
    cv.close()


def test_png_deflated_on_the_device_in_several_strips(hip, tmp_path):
    """two strips: two deflate blocks with an empty stored block between them, ONE zlib stream whose Adler-32 is the strips' combined;
    and a blank canvas (nothing but runs)"""
    sa = hip
    W, H = 8192, 2300
    cv = scene(sa, W, H, 0xfff8e0, n_ways=4000)
    img = cv.read()
    f = tmp_path / "big.png"
    st = cv.write_png(f, level=DEVICE, threads=16)
    got, info = po.png_decode(f, with_info=True)
    assert np.array_equal(got, po.png_pixels(img))
    assert st.n_opaque == 1 and info["idat_chunks"] >= 4
    cv.clear()
    st = cv.write_png(f, level=DEVICE)
    assert np.array_equal(po.png_decode(f), po.png_pixels(cv.read()))
    assert st.bytes_written < st.bytes_filtered / 60          # a run of 258 bytes is a symbol or two
    cv.close()
    with pytest.raises(sa.SmrHipError):
        sa.Canvas(64, 64, 300, 0).write_png(f, level=11)
