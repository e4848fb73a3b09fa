"""The PNG checker itself (CPU): oracle.png_pixels, the restatement of what libcairo's PNG writer makes of an ARGB32 image,
and oracle.png_decode against files written by the REAL libcairo (oracle/cairo_drive.c: cd_write_png_data).  These two carry
tests/test_gpu_png.py."""
import numpy as np
import pytest
from oracle import pyoracle as po


def premultiplied(rng, h, w, opaque=False):
    a = np.full((h, w), 255, np.uint32) if opaque else rng.integers(0, 256, (h, w)).astype(np.uint32)
    if not opaque:
        a[rng.random((h, w)) < 0.2] = 255
        a[rng.random((h, w)) < 0.1] = 0
    c = [(rng.integers(0, 256, (h, w)).astype(np.uint32) * a + 127) // 255 for _ in range(3)]
    return (a << 24 | c[0] << 16 | c[1] << 8 | c[2]).astype(np.uint32)


@pytest.mark.parametrize("opaque", [True, False])
def test_restated_conversion_and_decoder_against_libcairo(tmp_path, opaque):
    L = po.cairo()
    if L is None:
        pytest.skip("libcairo not loadable")
    rng = np.random.default_rng(11 + opaque)
    img = premultiplied(rng, 97, 131, opaque)
    img[10:40, 20:90] = img[10, 20]                 # a flat area: libpng picks other filters there
    f = tmp_path / "cairo.png"
    assert L.cd_write_png_data(po.P(img), 131, 97, str(f).encode()) == 0
    got, info = po.png_decode(f, with_info=True)
    want = po.png_pixels(img)
    assert got.shape == want.shape == (97, 131, 3 if opaque else 4)
    assert np.array_equal(got, want)
    assert info["filters"].sum() == 97
