"""CPU tests of image-texture loading (rtw_stb_image.h:23-121): baseline JPEG, PNG, PPM.

The texels are inputs of the sample loop, so the loader must hand the kernels the very bytes the reference's
rtw_image holds: stb_image's JPEG decode (its integer inverse DCT, its chroma filter, its fixed-point colour
conversion), then stbi_loadf's gamma-2.2 float mapping, then float_to_byte.  The golden arrays were dumped from
the reference's own loader (oracle/_ref `texels`, tests/golden/make_goldens.py) for small JPEGs written by Pillow;
where the reference tree exists its two real textures are compared live as well.
"""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from tests.conftest import GOLDEN

FIXTURES = ["jpg_444_40x24", "jpg_420_37x23", "jpg_420_1x1", "jpg_420_17x9_rst", "jpg_422_33x16", "jpg_grey_19x21", "jpg_420_64x64_noise",
            "jpg_prog_444_40x24", "jpg_prog_420_37x23", "jpg_prog_422_33x16", "jpg_prog_grey_19x21", "jpg_prog_420_100x75_noise"]


@pytest.mark.parametrize("name", FIXTURES)
def test_jpeg_texels_equal_the_reference_loaders_bytes(rt, name):
    got = rt.load_image_texels(os.path.join(GOLDEN, name + ".jpg"))
    want = np.load(os.path.join(GOLDEN, name + "_texels.npy"), allow_pickle=False)
    assert got is not None and got.shape == want.shape
    assert np.array_equal(got, want)


def test_flat_progressive_jpeg_above_256_pixels_per_byte_loads_like_the_reference(rt):
    """A uniform 1024x1024 progressive greyscale JPEG costs a few bits per 8x8 block (libjpeg's default progression: 163 pixels
    per byte of file; a DC-first scan without successive approximation gets to ~500, beyond what any baseline scan can reach).
    The loader's plausibility bound (allocations are sized from header fields: 1024 pixels per byte + slack) must let such
    files through -- the reference's stb_image loads them; the expected texels (one value everywhere) come from its loader."""
    path = os.path.join(GOLDEN, "flat_prog_grey_1024x1024.jpg")
    assert 1024 * 1024 / os.path.getsize(path) > 150
    got = rt.load_image_texels(path)
    want = np.load(os.path.join(GOLDEN, "flat_prog_grey_1024x1024_value.npy"), allow_pickle=False)
    assert got is not None and got.shape == (1024, 1024, 3)
    assert (got == want).all()
    # ... while a header that claims far more pixels than the file could hold is still refused before anything is allocated
    data = bytearray(open(path, "rb").read())
    at = data.index(b"\xff\xc2") + 5     # SOF2: height, width as 16-bit big-endian fields
    data[at:at + 4] = b"\xff\xff\xff\xff"
    with tempfile.TemporaryDirectory() as tmp:
        bad = os.path.join(tmp, "huge.jpg")
        open(bad, "wb").write(bytes(data))
        assert rt.load_image_texels(bad) is None


PNG_FIXTURES = ["png_rgb8_13x7", "png_rgba8_9x9_adam7", "png_grey8_17x5", "png_grey4_11x6", "png_grey2_10x4_adam7", "png_grey1_19x3",
                "png_greyalpha8_8x8", "png_grey16_6x5", "png_rgb16_7x4_adam7", "png_rgba16_5x5", "png_pal8_12x5", "png_pal4_9x7_adam7",
                "png_pal2_15x2", "png_pal1_21x3", "png_rgb8_1x1_stored", "png_rgb8_64x48_fixedhuff"]


@pytest.mark.parametrize("name", PNG_FIXTURES)
def test_png_texels_equal_the_reference_loaders_bytes(rt, name):
    """Every PNG colour type (grey, RGB, palette, grey+alpha, RGBA), bit depth (1, 2, 4, 8, 16) and both interlace
    methods, all five scanline filters, stored / fixed / dynamic deflate blocks, split IDAT, ancillary chunks."""
    got = rt.load_image_texels(os.path.join(GOLDEN, name + ".png"))
    want = np.load(os.path.join(GOLDEN, name + "_texels.npy"), allow_pickle=False)
    assert got is not None and got.shape == want.shape
    assert np.array_equal(got, want)


def test_damaged_png_files_fail_to_load(rt):
    data = open(os.path.join(GOLDEN, "png_rgb8_13x7.png"), "rb").read()
    with tempfile.TemporaryDirectory() as tmp:
        for cut in (7, 20, 60, len(data) - 30):
            p = os.path.join(tmp, f"cut{cut}.png")
            open(p, "wb").write(data[:cut])
            assert rt.load_image_texels(p) is None
        bad = bytearray(data)
        bad[25] = 7   # an invalid colour type in IHDR
        p = os.path.join(tmp, "badtype.png")
        open(p, "wb").write(bytes(bad))
        assert rt.load_image_texels(p) is None


def test_ppm_goes_through_the_same_byte_mapping(rt):
    got = rt.load_image_texels(os.path.join(GOLDEN, "earth_synth.ppm"))
    raw = open(os.path.join(GOLDEN, "earth_synth.ppm"), "rb").read()
    body = np.frombuffer(raw[raw.index(b"255\n") + 4:], np.uint8).reshape(32, 64, 3)
    lin = np.power(body.astype(np.float32) / np.float32(255.0), np.float32(2.2)).astype(np.float32)
    want = np.where(lin <= 0, 0, np.where(lin >= 1, 255, (256.0 * lin.astype(np.float64)).astype(np.int64))).astype(np.uint8)
    assert got.shape == (32, 64, 3)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1   # powf vs numpy's float32 power may differ in the last ulp
    assert (got == want).mean() > 0.99


def test_missing_and_damaged_jpeg_files_fail_to_load(rt):
    assert rt.load_image_texels("/nonexistent/file.jpg") is None
    data = open(os.path.join(GOLDEN, "jpg_420_37x23.jpg"), "rb").read()
    with tempfile.TemporaryDirectory() as tmp:
        for cut in (3, 20, 200):
            p = os.path.join(tmp, f"cut{cut}.jpg")
            open(p, "wb").write(data[:cut])
            assert rt.load_image_texels(p) is None
        p = os.path.join(tmp, "garbage.jpg")
        open(p, "wb").write(b"\xff\xd8" + bytes(range(256)) * 4)
        assert rt.load_image_texels(p) is None


def test_reference_textures_decode_identically_where_the_reference_exists(rt, orc):
    files = ["/root/reference/Images/earthmap.jpg", "/root/reference/male_texture.jpg", "/root/reference/Images/Sky.png",
             "/root/reference/Images/ImageOutputColors.png", "/root/reference/Images/final.png"]
    if not (os.path.exists(orc.REF_DRIVER) and all(os.path.exists(f) for f in files)):
        pytest.skip("needs /root/reference and oracle/_ref (build container only)")
    for f in files:
        with tempfile.TemporaryDirectory() as tmp:
            prefix = os.path.join(tmp, "t")
            subprocess.check_call([orc.REF_DRIVER, "texels", f, prefix])
            w, h = np.fromfile(prefix + ".dims", np.int32).tolist()
            want = np.fromfile(prefix + ".u8", np.uint8).reshape(h, w, 3)
        got = rt.load_image_texels(f)
        assert got is not None and np.array_equal(got, want), f
