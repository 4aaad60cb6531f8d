#!/usr/bin/env python3
"""Regenerates tests/golden/ from the REFERENCE's own classes (oracle/_ref/ref_driver).

Runs only where /root/reference exists (the build container).  Outputs are data:
  earth_synth.ppm            synthetic 64x32 RGB8 texture used by the textured scenes
  kat_*.npy, kat_scene.rtks  function-level known-answer vectors (ref_kats.inc)
  desc_sha256.json           sha256 of the flattened rtk_scene_desc of every scene, dumped from
                             the reference's pointer graph (bvh topology, perlin tables, texels ...)
  img_<scene>.npz            seed-matched linear framebuffers (+ bytes, + RNG-draw/segment counts)
                             rendered through the reference's hittable/material/texture classes
The reference has no tests or fixtures of its own; these vectors are what pins
oracle/rt_oracle.cpp, and through it the device kernels.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.scene_cases import IMAGE_CASES, SCENE_SEED, RENDER_SEED, scene_file  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "ref_driver")


def main():
    if not os.path.exists(REF):
        raise SystemExit("oracle/_ref/ref_driver missing: run `make -C oracle ref` where /root/reference exists")
    w, h = 64, 32
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([(x * 4) % 256, (y * 8) % 256, ((x * y) * 3) % 256], -1).astype(np.uint8)
    earth = os.path.join(HERE, "earth_synth.ppm")
    with open(earth, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h) + img.tobytes())

    env = dict(os.environ, RTK_KAT_IMAGE=earth)
    subprocess.check_call([REF, "kat", HERE], env=env)

    tmp = tempfile.mkdtemp()
    hashes = {}
    for name, W, H, spp, depth in IMAGE_CASES:
        desc = os.path.join(tmp, name + ".rtks")
        arg = scene_file(name, HERE)
        subprocess.check_call([REF, "desc", name, str(SCENE_SEED), arg, desc], stderr=subprocess.DEVNULL)
        hashes[name] = hashlib.sha256(open(desc, "rb").read()).hexdigest()
        prefix = os.path.join(tmp, "img_" + name)
        subprocess.check_call([REF, "render", name, str(SCENE_SEED), arg, str(W), str(H), str(spp), str(depth), str(RENDER_SEED), "4", prefix],
                              stderr=subprocess.DEVNULL)
        meta = json.load(open(prefix + ".json"))
        np.savez_compressed(os.path.join(HERE, f"img_{name}.npz"),
                            linear=np.fromfile(prefix + ".f64").reshape(H, W, 3), rgb8=np.fromfile(prefix + ".u8", np.uint8).reshape(H, W, 3),
                            camera=np.fromfile(prefix + ".cam", np.uint8),
                            counts=np.array([meta["rng_draws"], meta["segments"], meta["surface_hits"]], np.int64),
                            shape=np.array([W, H, spp, depth], np.int64))
        print(name, hashes[name][:16], meta["rng_draws"])
    json.dump(hashes, open(os.path.join(HERE, "desc_sha256.json"), "w"), indent=1, sort_keys=True)




def make_jpeg_fixtures():
    """Small baseline and progressive JPEGs written by Pillow (synthetic content) + the bytes the REFERENCE's rtw_image
    holds for them (ref_driver texels): 4:4:4, 4:2:0, 4:2:2, grey, sizes that are not multiples of the MCU, restart markers."""
    from PIL import Image

    rng = np.random.default_rng(20250418)
    cases = {
        "jpg_444_40x24": dict(size=(40, 24), subsampling=0, quality=90),
        "jpg_420_37x23": dict(size=(37, 23), subsampling=2, quality=85),
        "jpg_420_1x1": dict(size=(1, 1), subsampling=2, quality=85),
        "jpg_420_17x9_rst": dict(size=(17, 9), subsampling=2, quality=70, restart_marker_blocks=1),
        "jpg_422_33x16": dict(size=(33, 16), subsampling=1, quality=95),
        "jpg_grey_19x21": dict(size=(19, 21), grey=True, quality=80),
        "jpg_420_64x64_noise": dict(size=(64, 64), subsampling=2, quality=60, noise=True),
        "jpg_prog_444_40x24": dict(size=(40, 24), subsampling=0, quality=90, progressive=True),
        "jpg_prog_420_37x23": dict(size=(37, 23), subsampling=2, quality=85, progressive=True),
        "jpg_prog_422_33x16": dict(size=(33, 16), subsampling=1, quality=95, progressive=True),
        "jpg_prog_grey_19x21": dict(size=(19, 21), grey=True, quality=80, progressive=True),
        "jpg_prog_420_100x75_noise": dict(size=(100, 75), subsampling=2, quality=30, noise=True, progressive=True),
    }
    for name, c in cases.items():
        w, h = c["size"]
        y, x = np.mgrid[0:h, 0:w]
        if c.get("noise"):
            img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        else:
            img = np.stack([(x * 7 + y * 3) % 256, (255 - x * 5 + y * 11) % 256, (x * y + 40 * np.sin(x / 3.0)) % 256], -1).astype(np.uint8)
        path = os.path.join(HERE, name + ".jpg")
        kw = {k: v for k, v in c.items() if k in ("quality", "subsampling", "restart_marker_blocks")}
        im = Image.fromarray(img[:, :, 0] if c.get("grey") else img, "L" if c.get("grey") else "RGB")
        im.save(path, "JPEG", optimize=False, progressive=bool(c.get("progressive")), **kw)
        prefix = os.path.join(tempfile.mkdtemp(), name)
        subprocess.check_call([REF, "texels", path, prefix])
        dims = np.fromfile(prefix + ".dims", np.int32)
        assert dims.tolist() == [w, h], (name, dims)
        np.save(os.path.join(HERE, name + "_texels.npy"), np.fromfile(prefix + ".u8", np.uint8).reshape(h, w, 3))
        print(name, os.path.getsize(path), "bytes")




def make_flat_progressive_jpeg():
    """A uniform 1024x1024 progressive greyscale JPEG: ~1 bit per 8x8 block (the DC-first scan; the AC scans collapse into
    end-of-band runs), i.e. more than 256 pixels per byte of file -- the file the loader's plausibility bound must not
    reject (the reference's stb_image loads it).  The reference's texels are all one value: only that value is stored."""
    from PIL import Image

    path = os.path.join(HERE, "flat_prog_grey_1024x1024.jpg")
    Image.fromarray(np.full((1024, 1024), 137, np.uint8), "L").save(path, "JPEG", optimize=True, progressive=True, quality=75)
    prefix = os.path.join(tempfile.mkdtemp(), "flat")
    subprocess.check_call([REF, "texels", path, prefix])
    dims = np.fromfile(prefix + ".dims", np.int32)
    assert dims.tolist() == [1024, 1024], dims
    texels = np.fromfile(prefix + ".u8", np.uint8).reshape(1024, 1024, 3)
    assert (texels == texels[0, 0]).all()
    np.save(os.path.join(HERE, "flat_prog_grey_1024x1024_value.npy"), texels[0, 0].copy())
    print("flat progressive:", os.path.getsize(path), "bytes,", 1024 * 1024 / os.path.getsize(path), "pixels per byte, value", texels[0, 0])


def make_png_fixtures():
    """Small PNGs of every colour type / bit depth / interlace combination, hand-assembled (zlib + scanline filters of
    all five types) so that nothing depends on an imaging library's defaults, + the bytes the REFERENCE's rtw_image
    holds for them (ref_driver texels)."""
    import struct
    import zlib

    rng = np.random.default_rng(777)

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)

    def filter_row(ftype, row, prev, bpp):
        out = bytearray(len(row))
        for i in range(len(row)):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ftype == 0: p = 0
            elif ftype == 1: p = a
            elif ftype == 2: p = b
            elif ftype == 3: p = (a + b) >> 1
            else:
                pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[i] = (row[i] - p) & 0xFF
        return bytes(out)

    def pack_rows(samples, w, h, channels, depth):
        """samples[h][w][channels] ints < 2^depth -> list of packed scanlines."""
        rows = []
        for j in range(h):
            if depth == 8:
                rows.append(bytes(int(v) for px in samples[j] for v in px))
            elif depth == 16:
                rows.append(b"".join(struct.pack(">H", int(v)) for px in samples[j] for v in px))
            else:
                bits = "".join(format(int(v), "0%db" % depth) for px in samples[j] for v in px)
                bits += "0" * (-len(bits) % 8)
                rows.append(bytes(int(bits[k:k + 8], 2) for k in range(0, len(bits), 8)))
        return rows

    def encode(name, w, h, colour, depth, interlace, palette=None, trns=None, split_idat=False, level=6):
        channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[colour]
        samples = rng.integers(0, 1 << depth, (h, w, channels))
        if colour == 3:
            samples = rng.integers(0, len(palette) // 3, (h, w, 1))
        bpp = max(1, channels * depth // 8)
        raw = bytearray()
        passes = [(0, 0, 1, 1)] if not interlace else [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
        f = 0
        for (x0, y0, dx, dy) in passes:
            sub = samples[y0::dy, x0::dx]
            if sub.shape[0] == 0 or sub.shape[1] == 0:
                continue
            rows = pack_rows(sub, sub.shape[1], sub.shape[0], channels, depth)
            prev = bytes(len(rows[0]))
            for row in rows:
                ftype = f % 5
                f += 1
                raw.append(ftype)
                raw += filter_row(ftype, row, prev, bpp)
                prev = row
        z = zlib.compress(bytes(raw), level)
        body = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, colour, 0, 0, interlace))
        body += chunk(b"gAMA", struct.pack(">I", 45455))
        if palette is not None:
            body += chunk(b"PLTE", bytes(palette))
        if trns is not None:
            body += chunk(b"tRNS", bytes(trns))
        if split_idat:
            third = max(1, len(z) // 3)
            for k in range(0, len(z), third):
                body += chunk(b"IDAT", z[k:k + third])
        else:
            body += chunk(b"IDAT", z)
        body += chunk(b"tEXt", b"Comment\x00synthetic fixture") + chunk(b"IEND", b"")
        path = os.path.join(HERE, name + ".png")
        open(path, "wb").write(body)
        prefix = os.path.join(tempfile.mkdtemp(), name)
        subprocess.check_call([REF, "texels", path, prefix])
        dims = np.fromfile(prefix + ".dims", np.int32)
        assert dims.tolist() == [w, h], (name, dims)
        np.save(os.path.join(HERE, name + "_texels.npy"), np.fromfile(prefix + ".u8", np.uint8).reshape(h, w, 3))
        print(name, len(body), "bytes")

    pal = [int(v) for v in rng.integers(0, 256, 16 * 3)]
    encode("png_rgb8_13x7", 13, 7, 2, 8, 0)
    encode("png_rgba8_9x9_adam7", 9, 9, 6, 8, 1, split_idat=True)
    encode("png_grey8_17x5", 17, 5, 0, 8, 0)
    encode("png_grey4_11x6", 11, 6, 0, 4, 0)
    encode("png_grey2_10x4_adam7", 10, 4, 0, 2, 1)
    encode("png_grey1_19x3", 19, 3, 0, 1, 0, trns=[0, 1])
    encode("png_greyalpha8_8x8", 8, 8, 4, 8, 0)
    encode("png_grey16_6x5", 6, 5, 0, 16, 0)
    encode("png_rgb16_7x4_adam7", 7, 4, 2, 16, 1)
    encode("png_rgba16_5x5", 5, 5, 6, 16, 0)
    encode("png_pal8_12x5", 12, 5, 3, 8, 0, palette=pal, trns=[0, 128, 255])
    encode("png_pal4_9x7_adam7", 9, 7, 3, 4, 1, palette=pal)
    encode("png_pal2_15x2", 15, 2, 3, 2, 0, palette=pal[:12])
    encode("png_pal1_21x3", 21, 3, 3, 1, 0, palette=pal[:6])
    encode("png_rgb8_1x1_stored", 1, 1, 2, 8, 0, level=0)
    encode("png_rgb8_64x48_fixedhuff", 64, 48, 2, 8, 0, level=1)


if __name__ == "__main__":
    if "--jpeg-only" not in sys.argv and "--png-only" not in sys.argv:
        main()
    if "--png-only" not in sys.argv:
        make_jpeg_fixtures()
        make_flat_progressive_jpeg()
    if "--jpeg-only" not in sys.argv:
        make_png_fixtures()
