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
    """Small baseline JPEGs written by Pillow (synthetic content) + the bytes the REFERENCE's rtw_image holds for
    them (ref_driver texels): 4:4:4, 4:2:0, 4:2:2, grey, sizes that are not multiples of the MCU, restart markers."""
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
        im.save(path, "JPEG", optimize=False, progressive=False, **kw)
        prefix = os.path.join(tempfile.mkdtemp(), name)
        subprocess.check_call([REF, "texels", path, prefix])
        dims = np.fromfile(prefix + ".dims", np.int32)
        assert dims.tolist() == [w, h], (name, dims)
        np.save(os.path.join(HERE, name + "_texels.npy"), np.fromfile(prefix + ".u8", np.uint8).reshape(h, w, 3))
        print(name, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    if "--jpeg-only" not in sys.argv:
        main()
    make_jpeg_fixtures()
