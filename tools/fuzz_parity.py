"""A larger device-vs-oracle sweep than the test suite affords: random scenes (tests/test_fast_order_random.py's generators:
soups of spheres / quads / triangles with moving spheres, nested lists and rotate_y / translate instances; fog scenes with
sphere-bounded media inside one another and around the camera; "zoo" scenes with all seven materials, the five textures, a
textured medium and point lights; "degenerate" scenes of exact ties, zero radii, zero-area primitives and a sphere around the camera;
"big" scenes of 1 800 - 4 200 primitives whose traversal program does not fit LDS), each rendered on the device in the reference order and in
the fast order and compared with the CPU oracle of the same description at the same seed.

  python3 tools/fuzz_parity.py [n_per_family=100] [first_seed=50000]

Per scene: per-channel RMSE < 1e-12 against the oracle, equal u8 bytes, equal work counters (reference order); the fast
order's framebuffer equal to the reference order's wherever rtk_scene_optimize reports it exact; every fourth scene also
rendered as the interleaved tiles of 2 / 3 / 5 / 8 ranks, gathered and un-permuted: the same doubles and bytes.  Prints one line per
failure and a summary; exit code 1 on any failure.  (Test infrastructure: it uses oracle/ as the checker.)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import raytracingoneweekendapplication_amd as rt  # noqa: E402
from oracle import orc  # noqa: E402
from tests.test_fast_order_random import look_at_camera, random_fog_scene, random_big_scene, random_degenerate_scene, random_scene, random_sphere_scene, random_zoo_scene  # noqa: E402


def random_camera(rnd):
    """A camera of its own for every scene: ragged sizes (not multiples of the 8x8 tile), aspect ratios, fields of view, a lens or
    none, 1-6 samples, depth 1-9, a background colour (Camera.txt:136-175 through host/rtk_camera.h)."""
    cam = rt.derive_camera(rnd.randint(17, 90), rnd.choice([16 / 9, 1.0, 4 / 3, 2.35, 0.7]), spp=rnd.randint(1, 6), max_depth=rnd.randint(1, 9),
                           vfov=rnd.uniform(30, 100), lookfrom=(rnd.uniform(-1, 1), rnd.uniform(-0.3, 1.5), rnd.uniform(-0.5, 1.5)),
                           lookat=(rnd.uniform(-1, 1), rnd.uniform(-0.5, 1), rnd.uniform(-6, -4)), defocus_angle=rnd.choice([0.0, 0.0, 0.6, 2.5]),
                           focus_dist=rnd.uniform(3, 8))
    cam.background = rt.Vec3(rnd.uniform(0, 1), rnd.uniform(0, 1), rnd.uniform(0, 1))
    return cam


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
    renderer = rt.Renderer(0)
    import random
    fixed = look_at_camera(rt)
    families = (("spheres only (the lean MIXED kernel)", random_sphere_scene), ("soup", lambda s: random_scene(s)), ("soup+triangles", lambda s: random_scene(s, triangles=True)), ("fog", random_fog_scene), ("zoo", random_zoo_scene), ("degenerate", random_degenerate_scene),
                ("big (programs larger than LDS)", random_big_scene))
    from raytracingoneweekendapplication_amd import tiling
    dev = torch.device("cuda", 0)
    failures, worst, total, sharded = 0, 0.0, 0, 0
    kernels = {}
    for name, make in families:
        exact_fast = 0
        count = max(4, n // 10) if name.startswith("big") else n      # thousands of primitives each: a tenth as many scenes
        for k in range(count):
            seed = first + k
            scene = make(seed)
            cam = fixed if k % 2 == 0 else random_camera(random.Random(seed))   # every other scene through a camera of its own
            ref, ref8, ocnt = orc.render(scene.desc_ptr, cam, 7, 8)
            renderer.upload(scene)
            gpu, gpu8, cnt = renderer.render_host(cam, seed=7, count=True)
            err = float(np.sqrt(np.mean((gpu - ref) ** 2)))
            worst = max(worst, err)
            ok = err < 1e-12 and np.array_equal(gpu8, ref8) and cnt == ocnt
            info = renderer.upload_fast(scene, cam.center)
            kernels[renderer.kernel_name()] = kernels.get(renderer.kernel_name(), 0) + 1
            fast, fast8, _ = renderer.render_host(cam, seed=7)
            if info["exact"]:
                exact_fast += 1
                ok = ok and np.array_equal(fast, gpu) and np.array_equal(fast8, gpu8)
            else:
                ok = ok and float(np.sqrt(np.mean((fast - ref) ** 2))) < 0.25   # statistically the same picture, other random numbers
            if k % 4 == 1:    # the same frame cut into the interleaved tiles of 2-8 ranks (each rendered here), gathered and un-permuted
                n_ranks = (2, 3, 5, 8)[(k // 4) % 4]
                W, H = cam.image_width, cam.image_height
                tpr = tiling.tiles_per_rank(W, H, n_ranks)
                parts = []
                for rank in range(n_ranks):
                    buf = torch.full((tpr, 3, 64), float("nan"), dtype=torch.float64, device=dev)
                    renderer.render_device(cam, buf.data_ptr(), 0, seed=7, rank=rank, n_ranks=n_ranks)
                    parts.append(buf)
                gathered = torch.stack(parts).contiguous()
                image = torch.empty((H, W, 3), dtype=torch.float64, device=dev)
                rgb8 = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
                renderer.unpermute(W, H, n_ranks, rt.RTK_REAL_F64, gathered.data_ptr(), image.data_ptr(), rgb8.data_ptr())
                torch.cuda.synchronize()
                same = np.array_equal(image.cpu().numpy(), fast) and np.array_equal(rgb8.cpu().numpy(), fast8)
                sharded += 1
                if not same:
                    ok = False
                    print(f"     {name} seed {seed}: the frame of {n_ranks} ranks differs from the one-GPU frame", flush=True)
            total += 1
            if not ok:
                failures += 1
                print(f"FAIL {name} seed {seed}: rmse {err:.3e} bytes {np.array_equal(gpu8, ref8)} counters {cnt == ocnt} fast-exact {info['exact']}", flush=True)
        print(f"{name}: {count} scenes, fast order reported exact for {exact_fast}", flush=True)
    print(f"{sharded} of the scenes also rendered as the tiles of 2 / 3 / 5 / 8 ranks, gathered and un-permuted: equal to the one-GPU frame unless reported above", flush=True)
    print(f"{total} scenes, {failures} failures, worst RMSE against the oracle {worst:.3e}; kernels of the fast order: {kernels}", flush=True)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
