"""Marginal cost of a feature inside the SAME kernel: render book 2's final scene with one feature's data made trivial.

  python3 tools/knockout.py [spp]

Textures do not steer paths (an attenuation only scales the throughput; noise and image textures draw no random numbers), so
turning the noise / image texture into a solid colour leaves every ray, every record visit and every RNG stream as it was:
the time difference is what the kernel spends on that texture.  Thin media (density -> 0) never scatter: the difference is the
cost of the isotropic scatters and of the paths they lengthen or shorten (paths DO change there).
The description is patched in place through ctypes (a dev tool: the tables belong to librtk_host.so).
"""
import ctypes as C
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import raytracingoneweekendapplication_amd as rt  # noqa: E402


class Tex(C.Structure):
    _fields_ = [("kind", C.c_int32), ("even", C.c_int32), ("odd", C.c_int32), ("image", C.c_int32), ("color", C.c_double * 3), ("param", C.c_double)]


class Med(C.Structure):
    _fields_ = [("neg_inv_density", C.c_double), ("material", C.c_int32), ("_pad", C.c_int32)]


class Node(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("c", C.c_int32)]


def empty_the_media(desc_ptr):
    """Every constant_medium node becomes an empty hittable_list (hittable_list::hit of no objects: false, no random number drawn).
    The hierarchy above it keeps its shape and its boxes, so against 'media thin' (same paths: neither scatters) the difference
    is the cost of the medium steps themselves -- two per segment in book 2's final scene."""
    ints = (C.c_int32 * 16).from_address(desc_ptr)
    ptrs = (C.c_void_p * 15).from_address(desc_ptr + 72)
    nodes = (Node * ints[2]).from_address(ptrs[0])
    n = 0
    for node in nodes:
        if node.kind == 8:
            node.kind, node.a, node.b, node.c = 4, 0, 0, 0
            n += 1
    return n


def tables(desc_ptr):
    ints = (C.c_int32 * 16).from_address(desc_ptr)
    ptrs = (C.c_void_p * 15).from_address(desc_ptr + 72)
    n_media, n_textures = ints[10], ints[12]
    media = (Med * n_media).from_address(ptrs[8]) if n_media else []
    texs = (Tex * n_textures).from_address(ptrs[10]) if n_textures else []
    return media, texs


def render(scene, spp, label):
    cam = scene.camera(0, 0, spp, 0)
    r = rt.Renderer(0)
    fast = scene.fast_order(cam.center)
    r.upload_fast(scene, cam.center) if fast.exact else r.upload(scene)
    H, W = cam.image_height, cam.image_width
    img = torch.empty((H, W, 3), dtype=torch.float64, device="cuda:0")
    u8 = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r.render_device(cam, img.data_ptr(), u8.data_ptr(), real_mode=rt.RTK_REAL_F64, variant=0, stream=stream)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"{label:28s} {best:8.3f} ms  kernel={r.kernel_name(rt.RTK_REAL_F64, 0)}", flush=True)
    return best


if __name__ == "__main__":
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    tmp = tempfile.mkdtemp()
    earth = rt.write_synthetic_earth(os.path.join(tmp, "earth_synth.ppm"))

    def fresh():
        return rt.Scene.build(rt.CONFIG_SCENES["c5"], rt.SCENE_SEED, earth)

    base = render(fresh(), spp, "as is")
    for label, kinds, thin in (("noise -> solid", (5,), False), ("image -> solid", (4,), False), ("noise, image -> solid", (4, 5), False),
                               ("media thin (paths change)", (), True), ("all three", (4, 5), True)):
        sc = fresh()
        media, texs = tables(sc.desc_ptr)
        for t in texs:
            if t.kind in kinds:
                t.kind = 1
                t.color[0] = t.color[1] = t.color[2] = 0.5
        if thin:
            for m in media:
                m.neg_inv_density = -1e30
        t_ms = render(sc, spp, label)
        print(f"    {label}: {100 * (base - t_ms) / base:5.1f} % of the frame", flush=True)
    sc = fresh()
    n = empty_the_media(sc.desc_ptr)
    t_ms = render(sc, spp, f"media emptied ({n} nodes)")
    print(f"    media emptied: {100 * (base - t_ms) / base:5.1f} % of the frame (against 'media thin': the medium steps themselves)", flush=True)
