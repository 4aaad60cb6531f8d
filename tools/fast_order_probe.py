"""Developer probe: reference visiting order vs rtk_scene_optimize's fast order on one GPU.

  python3 tools/fast_order_probe.py [configs=c2,c3,c4,c5] [spp overrides c3=100,c5=100]
Prints kernel ms per frame for both orders (f64 and f32), whether the images are bit-identical, and the
exact work counters per sample.
"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracingoneweekendapplication_amd as rt

configs = (sys.argv[1] if len(sys.argv) > 1 else "c2,c3,c4,c5").split(",")
spp_over = {"c3": 100, "c5": 100}
settings = [(4, 1.0), (2, 1.0), (8, 1.0), (4, 1.6), (4, 0.6)]
if len(sys.argv) > 2:
    settings = [tuple(float(x) for x in s.split(":")) for s in sys.argv[2].split(",")]
tmp = tempfile.mkdtemp()
earth = rt.write_synthetic_earth(os.path.join(tmp, "earth_synth.ppm"))
dev = torch.device("cuda", 0)
r = rt.Renderer(0)
stream = torch.cuda.current_stream().cuda_stream


def measure(scene, cam, real, launches=3, fast=None):
    if fast is not None:
        r.upload_fast(scene, cam.center, *fast)
    else:
        r.upload(scene)
    H, W = cam.image_height, cam.image_width
    img = torch.empty((H, W, 3), dtype=torch.float64 if real == rt.RTK_REAL_F64 else torch.float32, device=dev)
    best = 1e30
    for _ in range(launches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r.render_device(cam, img.data_ptr(), 0, real_mode=real, stream=stream)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    cnt = torch.zeros(12, dtype=torch.int64, device=dev)
    scratch = torch.empty_like(img)
    r.render_device(cam, scratch.data_ptr(), 0, real_mode=real, d_counters=cnt.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    c = dict(zip(rt.COUNTER_FIELDS, [int(v) for v in cnt.tolist()]))
    return best, img, c, r.kernel_name(real)


for config in configs:
    scene = rt.Scene.build(rt.CONFIG_SCENES[config], rt.SCENE_SEED, earth)
    cam = scene.camera(0, 0, spp_over.get(config, 0), 0)
    n = cam.image_width * cam.image_height * cam.samples_per_pixel
    for real, rname in ((rt.RTK_REAL_F64, "f64"), (rt.RTK_REAL_F32, "f32")):
        ms0, img0, c0, k0 = measure(scene, cam, real)
        print(f"{config} {rname} reference order: {ms0:9.3f} ms  {n/ms0/1e3:8.1f} Msamples/s  box {c0['box_tests']/n:.2f} sph {c0['sphere_tests']/n:.2f} quad {c0['quad_tests']/n:.2f} "
              f"tri {c0['triangle_tests']/n:.2f} xf {c0['xform_enters']/n:.2f}  {k0}", flush=True)
        for max_leaf, pcs in settings:
            fast = scene.fast_order(cam.center, int(max_leaf), pcs)
            ms1, img1, c1, k1 = measure(scene, cam, real, fast=(int(max_leaf), pcs))
            same = bool(torch.equal(img0, img1))
            print(f"{config} {rname} fast order leaf<={int(max_leaf)} cost x{pcs}: {ms1:9.3f} ms  {n/ms1/1e3:8.1f} Msamples/s  x{ms0/ms1:.2f}  identical={same} exact={fast.exact} "
                  f"box {c1['box_tests']/n:.2f} sph {c1['sphere_tests']/n:.2f} quad {c1['quad_tests']/n:.2f} tri {c1['triangle_tests']/n:.2f} xf {c1['xform_enters']/n:.2f} "
                  f"ops {r.scene_info()['program_ops']}", flush=True)
            if real == rt.RTK_REAL_F64 and not same:
                d = (img0 - img1).abs()
                bad = (d.amax(-1) > 0).nonzero()
                print(f"      max |diff| {float(d.max()):.3e}  differing pixels {bad.shape[0]}  first (row, col): {bad[:5].tolist()}", flush=True)
                for (j, i) in bad[:3].tolist():
                    print(f"      pixel ({j},{i}): reference {img0[j, i].tolist()}  fast {img1[j, i].tolist()}", flush=True)
