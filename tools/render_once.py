"""Profiling target: upload a config scene and launch the render kernel a fixed number of times.

  python3 tools/render_once.py [config=c2] [real=f64|f32] [launches=2] [spp=0] [variant=0] [order=auto|reference|fast]
Used under rocprofv3 (--kernel-trace --stats, or --pmc passes); prints the event-timed kernel ms.
"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracingoneweekendapplication_amd as rt

config = sys.argv[1] if len(sys.argv) > 1 else "c2"
real = rt.RTK_REAL_F64 if (len(sys.argv) <= 2 or sys.argv[2] == "f64") else rt.RTK_REAL_F32
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 2
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 0
variant = int(sys.argv[5]) if len(sys.argv) > 5 else 0
order = sys.argv[6] if len(sys.argv) > 6 else "auto"
tmp = tempfile.mkdtemp()
earth = rt.write_synthetic_earth(os.path.join(tmp, "earth_synth.ppm"))
scene = rt.Scene.build(rt.CONFIG_SCENES[config], rt.SCENE_SEED, earth)
cam = scene.camera(0, 0, spp, 0)
r = rt.Renderer(0)
fast = scene.fast_order(cam.center)
use_fast = order == "fast" or (order == "auto" and fast.exact)  # same rule as bench.py
prim_cost, free_media = float(os.environ.get("AB_PRIM_COST", "0")), os.environ.get("AB_FREE_MEDIA", "0") == "1"   # sweeps of the optimiser's knobs
max_leaf = int(os.environ.get("AB_MAX_LEAF", "0"))
r.upload_fast(scene, cam.center, max_leaf=max_leaf, prim_cost_scale=prim_cost, free_media_order=free_media) if use_fast else r.upload(scene)
print(f"order: {'fast (rtk_scene_upload_fast)' if use_fast else 'reference (bvh.h)'}", flush=True)
dev = torch.device("cuda", 0)
H, W = cam.image_height, cam.image_width
img = torch.empty((H, W, 3), dtype=torch.float64 if real == rt.RTK_REAL_F64 else torch.float32, device=dev)
u8 = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for k in range(launches):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r.render_device(cam, img.data_ptr(), u8.data_ptr(), real_mode=real, variant=variant, stream=stream)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"launch {k}: {ms:.3f} ms  {W*H*cam.samples_per_pixel/ms/1e3:.1f} Msamples/s  kernel={r.kernel_name(real, variant)}", flush=True)
import hashlib
print("framebuffer sha256", hashlib.sha256(img.cpu().numpy().tobytes()).hexdigest()[:16], flush=True)
