"""Where do the render kernel's wave-cycles go?  Builds a diagnostic copy of the kernel library with
-DRTK_PROFILE (s_memtime stamps at the scheduler's phase boundaries), renders one frame and prints,
per phase: share of wave-cycles, steps, cycles per step, mean active lanes.  Tools only.

  python3 tools/profile_phases.py [config=c2] [real=f64] [spp=0] [order=auto|reference|fast] [variant=0]
"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
prof_lib = os.path.join(ROOT, "gpurun_out", "librtk_hip_prof.so")
os.makedirs(os.path.dirname(prof_lib), exist_ok=True)
csrc = os.path.join(ROOT, "raytracingoneweekendapplication_amd", "csrc")
if not os.path.exists(prof_lib) or os.path.getmtime(prof_lib) < os.path.getmtime(os.path.join(csrc, "rtk_trace.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DRTK_PROFILE",
                           "-I" + os.path.join(ROOT, "include"), "-I" + csrc, os.path.join(csrc, "rtk_api.cpp"), os.path.join(csrc, "rtk_optimize.cpp"), os.path.join(csrc, "rtk_trace.hip"), "-o", prof_lib])
os.environ["RTK_HIP_LIB"] = prof_lib
import torch
import raytracingoneweekendapplication_amd as rt

config = sys.argv[1] if len(sys.argv) > 1 else "c2"
real = rt.RTK_REAL_F64 if (len(sys.argv) <= 2 or sys.argv[2] == "f64") else rt.RTK_REAL_F32
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 0
order = sys.argv[4] if len(sys.argv) > 4 else "auto"
variant = int(sys.argv[5]) if len(sys.argv) > 5 else 0
tmp = tempfile.mkdtemp()
earth = rt.write_synthetic_earth(os.path.join(tmp, "earth_synth.ppm"))
scene = rt.Scene.build(rt.CONFIG_SCENES[config], rt.SCENE_SEED, earth)
cam = scene.camera(0, 0, spp, 0)
r = rt.Renderer(0)
fast = scene.fast_order(cam.center)
use_fast = order == "fast" or (order == "auto" and fast.exact)
r.upload_fast(scene, cam.center) if use_fast else r.upload(scene)
print("order:", "fast" if use_fast else "reference")
dev = torch.device("cuda", 0)
H, W = cam.image_height, cam.image_width
img = torch.empty((H, W, 3), dtype=torch.float64 if real == rt.RTK_REAL_F64 else torch.float32, device=dev)
prof = torch.zeros(18, dtype=torch.int64, device=dev)
# the profile build writes its counters through the d_counters pointer of the (non-counting) kernel
lib = rt.hip_lib()
import ctypes as C
opts = rt.RenderOpts(rt.RENDER_SEED, real, 0, 1, 0, variant, torch.cuda.current_stream().cuda_stream)
for k in range(2):
    prof.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = lib.rtk_render_device(r._ctx, C.byref(cam), C.byref(opts), img.data_ptr(), None, prof.data_ptr())
    assert rc == 0, lib.rtk_last_error()
    e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1)
v = prof.cpu().numpy().reshape(6, 3).astype(float)
total = v[:, 0].sum()
names = ["refill+vote", "box step", "sphere step", "shade+regen", "other op", "quad/tri step"]
print(f"{config} {'f64' if real == rt.RTK_REAL_F64 else 'f32'} {W}x{H}x{cam.samples_per_pixel}: {ms:.2f} ms (instrumented)")
print(f"{'phase':14s} {'cycles %':>9s} {'steps':>14s} {'cyc/step':>9s} {'lanes/step':>10s}")
for n, (t, steps, lanes) in zip(names, v):
    if steps:
        print(f"{n:14s} {100 * t / total:9.1f} {int(steps):14d} {t / steps:9.1f} {lanes / steps:10.1f}")
