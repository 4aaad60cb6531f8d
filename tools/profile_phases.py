"""Where do the render kernel's wave-cycles go?  Builds a diagnostic copy of the kernel library with
-DRTK_PROFILE (s_memtime stamps at the scheduler's phase boundaries), renders one frame and prints,
per phase: share of wave-cycles, steps, cycles per step, mean active lanes.  Tools only.

  python3 tools/profile_phases.py [config=c2] [real=f64] [spp=0] [order=auto|reference|fast] [variant=0] [rank=0] [n_ranks=1]
"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
extra = os.environ.get("RTK_PROF_FLAGS", "").split()   # e.g. RTK_PROF_FLAGS="-DRTK_PARK_MASK=0"
prof_lib = os.path.join(ROOT, "gpurun_out", "librtk_hip_prof%s.so" % ("_" + "_".join(f.strip("-D").replace("=", "") for f in extra) if extra else ""))
os.makedirs(os.path.dirname(prof_lib), exist_ok=True)
csrc = os.path.join(ROOT, "raytracingoneweekendapplication_amd", "csrc")
if os.environ.get("RTK_PROF_LIB"):   # a profile build made beforehand (tools/ab/build_local.sh name:"-DRTK_PROFILE"): nothing is compiled on the GPU box
    prof_lib = os.environ["RTK_PROF_LIB"]
elif not os.path.exists(prof_lib) or os.path.getmtime(prof_lib) < os.path.getmtime(os.path.join(csrc, "rtk_trace.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DRTK_PROFILE", *extra,
                           "-I" + os.path.join(ROOT, "include"), "-I" + csrc, os.path.join(csrc, "rtk_api.cpp"), os.path.join(csrc, "rtk_multi.cpp"), os.path.join(csrc, "rtk_optimize.cpp"), os.path.join(csrc, "rtk_trace.hip"), "-o", prof_lib])
os.environ["RTK_HIP_LIB"] = prof_lib
os.environ["RTK_DEV_TOOLS"] = "1"
import torch
import raytracingoneweekendapplication_amd as rt

config = sys.argv[1] if len(sys.argv) > 1 else "c2"
real = rt.RTK_REAL_F64 if (len(sys.argv) <= 2 or sys.argv[2] == "f64") else rt.RTK_REAL_F32
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 0
order = sys.argv[4] if len(sys.argv) > 4 else "auto"
variant = int(sys.argv[5]) if len(sys.argv) > 5 else 0
rank = int(sys.argv[6]) if len(sys.argv) > 6 else 0
n_ranks = int(sys.argv[7]) if len(sys.argv) > 7 else 1
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
img = torch.empty((H, W, 3), dtype=torch.float64 if real == rt.RTK_REAL_F64 else torch.float32, device=dev)  # large enough for any shard
prof = torch.zeros(32 + 4 * 4096 + 16, dtype=torch.int64, device=dev)
# the profile build writes its counters through the d_counters pointer of the (non-counting) kernel
lib = rt.hip_lib()
import ctypes as C
opts = rt.RenderOpts(rt.RENDER_SEED, real, rank, n_ranks, 0, variant, torch.cuda.current_stream().cuda_stream)
for k in range(3):
    prof.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = lib.rtk_render_device(r._ctx, C.byref(cam), C.byref(opts), img.data_ptr(), None, prof.data_ptr())
    assert rc == 0, lib.rtk_last_error()
    e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1)
raw = prof.cpu().numpy()
import numpy as np
v = np.concatenate([raw[:18], raw[25:31], raw[32 + 4 * 4096 + 8:32 + 4 * 4096 + 14]]).reshape(10, 3).astype(float)
total = v[:, 0].sum()
names = ["refill+vote", "box step", "sphere step", "shade: ray_color", "other op", "quad/tri step", "shade: new sample", "shade: new segment", "shade: vote -> entry", "(unused)"]
print(f"{config} {'f64' if real == rt.RTK_REAL_F64 else 'f32'} {W}x{H}x{cam.samples_per_pixel}: {ms:.2f} ms (instrumented)")
print(f"{'phase':24s} {'cycles %':>9s} {'steps':>14s} {'cyc/step':>9s} {'lanes/step':>10s}")
for n, (t, steps, lanes) in zip(names, v):
    if steps:
        print(f"{n:24s} {100 * t / total:9.1f} {int(steps):14d} {t / steps:9.1f} {lanes / steps:10.1f}")
sub = raw[32 + 4 * 4096:32 + 4 * 4096 + 8].astype(float)
if sub[1]:
    print("inside ray_color, as lane 0 saw it (wave-cycles per call of the part x calls = % of all phases): " + ", ".join(
        f"{n} {sub[2 * k] / max(sub[2 * k + 1], 1):.0f} cyc x {int(sub[2 * k + 1])} = {100 * sub[2 * k] / total:.1f} %" for k, n in enumerate(("hit record", "materials + textures", "miss path", "scattered ray"))))
waves = int(raw[22])
if waves:
    span = (int(raw[24]) - ((1 << 62) - int(raw[23]))) / 100.0   # us
    print(f"waves {waves}: kernel span {span:.1f} us; wave lifetime mean {raw[18] / waves / 100.0:.1f} us ({100 * raw[18] / waves / 100.0 / span:.1f} % of the span), max {raw[19] / 100.0:.1f} us; "
          f"after the work queue ran dry: mean {raw[20] / waves / 100.0:.1f} us, max {raw[21] / 100.0:.1f} us")
    first = (1 << 62) - int(raw[23])
    n_long = int(raw[31])
    recs = raw[32:32 + 4 * min(n_long, 4096)].reshape(-1, 4)
    print(f"(pixel, chunk)s that ended after the work queue ran dry and took over 300 us: {n_long}")
    if len(recs):
        ends = recs[:, 0] + recs[:, 1]
        for i in ends.argsort()[::-1][:16]:
            b, dur, segs, where = (int(x) for x in recs[i])
            print(f"   began {(b - first) / 100.0:9.1f} us  took {dur / 100.0:8.1f} us  ended {(b + dur - first) / 100.0:9.1f} us  segments {segs:5d}  slot {where >> 8} pixel {where & 255}")
        import numpy as np
        print(f"   durations: median {np.median(recs[:, 1]) / 100.0:.1f} us, max {recs[:, 1].max() / 100.0:.1f} us; segments median {np.median(recs[:, 2]):.0f}, max {recs[:, 2].max()}")
