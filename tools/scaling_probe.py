"""Developer probe: how does the render kernel's time scale with spp / depth / ablation flags?"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracingoneweekendapplication_amd as rt

scene = rt.Scene.build("book1_final")
r = rt.Renderer(0)
r.upload(scene)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream

def run(W, H, spp, depth, variant, real=rt.RTK_REAL_F64, count=False):
    cam = scene.camera(W, H, spp, depth)
    img = torch.zeros((H, W, 3), dtype=torch.float64 if real == rt.RTK_REAL_F64 else torch.float32, device=dev)
    cnt = torch.zeros(12, dtype=torch.int64, device=dev)
    best = 1e9
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cnt.zero_()
        e0.record()
        r.render_device(cam, img.data_ptr(), 0, real_mode=real, variant=variant, d_counters=cnt.data_ptr() if count else 0, stream=stream)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    c = dict(zip(rt.COUNTER_FIELDS, cnt.tolist()))
    return best, float(img.mean()), c

for (W, H, spp, depth, variant) in [(1920, 1080, 100, 50, 0), (1920, 1080, 10, 50, 0), (1920, 1080, 100, 1, 0),
                                   (960, 540, 100, 50, 0), (1920, 1080, 1, 50, 0), (1920, 1080, 100, 50, 1)]:
    ms, mean, _ = run(W, H, spp, depth, variant)
    print(f"{W}x{H} spp {spp:4d} depth {depth:3d} variant {variant:3d}: {ms:9.3f} ms  {W*H*spp/ms/1e3:9.1f} Msamples/s  image mean {mean:.5f}", flush=True)
