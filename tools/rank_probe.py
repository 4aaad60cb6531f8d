"""Developer probe: per-rank shard times and work shares for N ranks (all on this one GPU).
  python3 tools/rank_probe.py [N list=1,2,4,8] [spp=0]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, raytracingoneweekendapplication_amd as rt
from raytracingoneweekendapplication_amd import tiling
ns = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 4, 8]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 0
scene = rt.Scene.build("book1_final"); r = rt.Renderer(0); cam = scene.camera(0, 0, spp, 0)
r.upload_fast(scene, cam.center)
dev = torch.device("cuda", 0); W, H = cam.image_width, cam.image_height
for n in ns:
    tpr = tiling.tiles_per_rank(W, H, n)
    res = []
    for rank in range(n):
        buf = torch.empty((tpr, 3, 64), dtype=torch.float64, device=dev) if n > 1 else torch.empty((H, W, 3), dtype=torch.float64, device=dev)
        cnt = torch.zeros(12, dtype=torch.int64, device=dev)
        best = 1e9
        for rep in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r.render_device(cam, buf.data_ptr(), 0, rank=rank, n_ranks=n); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        r.render_device(cam, buf.data_ptr(), 0, rank=rank, n_ranks=n, d_counters=cnt.data_ptr()); torch.cuda.synchronize()
        c = dict(zip(rt.COUNTER_FIELDS, cnt.tolist()))
        res.append((best, c["segments"], c["box_tests"]))
    ms = [x[0] for x in res]; seg = [x[1] for x in res]
    print(f"N={n}: ms per rank {[round(m,3) for m in ms]} max {max(ms):.3f} mean {sum(ms)/n:.3f} sum {sum(ms):.3f}; segments rel {[round(s*n/sum(seg),3) for s in seg]}", flush=True)
