"""What do the procedural / image textures cost the full-feature kernel?  Renders a config scene as built, then with its
noise textures replaced by a solid colour, then with image textures replaced as well (the geometry and the paths' lengths
change a little -- other attenuations -- but not the traversal work): the difference is what texture evaluation inside
the shade step costs.  Developer probe, GPU box only.

  python3 tools/texture_cost_probe.py [config=c5] [spp=32]
"""
import ctypes as C, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracingoneweekendapplication_amd as rt
from tests.desc_builder import SceneDesc

config = sys.argv[1] if len(sys.argv) > 1 else "c5"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
TEX_SOLID, TEX_IMAGE, TEX_NOISE = 1, 4, 5
tmp = tempfile.mkdtemp()
earth = rt.write_synthetic_earth(os.path.join(tmp, "earth_synth.ppm"))
dev = torch.device("cuda", 0)
for label, drop in (("as built", ()), ("noise -> solid", (TEX_NOISE,)), ("noise, image -> solid", (TEX_NOISE, TEX_IMAGE))):
    scene = rt.Scene.build(rt.CONFIG_SCENES[config], rt.SCENE_SEED, earth)
    d = SceneDesc.from_address(scene.desc_ptr)
    for i in range(d.n_textures):
        if d.textures[i].kind in drop:
            d.textures[i].kind = TEX_SOLID
            d.textures[i].color.x = d.textures[i].color.y = d.textures[i].color.z = 0.5
    cam = scene.camera(0, 0, spp, 0)
    r = rt.Renderer(0)
    r.upload_fast(scene, cam.center)
    img = torch.empty((cam.image_height, cam.image_width, 3), dtype=torch.float64, device=dev)
    for k in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r.render_device(cam, img.data_ptr(), 0, stream=torch.cuda.current_stream().cuda_stream)
        e1.record()
        e1.synchronize()
    print(f"{label:24s} {e0.elapsed_time(e1):8.3f} ms  {r.kernel_name()}", flush=True)
