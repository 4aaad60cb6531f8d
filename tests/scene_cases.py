"""Scene/size table shared by the golden generator and the tests."""
SCENE_SEED = 0x5EED2025
RENDER_SEED = 1

# (scene, width, height, spp, max_depth): small enough for the CPU oracle in seconds.
IMAGE_CASES = [
    ("three_spheres", 64, 36, 8, 10),
    ("book1_final", 64, 36, 4, 50),
    ("cornell_box", 40, 40, 8, 25),
    ("mesh", 64, 36, 4, 10),
    ("book2_final", 64, 36, 4, 10),
    ("material_zoo", 96, 54, 8, 12),
    ("cornell_smoke", 48, 48, 8, 10),
    ("single_fog", 48, 32, 16, 8),
    ("obj_mesh", 64, 36, 4, 10),
]

# File argument of rtk_build_named_scene per scene (relative to tests/golden/): the texture image for the
# textured scenes, the OBJ path for obj_mesh.
SCENE_FILES = {"obj_mesh": "quad_tri.obj"}


def scene_file(name, golden_dir):
    import os

    return os.path.join(golden_dir, SCENE_FILES.get(name, "earth_synth.ppm"))
