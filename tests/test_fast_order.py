"""CPU tests of rtk_scene_optimize (the fast visiting order).

The optimiser is host code of librtk_hip.so and needs no device.  Its output is an ordinary
rtk_scene_desc, so the CPU oracle can execute it: for scenes where the pass claims exactness the
oracle's image on the re-grouped hierarchy must equal its image on the reference's own hierarchy
bit for bit (that hierarchy is pinned against the reference's classes in test_oracle_goldens.py),
with the same RNG draws and segments, and fewer aabb::hit calls.
"""
import ctypes as C

import numpy as np
import pytest

from tests.conftest import GOLDEN
from tests.scene_cases import IMAGE_CASES, RENDER_SEED, SCENE_SEED, scene_file

NODE_SPHERE, NODE_QUAD, NODE_TRI, NODE_LIST, NODE_BVH, NODE_TRANSLATE, NODE_ROTATE, NODE_MEDIUM = range(1, 9)


class Node(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("c", C.c_int32)]


class DescHead(C.Structure):  # the leading fields of rtk_scene_desc (include/rtk.h)
    _fields_ = [("abi_version", C.c_int32), ("root", C.c_int32), ("n_nodes", C.c_int32), ("n_list_children", C.c_int32),
                ("n_spheres", C.c_int32), ("n_quads", C.c_int32), ("n_triangles", C.c_int32), ("n_bvh_boxes", C.c_int32),
                ("n_translates", C.c_int32), ("n_rotates", C.c_int32), ("n_media", C.c_int32), ("n_materials", C.c_int32),
                ("n_textures", C.c_int32), ("n_images", C.c_int32), ("n_perlins", C.c_int32), ("n_lights", C.c_int32),
                ("n_texel_bytes", C.c_int64), ("nodes", C.POINTER(Node)), ("list_children", C.POINTER(C.c_int32))]


def reachable_primitives(desc_ptr):
    """Multiset of (kind, index, transform path) of the primitives the root reaches, with multiplicity collapsed:
    what the scene IS, independent of how it is grouped."""
    d = DescHead.from_address(desc_ptr)
    found = set()
    seen = 0

    def walk(node, path, depth):
        nonlocal seen
        assert 0 <= node < d.n_nodes and depth < 200
        seen += 1
        assert seen < 10_000_000
        n = d.nodes[node]
        if n.kind in (NODE_SPHERE, NODE_QUAD, NODE_TRI):
            found.add((n.kind, n.a, path))
        elif n.kind == NODE_LIST:
            assert 0 <= n.a and n.a + n.b <= d.n_list_children
            for k in range(n.b):
                walk(d.list_children[n.a + k], path, depth + 1)
        elif n.kind == NODE_BVH:
            assert 0 <= n.c < d.n_bvh_boxes
            walk(n.a, path, depth + 1)
            walk(n.b, path, depth + 1)
        elif n.kind in (NODE_TRANSLATE, NODE_ROTATE, NODE_MEDIUM):
            walk(n.b, path + ((n.kind, n.a),), depth + 1)
        else:
            raise AssertionError(f"unknown node kind {n.kind}")

    walk(d.root, (), 0)
    return found


def build(rt, name):
    return rt.Scene.build(name, SCENE_SEED, scene_file(name, GOLDEN))


@pytest.mark.parametrize("case", IMAGE_CASES, ids=[c[0] for c in IMAGE_CASES])
def test_fast_order_keeps_the_scene_and_the_image(rt, orc, case):
    name, w, h, spp, depth = case
    scene = build(rt, name)
    cam = scene.camera(w, h, spp, depth)
    fast = scene.fast_order(cam.center)
    assert reachable_primitives(fast.desc_ptr) == reachable_primitives(scene.desc_ptr)
    has_media = DescHead.from_address(scene.desc_ptr).n_media > 0
    has_tris = any(k == NODE_TRI for (k, _, _) in reachable_primitives(scene.desc_ptr))
    assert fast.info["has_media"] == has_media and fast.info["has_triangles"] == has_tris
    # exact = bit-identical to the reference order: closest hits are preserved, exact ties follow the reference's ranks, and a
    # medium (RNG draws inside hit()) keeps its position in the reference's visiting order.
    # Triangle scenes are flagged separately (float determinant caveat).
    assert fast.exact
    # ... PROVEN (rtk_optimize_info.exact == 2) exactly when the scene has no triangles: triangle.h:72,77 scales t by a float
    # reciprocal, so the reference's own boxes may cull a hit its triangle::hit accepts, depending on its order -- identical in
    # every measurement (below, and on the device), but a measurement: exact == 1
    assert fast.proven == (not has_tris) and fast.info["exactness"] == (1 if has_tris else 2)
    assert (fast.info["n_ordered_items"] > 0) == has_media

    ref, ref8, ref_cnt = orc.render(scene.desc_ptr, cam, RENDER_SEED, 4)
    got, got8, got_cnt = orc.render(fast.desc_ptr, cam, RENDER_SEED, 4)
    # exact scenes by construction; triangle scenes because no hit sits within float rounding of a box face
    # at these sizes (see rtk_optimize.cpp on triangle.h:72,77)
    assert np.array_equal(got, ref) and np.array_equal(got8, ref8)
    for k in ("samples", "segments", "surface_hits", "rng_draws", "noise_calls", "texel_fetches"):
        assert got_cnt[k] == ref_cnt[k], k   # (medium_tests may differ: a medium called where the reference skips it returns false before it draws)
    if has_media:
        # media re-grouped like everything else (opts.free_media_order): a medium draws inside hit(), so the RNG order
        # changes -- same estimator, other image -- and the pass says so
        free = scene.fast_order(cam.center, free_media_order=True)
        assert not free.exact and not free.proven and free.info["exactness"] == 0 and free.info["n_ordered_items"] == 0
        assert reachable_primitives(free.desc_ptr) == reachable_primitives(scene.desc_ptr)
        got, _, got_cnt = orc.render(free.desc_ptr, cam, RENDER_SEED, 4)
        assert got_cnt["samples"] == ref_cnt["samples"]
        assert abs(got.mean() - ref.mean()) < 0.08 * ref.mean() + 1e-3
        assert abs(got_cnt["segments"] - ref_cnt["segments"]) < 0.05 * ref_cnt["segments"]


def test_fast_order_cuts_the_slab_tests_of_the_benchmark_scenes(rt, orc):
    for name, w, h, spp, floor in (("book1_final", 96, 54, 2, 0.6), ("mesh", 96, 54, 2, 0.6)):
        scene = build(rt, name)
        cam = scene.camera(w, h, spp, 0)
        fast = scene.fast_order(cam.center)
        _, _, a = orc.render(scene.desc_ptr, cam, RENDER_SEED, 4)
        _, _, b = orc.render(fast.desc_ptr, cam, RENDER_SEED, 4)
        assert b["box_tests"] < floor * a["box_tests"], (name, a["box_tests"], b["box_tests"])
        prim = lambda c: c["sphere_tests"] + c["quad_tests"] + c["triangle_tests"]
        assert prim(b) <= prim(a)


def test_boxes_that_are_entered_anyway_are_dropped(rt, orc, monkeypatch):
    """A boxed pair whose box a ray enters whenever it entered the enclosing one is not worth its slab test (a box() of six
    quads, the walls of a room: every level of a binary hierarchy over them spans the same box).  The optimiser prices each
    pair with and without its box (rtk_optimize.cpp, `Built::first`) and hands the members of the cheaper form to the enclosing
    node: the same image bit for bit, a third fewer slab tests in the Cornell box, never more primitive tests than a tenth
    above.  RTK_OPT_FLATTEN=0 (tools/) keeps every box."""
    for name, w, h, spp, at_most in (("cornell_box", 60, 60, 4, 0.75), ("book2_final", 96, 54, 2, 0.99), ("mesh", 96, 54, 2, 0.99), ("book1_final", 96, 54, 2, 1.0)):
        scene = build(rt, name)
        cam = scene.camera(w, h, spp, 0)
        monkeypatch.setenv("RTK_OPT_FLATTEN", "0")
        every_box = scene.fast_order(cam.center)
        monkeypatch.delenv("RTK_OPT_FLATTEN")
        fast = scene.fast_order(cam.center)
        assert fast.info["n_bvh_nodes_out"] < every_box.info["n_bvh_nodes_out"]
        img_a, _, a = orc.render(every_box.desc_ptr, cam, RENDER_SEED, 4)
        img_b, _, b = orc.render(fast.desc_ptr, cam, RENDER_SEED, 4)
        assert np.array_equal(img_a, img_b) and a["rng_draws"] == b["rng_draws"]
        assert b["box_tests"] <= at_most * a["box_tests"], (name, a["box_tests"], b["box_tests"])
        prim = lambda c: c["sphere_tests"] + c["quad_tests"] + c["triangle_tests"]
        assert prim(b) <= 1.1 * prim(a), (name, prim(a), prim(b))


def test_fast_order_is_deterministic_and_eye_is_optional(rt, orc):
    scene = build(rt, "book1_final")
    cam = scene.camera(48, 27, 2, 50)
    ref, _, _ = orc.render(scene.desc_ptr, cam, RENDER_SEED, 4)
    images = []
    for eye in (cam.center, cam.center, None):
        f = scene.fast_order(eye)
        img, _, cnt = orc.render(f.desc_ptr, cam, RENDER_SEED, 4)
        images.append((img, cnt["box_tests"]))
        assert np.array_equal(img, ref)
    assert images[0][1] == images[1][1]  # same input, same hierarchy


def test_optimize_rejects_bad_input(rt):
    lib = rt.hip_lib()
    out = C.c_void_p()
    assert lib.rtk_scene_optimize(None, None, C.byref(out), None) == -1
    scene = build(rt, "three_spheres")
    head = DescHead.from_address(scene.desc_ptr)
    saved = head.root
    try:
        head.root = head.n_nodes + 5
        assert lib.rtk_scene_optimize(scene.desc_ptr, None, C.byref(out), None) == -1 and not out.value
    finally:
        head.root = saved
    assert lib.rtk_scene_optimize(scene.desc_ptr, None, C.byref(out), None) == 0 and out.value  # opts and info may be NULL
    lib.rtk_scene_optimized_free(out)
