"""Exact ties -- two primitives hit at EXACTLY the same distance -- in re-grouped hierarchies.

The reference resolves such a tie by its visiting order: sphere::hit accepts a root only strictly inside the interval
(interval::surrounds, sphere.h:44-48), so the earlier of two spheres stays; quad::hit / triangle::hit accept t == tmax
(quad.h:39, triangle.h:91), so the later quad or triangle replaces what was there, a sphere included.  rtk_scene_optimize
visits in another order; it records every primitive's rank in the reference's order (rtk_node.c) and the kernels -- and
the oracle, when a description carries ranks -- reproduce the reference's outcome from them.

The scenes here are hand-built so that ties are certain: pairs of coincident quads, of identical triangles and of
identical spheres with DIFFERENT materials (the image shows who won), and a sphere touching a quad's plane on the view
axis.  Each is rendered as a plain hittable_list in a given order (= the reference) and as the same primitives listed in
every other order with the ranks of the first; all of them must give the first image, bit for bit.
"""
import ctypes as C
import itertools

import numpy as np
import pytest

from tests.desc_builder import DescBuilder


def tie_scene(order, with_ranks=True):
    """Six primitives in three coincident pairs + a ground sphere, listed in `order` (a permutation of range(7));
    ranks (rtk_node.c) are those of the identity order."""
    b = DescBuilder()
    red, green, blue, yellow, cyan, magenta, grey = (b.lambertian(c) for c in ((0.9, 0.1, 0.1), (0.1, 0.9, 0.1), (0.1, 0.1, 0.9), (0.9, 0.9, 0.1),
                                                                             (0.1, 0.9, 0.9), (0.9, 0.1, 0.9), (0.5, 0.5, 0.5)))
    prims = [
        lambda: b.quad((-2.0, -0.5, -4.0), (1.5, 0.0, 0.0), (0.0, 1.5, 0.0), red),         # a quad and a larger coplanar one around it:
        lambda: b.quad((-2.5, -1.0, -4.0), (2.5, 0.0, 0.0), (0.0, 2.5, 0.0), green),       #   same normal, same D -> same t (quad.h:29-41)
        lambda: b.triangle((0.25, -0.5, -4.5), (1.75, -0.5, -4.5), (0.25, 1.0, -4.5), blue),
        lambda: b.triangle((0.25, -0.5, -4.5), (1.75, -0.5, -4.5), (0.25, 1.0, -4.5), yellow),   # the same triangle again
        lambda: b.sphere((0.0, 1.5, -5.0), 0.75, cyan),
        lambda: b.sphere((0.0, 1.5, -5.0), 0.75, magenta),                                  # the same sphere again
        lambda: b.sphere((0.0, -101.0, -5.0), 100.0, grey),
    ]
    nodes = {}
    for k in order:          # table order follows the listing order as well, as it would in a re-built scene
        nodes[k] = prims[k]()
        if with_ranks:
            b.rank(nodes[k], k + 1)
    return b.finish(b.list([nodes[k] for k in order]))


ORDERS = [(0, 1, 2, 3, 4, 5, 6), (1, 0, 3, 2, 5, 4, 6), (6, 5, 4, 3, 2, 1, 0), (3, 5, 1, 6, 0, 4, 2), (2, 0, 4, 6, 5, 3, 1)]


def camera(rt, w=64, h=36, spp=4, depth=4):
    return rt.Scene.build("three_spheres").camera(w, h, spp, depth)   # at the origin, looking down -z


def test_oracle_resolves_exact_ties_by_reference_rank_in_every_visiting_order(rt, orc):
    cam = camera(rt)
    scene = tie_scene(ORDERS[0], with_ranks=False)      # (a BuiltDesc owns its tables: keep it alive while its pointer is in use)
    reference, ref8, ref_cnt = orc.render(scene.desc_ptr, cam, 3, 4)
    differs_without_ranks = 0
    for order in ORDERS:
        scene = tie_scene(order)
        ranked, ranked8, cnt = orc.render(scene.desc_ptr, cam, 3, 4)
        assert np.array_equal(ranked, reference) and np.array_equal(ranked8, ref8), order
        assert cnt == ref_cnt                                     # a flat list tests everything: identical work, identical RNG use
        scene = tie_scene(order, with_ranks=False)
        plain, _, _ = orc.render(scene.desc_ptr, cam, 3, 4)
        differs_without_ranks += int(not np.array_equal(plain, reference))
    assert differs_without_ranks >= 3       # the ties are real: without ranks the visiting order shows in the image


def test_optimiser_output_carries_the_ranks_and_keeps_the_image(rt, orc):
    """rtk_scene_optimize on the tie scenes: its hierarchy visits in an order of its own, the image is the reference's."""
    cam = camera(rt)
    for order in ORDERS:
        scene = tie_scene(order, with_ranks=False)          # a reference-order description: ranks are the optimiser's job
        reference, ref8, _ = orc.render(scene.desc_ptr, cam, 3, 4)
        for eye in (cam.center, None):
            fast = rt.FastOrderScene(scene, eye)
            assert fast.exact
            got, got8, _ = orc.render(fast.desc_ptr, cam, 3, 4)
            assert np.array_equal(got, reference) and np.array_equal(got8, ref8), (order, eye is None)


def _hit(orc, desc_ptr, root, ray, tmin=0.001, tmax=float("inf")):
    out = (C.c_double * 12)()
    draws = C.c_uint64()
    r = (C.c_double * 7)(*ray)
    ok = orc.lib().orc_kat_node_hit(desc_ptr, root, r, tmin, tmax, 1, 0, 0, out, C.byref(draws))
    return ok, list(out)


def sphere_on_plane(order, with_ranks=True):
    """A unit sphere whose nearest point and a quad's plane coincide on the -z axis (both at t = 3 for the ray below,
    exactly: sphere.h:35-43 gives (4 - 1) / 1, quad.h:34 gives (-3 - 0) / -1), and a second sphere for sphere-sphere."""
    b = DescBuilder()
    m = [b.lambertian((0.2 * k, 0.5, 0.5)) for k in range(3)]
    prims = [lambda: b.sphere((0.0, 0.0, -4.0), 1.0, m[0]), lambda: b.quad((-1.0, -1.0, -3.0), (2.0, 0.0, 0.0), (0.0, 2.0, 0.0), m[1]),
             lambda: b.sphere((0.0, 0.0, -4.0), 1.0, m[2])]
    nodes = {}
    for k in order:
        nodes[k] = prims[k]()
        if with_ranks:
            b.rank(nodes[k], k + 1)
    root = b.list([nodes[k] for k in order])
    return b.finish(root), root


AXIS_RAY = (0.0, 0.0, 0.0, 0.0, 0.0, -1.0, 0.0)


def test_sphere_against_quad_tie_on_the_axis(rt, orc):
    """Mixed kinds: whatever the listing order, the quad's inclusive test beats a sphere's strict one (the reference:
    sphere then quad -> the quad replaces it; quad then sphere -> the sphere's root is not strictly inside)."""
    for ref_order in itertools.permutations(range(3)):
        scene, root = sphere_on_plane(ref_order, with_ranks=False)
        ok, rec = _hit(orc, scene.desc_ptr, root, AXIS_RAY)
        assert ok and rec[0] == 3.0 and rec[10] == 1.0          # t = 3 exactly; material 1 = the quad, in every reference order
    # sphere-sphere alone: the earlier one in the reference order, whatever order it is visited in
    for order in ((0, 2), (2, 0)):
        b = DescBuilder()
        m = [b.lambertian((0.3 * k, 0.5, 0.5)) for k in range(3)]
        first = b.rank(b.sphere((0.0, 0.0, -4.0), 1.0, m[0]), 1)
        second = b.rank(b.sphere((0.0, 0.0, -4.0), 1.0, m[2]), 2)
        root = b.list([first, second] if order == (0, 2) else [second, first])
        scene = b.finish(root)
        ok, rec = _hit(orc, scene.desc_ptr, root, AXIS_RAY)
        assert ok and rec[0] == 3.0 and rec[10] == 0.0          # rank 1 wins in either visiting order


@pytest.mark.gpu
def test_kernels_resolve_exact_ties_like_the_reference(rt, orc, renderer):
    """The device: the same scenes through the fast-order kernels (rtk_scene_upload_optimized on hand-permuted lists with
    ranks, and rtk_scene_upload_fast on the plain lists) against the reference-order render -- f64, bit for bit."""
    cam = camera(rt, 96, 54, 4, 4)
    plain = tie_scene(ORDERS[0], with_ranks=False)
    renderer.upload(plain)
    reference, ref8, _ = renderer.render_host(cam, seed=3)
    oracle, _, _ = orc.render(plain.desc_ptr, cam, 3, 4)
    assert float(np.sqrt(np.mean((reference - oracle) ** 2))) < 1e-12
    for order in ORDERS:
        ranked = tie_scene(order)
        renderer.upload_optimized(ranked, cam.center)
        assert ", 256u," not in renderer.kernel_name() and int(renderer.kernel_name().split(",")[1].strip(" u")) & 256    # a COMPACT (f32-box) kernel
        got, got8, _ = renderer.render_host(cam, seed=3)
        assert np.array_equal(got, reference) and np.array_equal(got8, ref8), order
        slot, _, _ = renderer.render_host(cam, seed=3, variant=1 << 20)      # the slot program with f64 boxes: same rule, other rank table
        assert np.array_equal(slot, reference), order
        f32, _, _ = renderer.render_host(cam, seed=3, real_mode=rt.RTK_REAL_F32)
        assert abs(f32.mean() - reference.mean()) < 0.03 * reference.mean()
        unranked = tie_scene(order, with_ranks=False)
        info = renderer.upload_fast(unranked, cam.center)   # ranks from rtk_scene_optimize itself
        assert info["exact"]
        # (reference of THIS listing order)
        renderer_ref = rt.Renderer(0)
        renderer_ref.upload(unranked)
        want, want8, _ = renderer_ref.render_host(cam, seed=3)
        got, got8, _ = renderer.render_host(cam, seed=3)
        assert np.array_equal(got, want) and np.array_equal(got8, want8), order
    # the sphere / quad tie on the axis, through the device's closest-hit entry point (slot program + rank table)
    for order in itertools.permutations(range(3)):
        scene, _ = sphere_on_plane(order)
        renderer.upload_optimized(scene)
        rec, _ = renderer.closest_hit(np.array([list(AXIS_RAY) + [0.001, np.inf]]), np.array([[1, 0, 0]]))
        assert rec[0, 0] == 1.0 and rec[0, 1] == 3.0 and rec[0, 11] == 1.0      # the quad (rank 2), hit at t = 3
    scene, _ = sphere_on_plane((2, 0))
    renderer.upload_optimized(scene)
    rec, _ = renderer.closest_hit(np.array([list(AXIS_RAY) + [0.001, np.inf]]), np.array([[1, 0, 0]]))
    assert rec[0, 1] == 3.0 and rec[0, 11] == 0.0                                  # sphere of rank 1, although visited second
