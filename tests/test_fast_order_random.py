"""CPU property test of rtk_scene_optimize on random scenes that scene_library.h does not cover: soups of spheres
(static and moving) and quads, nested lists, rotate_y/translate instances of sub-lists, metal / glass / lambertian / light
materials.  The reference order here is the plainest possible -- a flat hittable_list tests every object for every ray
(hittable_list.h:22-35), no box in sight -- so the oracle's image of it is ground truth for the closest hits, and the
oracle's image of the optimised hierarchy (SAH boxes recomputed from the primitives, rotated instance boxes, margins,
boxless runs, dropped root box) must equal it bit for bit."""
import ctypes as C
import math
import random

import numpy as np
import pytest

from tests.desc_builder import DescBuilder


def random_scene(seed, triangles=False):
    rnd = random.Random(seed)
    b = DescBuilder()
    mats = [b.lambertian((rnd.random(), rnd.random(), rnd.random())) for _ in range(3)]
    mats += [b.metal((0.8, 0.7, 0.6), rnd.random() * 0.4), b.dielectric(1.5), b.light((4.0, 4.0, 3.5))]

    def prim():
        m = rnd.choice(mats)
        if rnd.random() < 0.65:
            c = (rnd.uniform(-4, 4), rnd.uniform(-1, 3), rnd.uniform(-8, -2))
            motion = (rnd.uniform(-0.3, 0.3), rnd.uniform(-0.2, 0.2), 0.0) if rnd.random() < 0.25 else (0.0, 0.0, 0.0)
            return b.sphere(c, rnd.uniform(0.15, 0.9), m, motion)
        q = (rnd.uniform(-4, 3), rnd.uniform(-1, 2), rnd.uniform(-8, -3))
        u = (rnd.uniform(0.3, 1.5), rnd.uniform(-0.3, 0.3), rnd.uniform(-0.5, 0.5))
        v = (rnd.uniform(-0.3, 0.3), rnd.uniform(0.3, 1.5), rnd.uniform(-0.5, 0.5))
        if triangles and rnd.random() < 0.5:   # a triangle on the same three corners (triangle.h: float determinant and barycentrics)
            return b.triangle(q, tuple(q[k] + u[k] for k in range(3)), tuple(q[k] + v[k] for k in range(3)), m)
        return b.quad(q, u, v, m)

    top = [b.sphere((0, -101, -5), 100.0, mats[0])]           # a ground sphere: one huge box among small ones
    if triangles:                                             # at least one triangle whatever the draws
        top.append(b.triangle((-1.0, -0.5, -4.0), (1.0, -0.5, -4.5), (0.0, 1.2, -4.2), mats[1], ((0.1, 0.2), (0.9, 0.1), (0.5, 0.8))))
    for _ in range(rnd.randint(6, 14)):
        top.append(prim())
    for _ in range(rnd.randint(1, 3)):                        # instances of small sub-lists
        sub = b.list([prim() for _ in range(rnd.randint(1, 4))])
        inst = sub
        if rnd.random() < 0.7:
            inst = b.rotate_y(inst, rnd.uniform(-40, 40))
        if rnd.random() < 0.8:
            inst = b.translate(inst, (rnd.uniform(-1, 1), rnd.uniform(-0.5, 0.5), rnd.uniform(-1, 1)))
        if rnd.random() < 0.3:
            inst = b.rotate_y(inst, rnd.uniform(-20, 20))   # two rotations around a translation
        top.append(inst)
    if rnd.random() < 0.5:                                   # a nested list, and the same object listed twice
        top.append(b.list([prim(), top[1]]))
    rnd.shuffle(top)
    return b.finish(b.list(top))


def random_zoo_scene(seed):
    """Every feature at once, the way tests/golden's material_zoo has them in ONE fixed arrangement: spheres, quads and triangles
    under instances, all seven material kinds, the five textures (checker over noise / image, the uv checker, random RGB8 images, an
    image of width 0), a sphere-bounded medium with a textured phase function, and one or two point lights (Camera.txt:240-272)."""
    rnd = random.Random(seed)
    b = DescBuilder()
    img = b.image(rnd.randint(3, 17), rnd.randint(2, 11), rnd)
    nse = b.noise(rnd.uniform(0.5, 6.0), rnd)
    chk = b.checker(rnd.uniform(0.2, 1.5), b.solid((0.2, 0.3, 0.1)), rnd.choice([nse, img, b.solid((0.9, 0.9, 0.9))]))
    chk_uv = b.checker(rnd.uniform(0.1, 0.6), b.solid((0.8, 0.2, 0.2)), b.solid((0.1, 0.1, 0.7)), by_uv=True)
    textures = [img, nse, chk, chk_uv, b.image(0, 0, rnd)]
    mats = [b.textured(t) for t in textures]
    mats += [b.lambertian((rnd.random(), rnd.random(), rnd.random())), b.metal((0.8, 0.7, 0.6), rnd.random() * 0.5), b.dielectric(rnd.choice([1.5, 1.33, 1 / 1.5])),
             b.specular((0.7, 0.6, 0.5), rnd.uniform(2.0, 40.0)), b.light((4.0, 4.0, 3.5)), b.textured(rnd.choice(textures), kind=4)]

    def prim(tri_ok=True):
        m = rnd.choice(mats)
        r = rnd.random()
        if r < 0.5:
            c = (rnd.uniform(-4, 4), rnd.uniform(-1, 3), rnd.uniform(-8, -2))
            motion = (rnd.uniform(-0.3, 0.3), rnd.uniform(-0.2, 0.2), 0.0) if rnd.random() < 0.2 else (0.0, 0.0, 0.0)
            return b.sphere(c, rnd.uniform(0.2, 1.0), m, motion)
        q = (rnd.uniform(-4, 3), rnd.uniform(-1, 2), rnd.uniform(-8, -3))
        u = (rnd.uniform(0.4, 1.8), rnd.uniform(-0.3, 0.3), rnd.uniform(-0.5, 0.5))
        v = (rnd.uniform(-0.3, 0.3), rnd.uniform(0.4, 1.8), rnd.uniform(-0.5, 0.5))
        if tri_ok and r < 0.75:
            uvs = tuple((rnd.random(), rnd.random()) for _ in range(3))
            return b.triangle(q, tuple(q[k] + u[k] for k in range(3)), tuple(q[k] + v[k] for k in range(3)), m, uvs)
        return b.quad(q, u, v, m)

    top = [b.sphere((0, -101, -5), 100.0, mats[2])]          # the ground wears the checker
    for _ in range(rnd.randint(6, 12)):
        top.append(prim())
    for _ in range(rnd.randint(1, 2)):
        inst = b.list([prim() for _ in range(rnd.randint(1, 3))])
        if rnd.random() < 0.7:
            inst = b.rotate_y(inst, rnd.uniform(-40, 40))
        if rnd.random() < 0.8:
            inst = b.translate(inst, (rnd.uniform(-1, 1), rnd.uniform(-0.5, 0.5), rnd.uniform(-1, 1)))
        top.append(inst)
    if rnd.random() < 0.7:                                   # a fog ball whose phase function is textured (isotropic over a texture)
        shell = b.sphere((rnd.uniform(-2, 2), rnd.uniform(0, 1.5), rnd.uniform(-6, -3)), rnd.uniform(0.7, 1.6), mats[7])
        # (its texture reads the hit POINT only: constant_medium::hit leaves rec.u / rec.v as the record held them, constant_medium.h:45-50,
        # SURVEY Q12 -- a phase function over an image or a uv checker reads stale values in the reference and 0 on the device)
        phase = b.textured(rnd.choice([nse, b.solid((rnd.random(), rnd.random(), rnd.random()))]), kind=5)
        b.media.append(__import__("tests.desc_builder", fromlist=["Medium"]).Medium(-1.0 / rnd.uniform(0.3, 2.0), phase, 0))
        top.append(b._node(8, len(b.media) - 1, shell))
    for _ in range(rnd.randint(0, 2)):
        b.point_light((rnd.uniform(-3, 3), rnd.uniform(2, 5), rnd.uniform(-6, 0)), (rnd.uniform(2, 9), rnd.uniform(2, 9), rnd.uniform(2, 9)), rnd.uniform(0.1, 1.5))
    rnd.shuffle(top)
    return b.finish(b.list(top))


def random_big_scene(seed, n_prims=None):
    """Thousands of small primitives under the reference's kind of hierarchy (a bvh_node over everything, bvh.h:13-45): a
    traversal program that does not fit the 160 KB of LDS, so the device runs its boxes-in-LDS / hot-cold kernels -- on content
    no named scene has (spheres static and moving, quads and triangles mixed, three materials, a light, a fog ball)."""
    rnd = random.Random(seed)
    b = DescBuilder()
    mats = [b.lambertian((rnd.random(), rnd.random(), rnd.random())) for _ in range(3)]
    mats += [b.metal((0.8, 0.7, 0.6), rnd.random() * 0.4), b.dielectric(1.5), b.light((4.0, 4.0, 3.5))]
    n = n_prims or rnd.randint(1800, 4200)
    kinds = rnd.choice([(1.0, 0.0), (0.5, 0.5), (0.4, 0.3), (0.0, 1.0)])     # (share of spheres, share of quads): the rest are triangles
    members = [b.sphere((0, -101, -5), 100.0, mats[0])]
    for _ in range(n):
        c = (rnd.uniform(-6, 6), rnd.uniform(-0.9, 4), rnd.uniform(-12, -2))
        m = rnd.choice(mats)
        r = rnd.random()
        if r < kinds[0]:
            motion = (rnd.uniform(-0.1, 0.1), rnd.uniform(-0.1, 0.1), 0.0) if rnd.random() < 0.1 else (0.0, 0.0, 0.0)
            members.append(b.sphere(c, rnd.uniform(0.03, 0.2), m, motion))
            continue
        u = (rnd.uniform(0.1, 0.4), rnd.uniform(-0.1, 0.1), rnd.uniform(-0.15, 0.15))
        v = (rnd.uniform(-0.1, 0.1), rnd.uniform(0.1, 0.4), rnd.uniform(-0.15, 0.15))
        if r < kinds[0] + kinds[1]:
            members.append(b.quad(c, u, v, m))
        else:
            members.append(b.triangle(c, tuple(c[k] + u[k] for k in range(3)), tuple(c[k] + v[k] for k in range(3)), m))
    top = [b.bvh(members, rnd)]
    if rnd.random() < 0.5:
        shell = b.sphere((rnd.uniform(-2, 2), rnd.uniform(0, 1.5), rnd.uniform(-6, -3)), rnd.uniform(0.7, 1.6), mats[4])
        top += [b.medium(shell, rnd.uniform(0.3, 2.0), (rnd.random(), rnd.random(), rnd.random())), shell]
    return b.finish(b.list(top))


def random_sphere_scene(seed):
    """Spheres only -- what the lean MIXED kernel (the headline kernel: f32 centre / half-extent culling boxes, f64 spheres) runs:
    5 to 700 of them, static and moving, the three book-1 materials, radii over four orders of magnitude, clusters far from the
    origin (the float boxes' error budget scales with the coordinates), spheres inside spheres and around the camera."""
    rnd = random.Random(seed)
    b = DescBuilder()
    mats = [b.lambertian((rnd.random(), rnd.random(), rnd.random())) for _ in range(4)]
    mats += [b.metal((0.8, 0.7, 0.6), rnd.random() * 0.5), b.metal((0.9, 0.9, 0.9), 0.0), b.dielectric(1.5), b.dielectric(1 / 1.5)]
    top = [b.sphere((0, -1001, -5), 1000.0, mats[0])]
    n = rnd.choice([5, 20, 80, 300, 700])
    far = rnd.choice([0.0, 0.0, 300.0, 5000.0])              # a second cluster far away (seen small, or not at all)
    for k in range(n):
        c = (rnd.uniform(-5, 5), rnd.uniform(-0.9, 3), rnd.uniform(-12, -2))
        if far and k % 3 == 0:
            c = (c[0] + far, c[1], c[2] - far)
        r = rnd.choice([0.02, 0.1, 0.3, 0.3, 0.8, 2.0]) * rnd.uniform(0.5, 1.5)
        motion = (rnd.uniform(-0.4, 0.4), rnd.uniform(-0.3, 0.3), rnd.uniform(-0.2, 0.2)) if rnd.random() < 0.15 else (0.0, 0.0, 0.0)
        top.append(b.sphere(c, r, rnd.choice(mats), motion))
        if rnd.random() < 0.05:
            top.append(b.sphere(c, 0.5 * r, rnd.choice(mats)))       # one inside the other (a bubble, main.cpp:25-26)
    if rnd.random() < 0.2:
        top.append(b.sphere((0.0, 0.5, 0.0), rnd.uniform(20, 40), mats[6]))   # a glass ball around the camera and the scene
    rnd.shuffle(top)
    return b.finish(b.list(top))


def random_degenerate_scene(seed):
    """Soups seeded with the cases closest-hit code gets wrong first: the same primitive listed twice with two materials and
    coplanar overlapping quads / triangles (exact ties: the reference's visiting order decides, hittable_list.h:29-33,
    interval::surrounds vs contains), spheres of radius 0 (sphere.h:14 fmax(0, r)), a zero-area triangle and a needle quad,
    a sphere around the camera, glass of index 1, metal of fuzz 0 and 1, primitives touching one another."""
    rnd = random.Random(seed)
    b = DescBuilder()
    mats = [b.lambertian((rnd.random(), rnd.random(), rnd.random())) for _ in range(3)]
    mats += [b.metal((0.8, 0.7, 0.6), 0.0), b.metal((0.6, 0.7, 0.8), 1.0), b.dielectric(1.0), b.dielectric(1.5), b.light((4.0, 4.0, 3.5))]
    top = [b.sphere((0, -101, -5), 100.0, mats[0])]
    for _ in range(rnd.randint(5, 10)):
        m, m2 = rnd.choice(mats), rnd.choice(mats)
        c = (rnd.uniform(-3, 3), rnd.uniform(-0.8, 2), rnd.uniform(-7, -2.5))
        u = (rnd.uniform(0.5, 1.6), 0.0, 0.0)
        v = (0.0, rnd.uniform(0.5, 1.6), 0.0)
        case = rnd.randrange(9)
        if case == 0:      # the same sphere twice, two materials
            r = rnd.uniform(0.3, 0.9)
            top += [b.sphere(c, r, m), b.sphere(c, r, m2)]
        elif case == 1:    # the same quad twice
            top += [b.quad(c, u, v, m), b.quad(c, u, v, m2)]
        elif case == 2:    # coplanar, overlapping, axis-aligned quads (the overlap is an exact tie for every ray)
            top += [b.quad(c, u, v, m), b.quad((c[0] + 0.25 * u[0], c[1] + 0.25 * v[1], c[2]), u, v, m2)]
        elif case == 3:    # a triangle lying in a quad
            top += [b.quad(c, u, v, m), b.triangle(c, (c[0] + u[0], c[1], c[2]), (c[0], c[1] + v[1], c[2]), m2)]
        elif case == 4:    # radius 0 and a negative radius (both: a point nobody hits), next to a real sphere
            top += [b.sphere(c, 0.0, m), b.sphere((c[0] + 0.1, c[1], c[2]), -0.5, m2), b.sphere((c[0], c[1] + 0.5, c[2]), 0.4, m)]
        elif case == 5:    # a zero-area triangle and a needle
            top += [b.triangle(c, c, (c[0] + 1, c[1], c[2]), m), b.quad(c, (1.5, 0.0, 0.0), (1.5, 1e-9, 0.0), m2) if False else b.quad(c, (1.5, 0.0, 0.0), (0.0, 1e-7, 0.0), m2)]
        elif case == 6:    # two spheres touching in one point, and one inside the other
            r = rnd.uniform(0.3, 0.7)
            top += [b.sphere(c, r, m), b.sphere((c[0] + 2 * r, c[1], c[2]), r, m2), b.sphere(c, 0.5 * r, m2)]
        elif case == 7:    # a big sphere around the camera (every primary ray starts inside it)
            top.append(b.sphere((0.0, 0.0, 0.0), rnd.uniform(9, 14), rnd.choice(mats[:3] + [mats[6]])))
        else:
            top.append(b.sphere(c, rnd.uniform(0.2, 0.9), m, (rnd.uniform(-0.3, 0.3), 0.0, 0.0)))
    if rnd.random() < 0.4:
        inst = b.translate(b.rotate_y(b.list([top.pop(), top.pop()]), rnd.uniform(-30, 30)), (rnd.uniform(-0.5, 0.5), 0.0, 0.0))
        top.append(inst)
    rnd.shuffle(top)
    return b.finish(b.list(top))


def look_at_camera(rt):
    cam = rt.Scene.build("three_spheres").camera(48, 27, 3, 6)   # at the origin, looking down -z: the soup lies in front of it
    return cam


@pytest.mark.parametrize("seed", range(40))
def test_random_soups_render_identically_in_the_fast_order(rt, orc, seed):
    scene = random_scene(1000 + seed)
    cam = look_at_camera(rt)
    ref, ref8, rc = orc.render(scene.desc_ptr, cam, 7, 4)
    for eye in (cam.center, None):
        fast = rt.FastOrderScene(scene, eye)
        assert fast.exact
        got, got8, gc = orc.render(fast.desc_ptr, cam, 7, 4)
        assert np.array_equal(got, ref) and np.array_equal(got8, ref8), f"seed {seed}: max diff {np.abs(got - ref).max()}"
        for k in ("segments", "surface_hits", "rng_draws"):
            assert gc[k] == rc[k], k
        assert gc["box_tests"] > 0 and gc["sphere_tests"] + gc["quad_tests"] < rc["sphere_tests"] + rc["quad_tests"]
    assert ref.std() > 0.01   # the camera actually sees the soup


@pytest.mark.parametrize("seed", range(16))
def test_random_soups_with_triangles_render_identically_in_the_fast_order(rt, orc, seed):
    """The same with triangles among the primitives (triangle.h:65-113: float determinant and barycentrics): bit-identical
    again in every case -- as a measurement: the pass reports triangle scenes as empirically exact (1), never proven (2)."""
    scene = random_scene(3000 + seed, triangles=True)
    cam = look_at_camera(rt)
    ref, ref8, rc = orc.render(scene.desc_ptr, cam, 7, 4)
    assert rc["triangle_tests"] > 0
    for eye in (cam.center, None):
        fast = rt.FastOrderScene(scene, eye)
        assert fast.exact and fast.info["has_triangles"] and not fast.proven
        got, got8, gc = orc.render(fast.desc_ptr, cam, 7, 4)
        assert np.array_equal(got, ref) and np.array_equal(got8, ref8), f"seed {seed}: max diff {np.abs(got - ref).max()}"
        for k in ("segments", "surface_hits", "rng_draws"):
            assert gc[k] == rc[k], k


@pytest.mark.parametrize("seed", range(10))
def test_random_zoo_scenes_render_identically_in_the_fast_order(rt, orc, seed):
    """Every feature at once (all seven materials, the five textures, a textured medium, point lights, triangles and quads under
    instances): the re-grouped hierarchy renders the reference order's image bit for bit, with the same random numbers."""
    scene = random_zoo_scene(7000 + seed)
    cam = look_at_camera(rt)
    ref, ref8, rc = orc.render(scene.desc_ptr, cam, 7, 4)
    fast = rt.FastOrderScene(scene, cam.center)
    assert fast.exact and not fast.proven      # triangles: measured, not proven
    got, got8, gc = orc.render(fast.desc_ptr, cam, 7, 4)
    assert np.array_equal(got, ref) and np.array_equal(got8, ref8), f"seed {seed}: max diff {np.abs(got - ref).max()}"
    for k in ("segments", "surface_hits", "rng_draws", "noise_calls", "texel_fetches"):
        assert gc[k] == rc[k], k
    assert ref.std() > 0.01


@pytest.mark.parametrize("seed", range(4))
def test_random_big_scenes_render_identically_in_the_fast_order(rt, orc, seed):
    """Thousands of primitives under a reference-style bvh_node (programs larger than LDS on the device): the re-grouped
    hierarchy renders the same doubles with far fewer primitive tests."""
    scene = random_big_scene(9000 + seed, n_prims=2500)
    cam = look_at_camera(rt)
    ref, ref8, rc = orc.render(scene.desc_ptr, cam, 7, 4)
    fast = rt.FastOrderScene(scene, cam.center)
    assert fast.exact
    got, got8, gc = orc.render(fast.desc_ptr, cam, 7, 4)
    assert np.array_equal(got, ref) and np.array_equal(got8, ref8), f"seed {seed}: max diff {np.abs(got - ref).max()}"
    for k in ("segments", "surface_hits", "rng_draws"):
        assert gc[k] == rc[k], k
    assert ref.std() > 0.01


@pytest.mark.parametrize("seed", range(12))
def test_random_sphere_scenes_render_identically_in_the_fast_order(rt, orc, seed):
    """Sphere-only scenes (the lean MIXED kernel's domain): proven exact, and the re-grouped hierarchy renders the same doubles."""
    scene = random_sphere_scene(12000 + seed)
    cam = look_at_camera(rt)
    ref, ref8, rc = orc.render(scene.desc_ptr, cam, 7, 4)
    fast = rt.FastOrderScene(scene, cam.center)
    assert fast.exact and fast.proven
    got, got8, gc = orc.render(fast.desc_ptr, cam, 7, 4)
    assert np.array_equal(got, ref) and np.array_equal(got8, ref8), f"seed {seed}: max diff {np.abs(got - ref).max()}"
    for k in ("segments", "surface_hits", "rng_draws"):
        assert gc[k] == rc[k], k


@pytest.mark.parametrize("seed", range(30))
def test_random_degenerate_scenes_render_identically_in_the_fast_order(rt, orc, seed):
    """Exact ties (duplicates, coplanar overlaps), zero and negative radii, zero-area primitives, a sphere around the camera:
    the re-grouped hierarchy resolves every one of them the way the reference's visiting order does."""
    scene = random_degenerate_scene(11000 + seed)
    cam = look_at_camera(rt)
    ref, ref8, rc = orc.render(scene.desc_ptr, cam, 7, 4)
    fast = rt.FastOrderScene(scene, cam.center)
    assert fast.exact
    got, got8, gc = orc.render(fast.desc_ptr, cam, 7, 4)
    assert np.array_equal(got, ref) and np.array_equal(got8, ref8), f"seed {seed}: max diff {np.abs(got - ref).max()}"
    for k in ("segments", "surface_hits", "rng_draws"):
        assert gc[k] == rc[k], k


def test_degenerate_scene_of_coincident_spheres_is_handled(rt, orc):
    """Two thousand spheres on top of each other (identical boxes): the optimiser must neither recurse without bound nor
    change the image -- the first sphere in list order wins every tie in the reference (strict `surrounds`), and with
    identical materials any winner gives the same pixels."""
    from tests.desc_builder import DescBuilder

    b = DescBuilder()
    m = b.lambertian((0.6, 0.4, 0.3))
    members = [b.sphere((0.0, 0.0, -3.0), 0.8, m) for _ in range(2000)]
    members.append(b.sphere((0, -101, -3), 100.0, b.lambertian((0.5, 0.5, 0.5))))
    scene = b.finish(b.list(members))
    cam = look_at_camera(rt)
    fast = rt.FastOrderScene(scene, cam.center)
    ref, _, _ = orc.render(scene.desc_ptr, cam, 7, 4)
    got, _, cnt = orc.render(fast.desc_ptr, cam, 7, 4)
    assert np.array_equal(got, ref)
    assert cnt["sphere_tests"] > 0


def random_fog_scene(seed):
    """Spheres and constant media (bounded by spheres, some inside one another, one enclosing the camera) arranged the way
    the reference's scenes are: bvh_nodes over parts of the world, plain lists around them, a medium listed twice."""
    rnd = random.Random(seed)
    b = DescBuilder()
    mats = [b.lambertian((rnd.random(), rnd.random(), rnd.random())) for _ in range(3)]
    mats += [b.metal((0.8, 0.7, 0.6), rnd.random() * 0.4), b.dielectric(1.5), b.light((4.0, 4.0, 3.5))]

    def ball():
        c = (rnd.uniform(-4, 4), rnd.uniform(-1, 3), rnd.uniform(-9, -2))
        motion = (rnd.uniform(-0.3, 0.3), rnd.uniform(-0.2, 0.2), 0.0) if rnd.random() < 0.2 else (0.0, 0.0, 0.0)
        return b.sphere(c, rnd.uniform(0.15, 0.9), rnd.choice(mats), motion)

    def fog():
        c = (rnd.uniform(-3, 3), rnd.uniform(-0.5, 2), rnd.uniform(-8, -3))
        shell = b.sphere(c, rnd.uniform(0.6, 1.8), mats[4])
        parts = [b.medium(shell, rnd.uniform(0.3, 2.5), (rnd.random(), rnd.random(), rnd.random()))]
        if rnd.random() < 0.5:
            parts.append(shell)        # the glass boundary is an object of the world too (main.cpp:301-303)
        return parts

    def group(n_balls, n_fogs):
        members = [ball() for _ in range(n_balls)]
        for _ in range(n_fogs):
            members += fog()
        rnd.shuffle(members)
        return members

    top = [b.sphere((0, -101, -5), 100.0, mats[0])]
    top.append(b.bvh(group(rnd.randint(4, 12), rnd.randint(1, 2)), rnd))        # media deep inside a bvh_node
    top.append(b.bvh(group(rnd.randint(3, 10), 0), rnd))                        # a bvh_node without any
    top += group(rnd.randint(2, 5), rnd.randint(0, 1))                          # loose objects and media in the top-level list
    if rnd.random() < 0.6:                                                      # a thin haze around everything, camera included
        top.append(b.medium(b.sphere((0, 0, 0), 60.0, mats[4]), 0.02, (1, 1, 1)))
    if rnd.random() < 0.5:
        top.append(b.list(group(2, 1)))                                         # a nested list with a medium
    if rnd.random() < 0.3:
        top.append(b.bvh([m for m in fog() if True][:1], rnd))                  # a bvh_node of one medium: tested twice (bvh.h:30-32)
    rnd.shuffle(top)
    root = b.list(top)
    if rnd.random() < 0.5:                                                      # main.cpp:442: the whole world inside one bvh_node
        root = b.list([b.bvh(top, rnd)])
    return b.finish(root)


@pytest.mark.parametrize("seed", range(24))
def test_random_fog_scenes_render_identically_in_the_fast_order(rt, orc, seed):
    """constant_medium::hit draws inside hit() (constant_medium.h:40) and runs only when the bvh_nodes above it let the ray
    through with the interval as it stands at that moment: the optimiser keeps every medium's position in the reference's
    visiting order and re-groups the runs of objects in between, so the oracle's image of its output must still be the
    reference order's, bit for bit, draws included.  With free_media_order the pass says the image may differ."""
    scene = random_fog_scene(4000 + seed)
    cam = look_at_camera(rt)
    ref, ref8, rc = orc.render(scene.desc_ptr, cam, 7, 4)
    assert rc["medium_tests"] > 0
    for eye in (cam.center, None):
        fast = rt.FastOrderScene(scene, eye)
        assert fast.exact and fast.info["has_media"] and fast.info["n_ordered_items"] >= 1
        got, got8, gc = orc.render(fast.desc_ptr, cam, 7, 4)
        assert np.array_equal(got, ref) and np.array_equal(got8, ref8), f"seed {seed}: max diff {np.abs(got - ref).max()}"
        for k in ("segments", "surface_hits", "rng_draws"):
            assert gc[k] == rc[k], k
    free = rt.FastOrderScene(scene, cam.center, free_media_order=True)
    assert not free.exact and free.info["n_ordered_items"] == 0
    got, _, gc = orc.render(free.desc_ptr, cam, 7, 4)
    assert gc["samples"] == rc["samples"] and abs(got.mean() - ref.mean()) < 0.1 * ref.mean() + 1e-3
