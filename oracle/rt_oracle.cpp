// rt_oracle.cpp -- CPU restatement of the reference's per-pixel sample loop.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product (the package, include/,
// csrc/) may include, link or call this file; only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg use it, and only as the checker.
//
// It restates, in plain scalar double-precision C++, the algorithm of
// /root/reference for the path camera::render -> get_ray -> ray_color ->
// hittable::hit -> material::scatter -> texture::value, function by function,
// keeping the reference's recursion structure (recursive ray_color, recursive
// bvh_node::hit visiting left then right, hittable_list's temp_rec copy) so that
// it is an independent statement of WHAT is computed, not of how the device
// kernel organises it (the kernel is iterative, stackless and defers the hit
// record).  Input is the index-linked scene description of include/rtk.h.
//
// Pinning: every function below is checked against the reference's own classes
// compiled from /root/reference (oracle/ref_driver.cpp -> oracle/_ref/), through
// the golden vectors in tests/golden/ (function-level KATs and seed-matched
// framebuffers).  The camera (Camera.txt) and point_light.h cannot be compiled
// here without stand-ins for windows.h / cuda_runtime.h, so get_ray, ray_color,
// get_lighting and initialize are restated from the text of Camera.txt in BOTH
// this file and ref_driver.cpp; everything they call is the real reference code
// in ref_driver.
//
// The RNG is the build's own (the reference's std::rand() is not reproducible,
// SURVEY Q9): PCG-RXS-M-XS-32, one stream per (seed, pixel, sample), 24-bit
// uniforms.  ref_driver interposes rand() with the same generator.
//
// Argument evaluation order: the reference writes vec3(random_double(),
// random_double(), random_double()) (vec3.h:50-56,137; Camera.txt:194); g++
// evaluates such arguments right to left, so the first draw goes to the LAST
// component.  The restatement follows what the g++-compiled reference does.

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../include/rtk.h"

namespace {

const double kInf = std::numeric_limits<double>::infinity();
const double kPi = 3.1415926535897932385;  // rtweekend.h:18

// ---------------------------------------------------------------- RNG -----
struct Rng {
    uint32_t s;
    uint64_t draws;
};
inline uint32_t pcg_hash(uint32_t v) {
    uint32_t st = v * 747796405u + 2891336453u;
    uint32_t w = ((st >> ((st >> 28u) + 4u)) ^ st) * 277803737u;
    return (w >> 22u) ^ w;
}
inline void rng_seed(Rng& g, uint32_t seed, uint32_t pixel, uint32_t sample) {
    g.s = pcg_hash(pixel + pcg_hash(sample + pcg_hash(seed)));
}
// rtweekend.h:26-29 random_double(): here u24 / 2^24.
inline double rnd(Rng& g) {
    uint32_t old = g.s;
    g.s = old * 747796405u + 2891336453u;
    uint32_t w = ((old >> ((old >> 28u) + 4u)) ^ old) * 277803737u;
    g.draws++;
    return double(((w >> 22u) ^ w) >> 8) * (1.0 / 16777216.0);
}
inline double rnd(Rng& g, double lo, double hi) { return lo + (hi - lo) * rnd(g); }  // rtweekend.h:30-33

// ---------------------------------------------------------------- vec3 ----
struct V3 {
    double x, y, z;
};
inline V3 v3(double a, double b, double c) { return V3{a, b, c}; }
inline V3 v3(const rtk_vec3& a) { return V3{a.x, a.y, a.z}; }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(double t, V3 a) { return v3(t * a.x, t * a.y, t * a.z); }
inline V3 operator/(V3 a, double t) { return (1 / t) * a; }                          // vec3.h:91-93
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }           // vec3.h:94-98
inline V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline double length_squared(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }     // vec3.h:46-48 (pow(e,2) == e*e)
inline double length(V3 a) { return std::sqrt(length_squared(a)); }
inline V3 unit_vector(V3 a) { return a / length(a); }                                // vec3.h:104-106
inline bool near_zero(V3 a) { return std::fabs(a.x) < 1e-8 && std::fabs(a.y) < 1e-8 && std::fabs(a.z) < 1e-8; }

// vec3.h:54-56 with g++ argument order: z, y, x.
inline V3 random_vec(Rng& g, double lo, double hi) {
    double c = rnd(g, lo, hi), b = rnd(g, lo, hi), a = rnd(g, lo, hi);
    return v3(a, b, c);
}
// vec3.h:107-115: the test `1e-160 < lensq <= 1` is always true.
inline V3 random_unit_vector(Rng& g) {
    V3 p = random_vec(g, -1, 1);
    double lensq = length_squared(p);
    return p / std::sqrt(lensq);
}
inline V3 random_on_hemisphere(Rng& g, V3 n) {  // vec3.h:116-124
    V3 s = random_unit_vector(g);
    return dot(s, n) > 0.0 ? s : -s;
}
inline V3 reflect(V3 v, V3 n) { return v - (2 * dot(v, n)) * n; }  // vec3.h:125-127
inline V3 refract(V3 uv, V3 n, double eta) {                        // vec3.h:128-133
    double cos_theta = std::fmin(dot(-uv, n), 1.0);
    V3 perp = eta * (uv + cos_theta * n);
    V3 par = (-std::sqrt(std::fabs(1.0 - length_squared(perp)))) * n;
    return perp + par;
}
inline V3 random_in_unit_disk(Rng& g) {  // vec3.h:135-142, y drawn first
    for (;;) {
        double b = rnd(g, -1, 1), a = rnd(g, -1, 1);
        V3 p = v3(a, b, 0);
        if (length_squared(p) < 1) return p;
    }
}

struct Ray {  // ray.h:6-32
    V3 o, d;
    double tm;
};
inline V3 at(const Ray& r, double t) { return r.o + t * r.d; }

struct HitRec {  // hittable.h:11-27
    V3 p{0, 0, 0}, normal{0, 0, 0};
    int mat = -1;
    double t = 0;
    bool front_face = false;
    double u = 0, v = 0;
    // not in the reference: which primitive produced the record -- its node kind and rtk_node.c (1 + rank in the
    // reference's visiting order; 0 in reference-order descriptions).  Only read by the tie rule below.
    int prim_kind = 0, rank = 0;
};

// ---- exact ties in a re-grouped hierarchy (rtk_scene_optimize output: primitive nodes carry rtk_node.c) ---------------
// The reference resolves two primitives hit at exactly the same t by its visiting order: sphere::hit accepts a root only
// strictly inside (tmin, tmax) (interval::surrounds, sphere.h:44-48), so the EARLIER sphere stays; quad::hit and
// triangle::hit accept t == tmax (interval::contains, quad.h:39; `t > ray_t.max` rejects, triangle.h:91), so the LATER one
// replaces it.  A re-grouped hierarchy visits in another order; with the reference ranks at hand the same outcome follows
// from: a quad/triangle beats a sphere; among quads/triangles the higher rank wins; among spheres the lower rank.  `Tie`
// says who set the tmax a test is called with.  In a reference-order description every rank is 0 and nothing changes.
struct Tie {
    int kind = 0, rank = 0;
};
inline Tie tie_of(const HitRec& rec) { return Tie{rec.prim_kind, rec.rank}; }
inline bool strict_kind(int k) { return k == RTK_NODE_SPHERE; }
inline bool inclusive_kind(int k) { return k == RTK_NODE_QUAD || k == RTK_NODE_TRIANGLE; }
inline bool tie_sphere_wins(int my_rank, const Tie& best) { return my_rank != 0 && best.rank != 0 && strict_kind(best.kind) && my_rank < best.rank; }
inline bool tie_inclusive_wins(int my_rank, const Tie& best) { return !inclusive_kind(best.kind) || my_rank == 0 || best.rank == 0 || my_rank > best.rank; }
inline void set_face_normal(HitRec& rec, const Ray& r, V3 outward) {  // hittable.h:23-26
    rec.front_face = dot(r.d, outward) < 0;
    rec.normal = rec.front_face ? outward : -outward;
}

struct Ctx {
    const rtk_scene_desc* sc;
    Rng rng;
    rtk_work_counters cnt;
};

// ---------------------------------------------------------------- aabb ----
// aabb.h:61-85.
bool aabb_hit(const rtk_aabb& b, const Ray& r, double tmin, double tmax) {
    const double lo[3] = {b.xmin, b.ymin, b.zmin}, hi[3] = {b.xmax, b.ymax, b.zmax};
    const double o[3] = {r.o.x, r.o.y, r.o.z}, d[3] = {r.d.x, r.d.y, r.d.z};
    for (int axis = 0; axis < 3; axis++) {
        const double adinv = 1.0 / d[axis];
        double t0 = (lo[axis] - o[axis]) * adinv;
        double t1 = (hi[axis] - o[axis]) * adinv;
        if (t0 < t1) {
            if (t0 > tmin) tmin = t0;
            if (t1 < tmax) tmax = t1;
        } else {
            if (t1 > tmin) tmin = t1;
            if (t0 < tmax) tmax = t0;
        }
        if (tmax <= tmin) return false;
    }
    return true;
}

// ---------------------------------------------------------------- sphere --
// sphere.h:32-58, 67-73.
bool sphere_hit(const rtk_sphere& s, const Ray& r, double tmin, double tmax, HitRec& rec, int rank = 0, const Tie& best = Tie{}) {
    V3 current_center = v3(s.center0) + r.tm * v3(s.center_dir);  // center.at(r.time())
    V3 oc = current_center - r.o;
    double a = length_squared(r.d);
    double h = dot(r.d, oc);
    double c = length_squared(oc) - s.radius * s.radius;
    double discriminant = h * h - a * c;
    if (discriminant < 0) return false;
    double sqrtd = std::sqrt(discriminant);
    double root = (h - sqrtd) / a;
    auto accepts = [&](double t) { return (tmin < t && t < tmax) /* interval::surrounds */ || (t == tmax && tmin < t && tie_sphere_wins(rank, best)); };
    if (!accepts(root)) {
        root = (h + sqrtd) / a;
        if (!accepts(root)) return false;
    }
    rec.t = root;
    rec.prim_kind = RTK_NODE_SPHERE;
    rec.rank = rank;
    rec.p = at(r, rec.t);
    V3 outward = (rec.p - current_center) / s.radius;
    set_face_normal(rec, r, outward);
    double theta = std::acos(-outward.y);
    double phi = std::atan2(-outward.z, outward.x) + kPi;
    rec.u = phi / (2 * kPi);
    rec.v = theta / kPi;
    rec.mat = s.material;
    return true;
}

// ---------------------------------------------------------------- quad ----
// quad.h:29-73.
bool quad_hit(const rtk_quad& q, const Ray& r, double tmin, double tmax, HitRec& rec, int rank = 0, const Tie& best = Tie{}) {
    V3 normal = v3(q.normal);
    double denom = dot(normal, r.d);
    if (std::fabs(denom) < 1e-8) return false;
    double t = (q.D - dot(normal, r.o)) / denom;
    if (!(tmin <= t && t <= tmax)) return false;  // interval::contains
    V3 intersection = at(r, t);
    V3 planar = intersection - v3(q.Q);
    double alpha = dot(v3(q.w), cross(planar, v3(q.v)));
    double beta = dot(v3(q.w), cross(v3(q.u), planar));
    if (!(0 <= alpha && alpha <= 1) || !(0 <= beta && beta <= 1)) return false;  // is_interior
    if (t == tmax && !tie_inclusive_wins(rank, best)) return false;                // (tie rule above; never in a reference-order description)
    rec.prim_kind = RTK_NODE_QUAD;
    rec.rank = rank;
    rec.u = alpha;
    rec.v = beta;
    rec.t = t;
    rec.p = intersection;
    rec.mat = q.material;
    set_face_normal(rec, r, normal);
    return true;
}

// ---------------------------------------------------------------- triangle
// triangle.h:65-122.  det, invDet, alpha, beta, gamma are `float` there
// (triangle.h:72,77,96-98) and the UV mix is float arithmetic.
bool triangle_hit(const rtk_triangle& tr, const Ray& r, double tmin, double tmax, HitRec& rec, int rank = 0, const Tie& best = Tie{}) {
    V3 p0 = v3(tr.p0);
    V3 v0v1 = v3(tr.p1) - p0;
    V3 v0v2 = v3(tr.p2) - p0;
    V3 pvec = cross(r.d, v0v2);
    float det = float(dot(v0v1, pvec));
    if (std::fabs(det) < 1e-8) return false;
    float invDet = 1.0f / det;
    V3 tvec = r.o - p0;
    double u = dot(tvec, pvec) * invDet;
    if (u < 0.0f || u > 1.0f) return false;
    V3 qvec = cross(tvec, v0v1);
    double v = dot(r.d, qvec) * invDet;
    if (v < 0.0f || u + v > 1.0f) return false;
    double t = dot(v0v2, qvec) * invDet;
    if (t < tmin || t > tmax) return false;
    V3 intersection = at(r, t);
    float alpha = float(1 - u - v);
    float beta = float(u);
    float gamma = float(v);
    double da = alpha, db = beta;
    if (!(0 <= da && da <= 1) || !(0 <= db && db <= 1)) return false;  // is_interior(alpha, beta)
    if (t == tmax && !tie_inclusive_wins(rank, best)) return false;     // (tie rule above)
    rec.prim_kind = RTK_NODE_TRIANGLE;
    rec.rank = rank;
    rec.u = alpha * tr.uv0[0] + beta * tr.uv1[0] + gamma * tr.uv2[0];
    rec.v = alpha * tr.uv0[1] + beta * tr.uv1[1] + gamma * tr.uv2[1];
    rec.t = t;
    rec.p = intersection;
    rec.mat = tr.material;
    set_face_normal(rec, r, v3(tr.normal));
    return true;
}

// `best` travels with tmax: the primitive whose hit set it (Tie{} when tmax is the caller's own bound).
bool node_hit(Ctx& cx, int node, const Ray& r, double tmin, double tmax, HitRec& rec, const Tie& best = Tie{});

// hittable_list.h:22-35.
bool list_hit(Ctx& cx, const rtk_node& n, const Ray& r, double tmin, double tmax, HitRec& rec, const Tie& best) {
    HitRec temp_rec;
    bool hit_anything = false;
    double closest_so_far = tmax;
    Tie closest = best;
    for (int i = 0; i < n.b; i++) {
        if (node_hit(cx, cx.sc->list_children[n.a + i], r, tmin, closest_so_far, temp_rec, closest)) {
            hit_anything = true;
            closest_so_far = temp_rec.t;
            closest = tie_of(temp_rec);
            rec = temp_rec;
        }
    }
    return hit_anything;
}

// bvh.h:64-72.
bool bvh_hit(Ctx& cx, const rtk_node& n, const Ray& r, double tmin, double tmax, HitRec& rec, const Tie& best) {
    cx.cnt.box_tests++;
    if (!aabb_hit(cx.sc->bvh_boxes[n.c], r, tmin, tmax)) return false;
    bool hit_left = node_hit(cx, n.a, r, tmin, tmax, rec, best);
    bool hit_right = node_hit(cx, n.b, r, tmin, hit_left ? rec.t : tmax, rec, hit_left ? tie_of(rec) : best);
    return hit_left || hit_right;
}

// hittable.h:46-58.
bool translate_hit(Ctx& cx, const rtk_node& n, const Ray& r, double tmin, double tmax, HitRec& rec, const Tie& best) {
    V3 offset = v3(cx.sc->translates[n.a].offset);
    Ray moved{r.o - offset, r.d, r.tm};
    if (!node_hit(cx, n.b, moved, tmin, tmax, rec, best)) return false;
    rec.p = rec.p + offset;
    return true;
}

// hittable.h:101-139.
bool rotate_y_hit(Ctx& cx, const rtk_node& n, const Ray& r, double tmin, double tmax, HitRec& rec, const Tie& best) {
    const double s = cx.sc->rotates[n.a].sin_theta, c = cx.sc->rotates[n.a].cos_theta;
    V3 origin = v3((c * r.o.x) - (s * r.o.z), r.o.y, (s * r.o.x) + (c * r.o.z));
    V3 direction = v3((c * r.d.x) - (s * r.d.z), r.d.y, (s * r.d.x) + (c * r.d.z));
    Ray rotated{origin, direction, r.tm};
    if (!node_hit(cx, n.b, rotated, tmin, tmax, rec, best)) return false;
    rec.p = v3((c * rec.p.x) + (s * rec.p.z), rec.p.y, (-s * rec.p.x) + (c * rec.p.z));
    rec.normal = v3((c * rec.normal.x) + (s * rec.normal.z), rec.normal.y, (-s * rec.normal.x) + (c * rec.normal.z));
    return true;
}

// constant_medium.h:20-53.  One RNG draw per call that reaches line 40.
bool medium_hit(Ctx& cx, const rtk_node& n, const Ray& r, double tmin, double tmax, HitRec& rec) {
    const rtk_medium& m = cx.sc->media[n.a];
    HitRec rec1, rec2;
    if (!node_hit(cx, n.b, r, -kInf, kInf, rec1)) return false;
    if (!node_hit(cx, n.b, r, rec1.t + 0.0001, kInf, rec2)) return false;
    if (rec1.t < tmin) rec1.t = tmin;
    if (rec2.t > tmax) rec2.t = tmax;
    if (rec1.t >= rec2.t) return false;
    if (rec1.t < 0) rec1.t = 0;
    double ray_length = length(r.d);
    double distance_inside_boundary = (rec2.t - rec1.t) * ray_length;
    double hit_distance = m.neg_inv_density * std::log(rnd(cx.rng));
    if (hit_distance > distance_inside_boundary) return false;
    rec.t = rec1.t + hit_distance / ray_length;
    rec.prim_kind = RTK_NODE_MEDIUM;
    rec.rank = 0;
    rec.p = at(r, rec.t);
    rec.normal = v3(1, 0, 0);
    rec.front_face = true;
    rec.mat = m.material;
    return true;
}

bool node_hit(Ctx& cx, int node, const Ray& r, double tmin, double tmax, HitRec& rec, const Tie& best) {
    const rtk_node& n = cx.sc->nodes[node];
    switch (n.kind) {
        case RTK_NODE_SPHERE: cx.cnt.sphere_tests++; return sphere_hit(cx.sc->spheres[n.a], r, tmin, tmax, rec, n.c, best);
        case RTK_NODE_QUAD: cx.cnt.quad_tests++; return quad_hit(cx.sc->quads[n.a], r, tmin, tmax, rec, n.c, best);
        case RTK_NODE_TRIANGLE: cx.cnt.triangle_tests++; return triangle_hit(cx.sc->triangles[n.a], r, tmin, tmax, rec, n.c, best);
        case RTK_NODE_LIST: return list_hit(cx, n, r, tmin, tmax, rec, best);
        case RTK_NODE_BVH: return bvh_hit(cx, n, r, tmin, tmax, rec, best);
        case RTK_NODE_TRANSLATE: cx.cnt.xform_enters++; return translate_hit(cx, n, r, tmin, tmax, rec, best);
        case RTK_NODE_ROTATE_Y: cx.cnt.xform_enters++; return rotate_y_hit(cx, n, r, tmin, tmax, rec, best);
        case RTK_NODE_MEDIUM: cx.cnt.medium_tests++; return medium_hit(cx, n, r, tmin, tmax, rec);
    }
    return false;
}

// ---------------------------------------------------------------- perlin --
// perlin.h:14-37, 72-89.
double perlin_noise(Ctx& cx, const rtk_perlin& pn, V3 p) {
    cx.cnt.noise_calls++;
    double u = p.x - std::floor(p.x);
    double v = p.y - std::floor(p.y);
    double w = p.z - std::floor(p.z);
    int i = int(std::floor(p.x));
    int j = int(std::floor(p.y));
    int k = int(std::floor(p.z));
    V3 c[2][2][2];
    for (int di = 0; di < 2; di++)
        for (int dj = 0; dj < 2; dj++)
            for (int dk = 0; dk < 2; dk++) {
                int idx = pn.perm_x[(i + di) & 255] ^ pn.perm_y[(j + dj) & 255] ^ pn.perm_z[(k + dk) & 255];
                c[di][dj][dk] = v3(pn.randvec[idx][0], pn.randvec[idx][1], pn.randvec[idx][2]);
            }
    double uu = u * u * (3 - 2 * u);
    double vv = v * v * (3 - 2 * v);
    double ww = w * w * (3 - 2 * w);
    double accum = 0.0;
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++)
            for (int cc = 0; cc < 2; cc++) {
                V3 weight_v = v3(u - a, v - b, w - cc);
                accum += (a * uu + (1 - a) * (1 - uu)) * (b * vv + (1 - b) * (1 - vv)) * (cc * ww + (1 - cc) * (1 - ww)) *
                         dot(c[a][b][cc], weight_v);
            }
    return accum;
}
// perlin.h:38-50.
double perlin_turb(Ctx& cx, const rtk_perlin& pn, V3 p, int depth) {
    double accum = 0.0;
    V3 temp_p = p;
    double weight = 1.0;
    for (int i = 0; i < depth; i++) {
        accum += weight * perlin_noise(cx, pn, temp_p);
        weight *= 0.5;
        temp_p = v3(temp_p.x * 2, temp_p.y * 2, temp_p.z * 2);
    }
    return std::fabs(accum);
}

// ---------------------------------------------------------------- texture -
// texture.h:20-120.
V3 texture_value(Ctx& cx, int tex, double u, double v, V3 p) {
    const rtk_texture& t = cx.sc->textures[tex];
    switch (t.kind) {
        case RTK_TEX_SOLID: return v3(t.color);
        case RTK_TEX_CHECKER: {
            int xi = int(std::floor(t.param * p.x));
            int yi = int(std::floor(t.param * p.y));
            int zi = int(std::floor(t.param * p.z));
            bool is_even = (xi + yi + zi) % 2 == 0;
            return texture_value(cx, is_even ? t.even : t.odd, u, v, p);
        }
        case RTK_TEX_CHECKER_TRI: {
            v = 1.0 - v;
            int ui = int(std::round(t.param * u * 10));
            int vi = int(std::round(t.param * v * 10));
            bool is_even = (ui + vi) % 2 == 0;
            return texture_value(cx, is_even ? t.even : t.odd, u, v, p);
        }
        case RTK_TEX_IMAGE: {
            const rtk_image& im = cx.sc->images[t.image];
            if (im.width <= 0 || im.height <= 0) return v3(0, 1, 1);
            u = u < 0 ? 0 : (u > 1 ? 1 : u);
            double vc = v < 0 ? 0 : (v > 1 ? 1 : v);
            v = 1.0 - vc;
            int i = int(u * im.width);
            int j = int(v * im.height);
            // rtw_stb_image.h:71-81,92-97
            i = i < 0 ? 0 : (i < im.width ? i : im.width - 1);
            j = j < 0 ? 0 : (j < im.height ? j : im.height - 1);
            cx.cnt.texel_fetches++;
            const uint8_t* px = cx.sc->texels + im.texel_offset + (int64_t(j) * im.width + i) * 3;
            double color_scale = 1.0 / 255.0;
            return v3(color_scale * px[0], color_scale * px[1], color_scale * px[2]);
        }
        case RTK_TEX_NOISE: {
            double s = 1 + std::sin(t.param * p.z + 10 * perlin_turb(cx, cx.sc->perlins[t.image], p, 7));
            return s * v3(.5, .5, .5);
        }
    }
    return v3(0, 0, 0);
}

// ---------------------------------------------------------------- material
// material.h:14-16, 99-101, 111-113.
V3 material_emitted(Ctx& cx, int mat, double u, double v, V3 p) {
    const rtk_material& m = cx.sc->materials[mat];
    if (m.kind == RTK_MAT_DIFFUSE_LIGHT) return texture_value(cx, m.texture, u, v, p);
    return v3(0, 0, 0);
}

double reflectance(double cosine, double refraction_index) {  // material.h:69-74
    double r0 = (1 - refraction_index) / (1 + refraction_index);
    r0 = r0 * r0;
    return r0 + (1 - r0) * std::pow((1 - cosine), 5);
}

bool material_scatter(Ctx& cx, int mat, const Ray& r_in, const HitRec& rec, V3& attenuation, Ray& scattered) {
    const rtk_material& m = cx.sc->materials[mat];
    switch (m.kind) {
        case RTK_MAT_LAMBERTIAN: {  // material.h:29-38
            V3 dir = rec.normal + random_unit_vector(cx.rng);
            if (near_zero(dir)) dir = rec.normal;
            scattered = Ray{rec.p, dir, r_in.tm};
            attenuation = texture_value(cx, m.texture, rec.u, rec.v, rec.p);
            return true;
        }
        case RTK_MAT_METAL: {  // material.h:82-88
            V3 reflected = reflect(r_in.d, rec.normal);
            V3 fuzzed = m.param * random_unit_vector(cx.rng);
            reflected = unit_vector(reflected) + fuzzed;
            scattered = Ray{rec.p, reflected, r_in.tm};
            attenuation = v3(m.albedo);
            return dot(scattered.d, rec.normal) > 0;
        }
        case RTK_MAT_DIELECTRIC: {  // material.h:47-65
            attenuation = v3(1.0, 1.0, 1.0);
            double ri = rec.front_face ? (1.0 / m.param) : m.param;
            V3 unit_direction = unit_vector(r_in.d);
            double cos_theta = std::fmin(dot(-unit_direction, rec.normal), 1.0);
            double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
            bool cannot_refract = ri * sin_theta > 1.0;
            V3 direction;
            if (cannot_refract || reflectance(cos_theta, ri) > rnd(cx.rng))
                direction = reflect(unit_direction, rec.normal);
            else
                direction = refract(unit_direction, rec.normal, ri);
            scattered = Ray{rec.p, direction, r_in.tm};
            return true;
        }
        case RTK_MAT_DIFFUSE_LIGHT: return false;  // material.h:18-20, 116-118
        case RTK_MAT_ISOTROPIC: {                  // material.h:129-134
            scattered = Ray{rec.p, random_unit_vector(cx.rng), r_in.tm};
            attenuation = texture_value(cx, m.texture, rec.u, rec.v, rec.p);
            return true;
        }
        case RTK_MAT_SPECULAR: {  // material.h:145-167
            V3 reflected = reflect(unit_vector(r_in.d), rec.normal);
            V3 diffuse = random_on_hemisphere(cx.rng, rec.normal);
            double reflection_factor = std::pow(1.0 - dot(reflected, unit_vector(r_in.d)), m.param);
            V3 dir = reflection_factor * reflected + (1.0 - reflection_factor) * diffuse;
            if (near_zero(dir)) dir = rec.normal;
            scattered = Ray{rec.p, dir, r_in.tm};
            attenuation = v3(m.albedo);
            return true;
        }
    }
    return false;
}

// ---------------------------------------------------------------- camera --
// Camera.txt:240-272 (no shadow ray; `max` is the windows.h macro there).
V3 get_lighting(Ctx& cx, V3 p, V3 normal) {
    V3 result = v3(0, 0, 0);
    for (int i = 0; i < cx.sc->n_lights; i++) {
        const rtk_point_light& light = cx.sc->lights[i];
        V3 light_dir = v3(light.position) - p;
        double distance_squared = length_squared(light_dir);
        light_dir = unit_vector(light_dir);
        double d = dot(normal, light_dir);
        double diffuse = (d > 0.0) ? d : 0.0;
        double size_factor = light.size;
        double radius_effect = size_factor * 0.1;
        if (distance_squared <= size_factor * size_factor) {
            result = result + diffuse * v3(light.intensity);
        } else {
            double attenuation = 1.0 / (distance_squared + radius_effect);
            V3 intensity = attenuation * v3(light.intensity);
            result = result + diffuse * intensity;
        }
    }
    return result;
}

// Camera.txt:203-238.
V3 ray_color(Ctx& cx, const rtk_camera& cam, const Ray& r, int depth) {
    if (depth <= 0) return v3(0, 0, 0);
    HitRec rec;
    cx.cnt.segments++;
    if (!node_hit(cx, cx.sc->root, r, 0.001, kInf, rec)) return v3(cam.background);
    cx.cnt.surface_hits++;
    V3 color_from_emission = material_emitted(cx, rec.mat, rec.u, rec.v, rec.p);
    Ray scattered;
    V3 attenuation;
    if (!material_scatter(cx, rec.mat, r, rec, attenuation, scattered)) return color_from_emission;
    V3 lighting = attenuation * get_lighting(cx, rec.p, rec.normal);
    V3 color_from_scatter = attenuation * ray_color(cx, cam, scattered, depth - 1);
    return color_from_emission + lighting + color_from_scatter;
}

// Camera.txt:177-200.  sample_square draws y first (g++ argument order).
Ray get_ray(Ctx& cx, const rtk_camera& cam, int i, int j) {
    double oy = rnd(cx.rng) - 0.5, ox = rnd(cx.rng) - 0.5;
    V3 pixel_sample = v3(cam.pixel00_loc) + ((i + ox) * v3(cam.pixel_delta_u)) + ((j + oy) * v3(cam.pixel_delta_v));
    V3 origin;
    if (cam.defocus_angle <= 0) {
        origin = v3(cam.center);
    } else {
        V3 p = random_in_unit_disk(cx.rng);
        origin = v3(cam.center) + (p.x * v3(cam.defocus_disk_u)) + (p.y * v3(cam.defocus_disk_v));
    }
    V3 direction = pixel_sample - origin;
    double ray_time = rnd(cx.rng);
    return Ray{origin, direction, ray_time};
}

inline double linear_to_gamma(double x) { return x > 0 ? std::sqrt(x) : 0; }  // Camera.txt:29-34
inline double clamp_intensity(double x) { return x < 0.000 ? 0.000 : (x > 0.999 ? 0.999 : x); }

void add_counters(rtk_work_counters& a, const rtk_work_counters& b) {
    uint64_t* pa = reinterpret_cast<uint64_t*>(&a);
    const uint64_t* pb = reinterpret_cast<const uint64_t*>(&b);
    for (size_t k = 0; k < sizeof(rtk_work_counters) / sizeof(uint64_t); k++) pa[k] += pb[k];
}

// Camera.txt:65-93 for rows [j0, j1).
void render_rows(const rtk_scene_desc* sc, const rtk_camera* cam, uint32_t seed, int j0, int j1, double* linear, uint8_t* rgb8,
                 rtk_work_counters* out) {
    Ctx cx;
    cx.sc = sc;
    std::memset(&cx.cnt, 0, sizeof cx.cnt);
    const int W = cam->image_width;
    for (int j = j0; j < j1; ++j) {
        for (int i = 0; i < W; ++i) {
            V3 pixel_color = v3(0, 0, 0);
            for (int sample = 0; sample < cam->samples_per_pixel; ++sample) {
                rng_seed(cx.rng, seed, uint32_t(j * W + i), uint32_t(sample));
                cx.rng.draws = 0;
                Ray r = get_ray(cx, *cam, i, j);
                pixel_color = pixel_color + ray_color(cx, *cam, r, cam->max_depth);
                cx.cnt.samples++;
                cx.cnt.rng_draws += cx.rng.draws;
            }
            pixel_color = cam->pixel_samples_scale * pixel_color;
            size_t idx = (size_t(j) * W + i) * 3;
            if (linear) {
                linear[idx] = pixel_color.x;
                linear[idx + 1] = pixel_color.y;
                linear[idx + 2] = pixel_color.z;
            }
            if (rgb8) {
                rgb8[idx] = uint8_t(int(255.999 * clamp_intensity(linear_to_gamma(pixel_color.x))));
                rgb8[idx + 1] = uint8_t(int(255.999 * clamp_intensity(linear_to_gamma(pixel_color.y))));
                rgb8[idx + 2] = uint8_t(int(255.999 * clamp_intensity(linear_to_gamma(pixel_color.z))));
            }
        }
    }
    *out = cx.cnt;
}

}  // namespace

extern "C" {

// Whole-image render.  threads <= 0 -> hardware_concurrency.  Rows are dealt to
// threads in contiguous blocks exactly as Camera.txt:59-61,96-100 (last thread
// takes the remainder); because the RNG is keyed by (pixel, sample) the result
// does not depend on the thread count.
int orc_render(const rtk_scene_desc* sc, const rtk_camera* cam, uint32_t seed, int threads, double* linear, uint8_t* rgb8,
               rtk_work_counters* counters) {
    if (!sc || !cam || sc->root < 0 || sc->root >= sc->n_nodes) return -1;
    int H = cam->image_height;
    int nt = threads > 0 ? threads : int(std::thread::hardware_concurrency());
    if (nt < 1) nt = 1;
    if (nt > H) nt = H;
    std::vector<rtk_work_counters> parts(nt);
    std::vector<std::thread> pool;
    int rows_per_thread = H / nt;
    for (int t = 0; t < nt; t++) {
        int j0 = t * rows_per_thread;
        int j1 = (t == nt - 1) ? H : j0 + rows_per_thread;
        pool.emplace_back(render_rows, sc, cam, seed, j0, j1, linear, rgb8, &parts[t]);
    }
    for (auto& th : pool) th.join();
    rtk_work_counters total;
    std::memset(&total, 0, sizeof total);
    for (auto& p : parts) add_counters(total, p);
    if (counters) *counters = total;
    return 0;
}

// One sample of one pixel: radiance + the number of RNG draws it consumed.
int orc_sample(const rtk_scene_desc* sc, const rtk_camera* cam, uint32_t seed, int i, int j, int sample, double rgb[3],
               uint64_t* draws) {
    Ctx cx;
    cx.sc = sc;
    std::memset(&cx.cnt, 0, sizeof cx.cnt);
    rng_seed(cx.rng, seed, uint32_t(j * cam->image_width + i), uint32_t(sample));
    cx.rng.draws = 0;
    Ray r = get_ray(cx, *cam, i, j);
    V3 c = ray_color(cx, *cam, r, cam->max_depth);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
    if (draws) *draws = cx.rng.draws;
    return 0;
}

// The first n uniforms of stream (seed, pixel, sample) -- lets tests pin the
// generator itself against the device and against ref_driver's rand().
void orc_rng_stream(uint32_t seed, uint32_t pixel, uint32_t sample, int n, double* out) {
    Rng g;
    rng_seed(g, seed, pixel, sample);
    for (int k = 0; k < n; k++) out[k] = rnd(g);
}

// ---- function-level entry points for the known-answer tests ---------------
// Each takes scripted uniforms where the reference function draws: the script
// is consumed in call order (the KAT files record the uniforms ref_driver's
// interposed rand() handed out).

struct ScriptRng {
    const double* u;
    int n, i;
};

int orc_kat_aabb(const rtk_aabb* box, const double ray_od[6], double tmin, double tmax) {
    Ray r{v3(ray_od[0], ray_od[1], ray_od[2]), v3(ray_od[3], ray_od[4], ray_od[5]), 0};
    return aabb_hit(*box, r, tmin, tmax) ? 1 : 0;
}

// out[11] = t, p(3), normal(3), front_face, u, v, mat
static void pack_rec(const HitRec& rec, double* out) {
    out[0] = rec.t;
    out[1] = rec.p.x; out[2] = rec.p.y; out[3] = rec.p.z;
    out[4] = rec.normal.x; out[5] = rec.normal.y; out[6] = rec.normal.z;
    out[7] = rec.front_face ? 1 : 0;
    out[8] = rec.u; out[9] = rec.v;
    out[10] = rec.mat;
}

// Closest hit of scene node `node` for one ray; seeds the RNG stream with
// (seed, pixel, sample) first (only media draw).  Returns hit flag.
int orc_kat_node_hit(const rtk_scene_desc* sc, int node, const double ray_odt[7], double tmin, double tmax, uint32_t seed,
                     uint32_t pixel, uint32_t sample, double out[11], uint64_t* draws) {
    Ctx cx;
    cx.sc = sc;
    std::memset(&cx.cnt, 0, sizeof cx.cnt);
    rng_seed(cx.rng, seed, pixel, sample);
    cx.rng.draws = 0;
    Ray r{v3(ray_odt[0], ray_odt[1], ray_odt[2]), v3(ray_odt[3], ray_odt[4], ray_odt[5]), ray_odt[6]};
    HitRec rec;
    bool h = node_hit(cx, node, r, tmin, tmax, rec);
    pack_rec(rec, out);
    if (draws) *draws = cx.rng.draws;
    return h ? 1 : 0;
}

// material::scatter + emitted for a given hit record.
// rec_in[11] as pack_rec; out[10] = scattered o(3), d(3), attenuation(3), tm ; emitted[3]
int orc_kat_scatter(const rtk_scene_desc* sc, int mat, const double ray_odt[7], const double rec_in[11], uint32_t seed,
                    uint32_t pixel, uint32_t sample, double out[10], double emitted[3], uint64_t* draws) {
    Ctx cx;
    cx.sc = sc;
    std::memset(&cx.cnt, 0, sizeof cx.cnt);
    rng_seed(cx.rng, seed, pixel, sample);
    cx.rng.draws = 0;
    Ray r{v3(ray_odt[0], ray_odt[1], ray_odt[2]), v3(ray_odt[3], ray_odt[4], ray_odt[5]), ray_odt[6]};
    HitRec rec;
    rec.t = rec_in[0];
    rec.p = v3(rec_in[1], rec_in[2], rec_in[3]);
    rec.normal = v3(rec_in[4], rec_in[5], rec_in[6]);
    rec.front_face = rec_in[7] != 0;
    rec.u = rec_in[8];
    rec.v = rec_in[9];
    rec.mat = mat;
    V3 e = material_emitted(cx, mat, rec.u, rec.v, rec.p);
    emitted[0] = e.x; emitted[1] = e.y; emitted[2] = e.z;
    V3 att = v3(0, 0, 0);
    Ray sc_ray{v3(0, 0, 0), v3(0, 0, 0), 0};
    bool ok = material_scatter(cx, mat, r, rec, att, sc_ray);
    out[0] = sc_ray.o.x; out[1] = sc_ray.o.y; out[2] = sc_ray.o.z;
    out[3] = sc_ray.d.x; out[4] = sc_ray.d.y; out[5] = sc_ray.d.z;
    out[6] = att.x; out[7] = att.y; out[8] = att.z;
    out[9] = sc_ray.tm;
    if (draws) *draws = cx.rng.draws;
    return ok ? 1 : 0;
}

void orc_kat_texture(const rtk_scene_desc* sc, int tex, double u, double v, const double p[3], double out[3]) {
    Ctx cx;
    cx.sc = sc;
    std::memset(&cx.cnt, 0, sizeof cx.cnt);
    V3 c = texture_value(cx, tex, u, v, v3(p[0], p[1], p[2]));
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

// Camera.txt:177-191 for one (pixel, sample): out[7] = origin, direction, time.
void orc_kat_get_ray(const rtk_camera* cam, uint32_t seed, int i, int j, int sample, double out[7], uint64_t* draws) {
    Ctx cx;
    cx.sc = nullptr;
    rng_seed(cx.rng, seed, uint32_t(j * cam->image_width + i), uint32_t(sample));
    cx.rng.draws = 0;
    Ray r = get_ray(cx, *cam, i, j);
    out[0] = r.o.x; out[1] = r.o.y; out[2] = r.o.z;
    out[3] = r.d.x; out[4] = r.d.y; out[5] = r.d.z;
    out[6] = r.tm;
    if (draws) *draws = cx.rng.draws;
}

}  // extern "C"
