// rtk_math.h -- host-side scalar/vector vocabulary of the scene API.
//
// Mirrors the *interface* of the reference's rtweekend.h / vec3.h / ray.h /
// interval.h / aabb.h so that scene-setup code written against the reference
// (main.cpp:128-442) compiles unchanged.  These types are used for scene
// CONSTRUCTION only (bounding boxes, BVH split, quad/triangle plane constants);
// no ray is ever traced on the host -- the sample loop lives in csrc/*.hip.
//
// Numerical contract: every derived value that ends up in rtk_scene_desc is
// computed with the same operation order as the reference, because the device
// consumes it verbatim and parity is checked bit-for-bit against the
// reference's own classes (oracle/_ref).  Notable cases are called out inline.
#ifndef RTK_MATH_H
#define RTK_MATH_H

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <limits>
#include <memory>

using std::make_shared;
using std::shared_ptr;

const double infinity = std::numeric_limits<double>::infinity();
const double pi = 3.1415926535897932385;

inline double degrees_to_radians(double degrees) { return degrees * pi / 180.0; }

// ---------------------------------------------------------------------------
// Scene-construction RNG.  The reference draws from the process-global
// std::rand() (rtweekend.h:26-29), which is neither reproducible nor
// thread-safe.  Here the same call sites draw from one sequential
// PCG-RXS-M-XS-32 stream with 24-bit uniforms; oracle/_ref interposes rand()
// with the identical generator, so random scenes and Perlin tables match the
// reference's own constructors value for value.
// ---------------------------------------------------------------------------
namespace rtk {
struct host_rng_state { uint32_t s = 0x5EED2025u; uint64_t draws = 0; };
inline host_rng_state& host_rng() { static host_rng_state g; return g; }
inline void seed_scene_rng(uint32_t seed) { host_rng().s = seed; host_rng().draws = 0; }
inline uint32_t host_rng_next_u24() {
    host_rng_state& g = host_rng();
    uint32_t old = g.s;
    g.s = old * 747796405u + 2891336453u;
    uint32_t word = ((old >> ((old >> 28u) + 4u)) ^ old) * 277803737u;
    g.draws++;
    return ((word >> 22u) ^ word) >> 8;
}
}  // namespace rtk

inline double random_double() { return rtk::host_rng_next_u24() * (1.0 / 16777216.0); }
inline double random_double(double min, double max) { return min + (max - min) * random_double(); }
inline int random_int(int min, int max) { return int(random_double(min, max + 1)); }

// ---------------------------------------------------------------------------
class vec3 {
public:
    double e[3];

    vec3() : e{0, 0, 0} {}
    vec3(double a, double b, double c) : e{a, b, c} {}

    double x() const { return e[0]; }
    double y() const { return e[1]; }
    double z() const { return e[2]; }
    double operator[](int i) const { return e[i]; }
    double& operator[](int i) { return e[i]; }

    vec3 operator-() const { return vec3(-e[0], -e[1], -e[2]); }
    vec3& operator+=(const vec3& o) { e[0] += o.e[0]; e[1] += o.e[1]; e[2] += o.e[2]; return *this; }
    vec3& operator*=(double t) { e[0] *= t; e[1] *= t; e[2] *= t; return *this; }
    // The reference divides by multiplying with the reciprocal (vec3.h:39-41,
    // 91-93); x * (1/t) != x / t in the last bit, so this is kept.
    vec3& operator/=(double t) { return *this *= 1 / t; }
    bool operator==(const vec3& o) const { return e[0] == o.e[0] && e[1] == o.e[1] && e[2] == o.e[2]; }

    // vec3.h:47 writes pow(e,2); g++ folds that to e*e.  Sum order (x+y)+z.
    double length_squared() const { return e[0] * e[0] + e[1] * e[1] + e[2] * e[2]; }
    double length() const { return std::sqrt(length_squared()); }
    bool near_zero() const {
        const double s = 1e-8;
        return std::fabs(e[0]) < s && std::fabs(e[1]) < s && std::fabs(e[2]) < s;
    }

    // vec3.h:50-56 builds vec3(random_double(), random_double(), random_double());
    // g++ evaluates constructor arguments right to left, so the FIRST draw lands
    // in z.  Spelled out here so the behaviour does not depend on the compiler.
    static vec3 random() {
        double c = random_double(), b = random_double(), a = random_double();
        return vec3(a, b, c);
    }
    static vec3 random(double lo, double hi) {
        double c = random_double(lo, hi), b = random_double(lo, hi), a = random_double(lo, hi);
        return vec3(a, b, c);
    }
};

using point3 = vec3;
using color = vec3;

inline std::ostream& operator<<(std::ostream& os, const vec3& v) { return os << v.e[0] << ' ' << v.e[1] << ' ' << v.e[2]; }
inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }
inline vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }
inline vec3 operator*(const vec3& a, const vec3& b) { return vec3(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); }
inline vec3 operator*(double t, const vec3& v) { return vec3(t * v.e[0], t * v.e[1], t * v.e[2]); }
inline vec3 operator*(const vec3& v, double t) { return t * v; }
inline vec3 operator/(const vec3& v, double t) { return (1 / t) * v; }
inline double dot(const vec3& a, const vec3& b) { return a.e[0] * b.e[0] + a.e[1] * b.e[1] + a.e[2] * b.e[2]; }
inline vec3 cross(const vec3& a, const vec3& b) {
    return vec3(a.e[1] * b.e[2] - a.e[2] * b.e[1], a.e[2] * b.e[0] - a.e[0] * b.e[2], a.e[0] * b.e[1] - a.e[1] * b.e[0]);
}
inline vec3 unit_vector(const vec3& v) { return v / v.length(); }

// vec3.h:107-115.  The rejection test `1e-160 < lensq <= 1` is always true
// (SURVEY Q1): a normalised point of the cube, exactly three draws.
inline vec3 random_unit_vector() {
    vec3 p = vec3::random(-1, 1);
    return p / std::sqrt(p.length_squared());
}
inline vec3 random_on_hemisphere(const vec3& normal) {
    vec3 s = random_unit_vector();
    return dot(s, normal) > 0.0 ? s : -s;
}
inline vec3 reflect(const vec3& v, const vec3& n) { return v - 2 * dot(v, n) * n; }
inline vec3 refract(const vec3& uv, const vec3& n, double etai_over_etat) {
    double cos_theta = std::fmin(dot(-uv, n), 1.0);
    vec3 perp = etai_over_etat * (uv + cos_theta * n);
    vec3 par = -std::sqrt(std::fabs(1.0 - perp.length_squared())) * n;
    return perp + par;
}
// vec3.h:135-142: vec3(random_double(-1,1), random_double(-1,1), 0) -- y is drawn first.
inline vec3 random_in_unit_disk() {
    for (;;) {
        double b = random_double(-1, 1), a = random_double(-1, 1);
        vec3 p(a, b, 0);
        if (p.length_squared() < 1) return p;
    }
}
inline vec3 min_point(vec3 a, vec3 b, vec3 c) {
    return vec3(fmin(a.x(), fmin(b.x(), c.x())), fmin(a.y(), fmin(b.y(), c.y())), fmin(a.z(), fmin(b.z(), c.z())));
}
inline vec3 max_point(vec3 a, vec3 b, vec3 c) {
    return vec3(fmax(a.x(), fmax(b.x(), c.x())), fmax(a.y(), fmax(b.y(), c.y())), fmax(a.z(), fmax(b.z(), c.z())));
}

// ---------------------------------------------------------------------------
class ray {
public:
    ray() : tm(0) {}
    ray(const point3& o, const vec3& d, double time) : orig(o), dir(d), tm(time) {}
    ray(const point3& o, const vec3& d) : ray(o, d, 0) {}
    const point3& origin() const { return orig; }
    const vec3& direction() const { return dir; }
    double time() const { return tm; }
    point3 at(double t) const { return orig + t * dir; }

private:
    point3 orig;
    vec3 dir;
    double tm;
};

// ---------------------------------------------------------------------------
class interval {
public:
    double min, max;
    interval() : min(+infinity), max(-infinity) {}
    interval(double lo, double hi) : min(lo), max(hi) {}
    interval(const interval& a, const interval& b) {
        min = a.min <= b.min ? a.min : b.min;
        max = a.max >= b.max ? a.max : b.max;
    }
    double size() const { return max - min; }
    bool contains(double x) const { return min <= x && x <= max; }
    bool surrounds(double x) const { return min < x && x < max; }
    double clamp(double x) const { return x < min ? min : (x > max ? max : x); }
    interval expand(double delta) const {
        double pad = delta / 2;
        return interval(min - pad, max + pad);
    }
    static const interval empty, universe;
};
inline const interval interval::empty = interval(+infinity, -infinity);
inline const interval interval::universe = interval(-infinity, +infinity);
inline interval operator+(const interval& iv, double d) { return interval(iv.min + d, iv.max + d); }
inline interval operator+(double d, const interval& iv) { return iv + d; }

// ---------------------------------------------------------------------------
// Bounding boxes are host-only except for bvh_node boxes, which are the only
// boxes the reference ever slab-tests (bvh.h:65).
class aabb {
public:
    interval x, y, z;
    aabb() {}
    // interval ctor and the merging ctor pad each axis to >= 1e-4 (aabb.h:16-19,
    // 47-53); the two-point ctor does NOT (aabb.h:21-45).
    aabb(const interval& ix, const interval& iy, const interval& iz) : x(ix), y(iy), z(iz) { pad_to_minimums(); }
    aabb(const point3& a, const point3& b) {
        x = a[0] <= b[0] ? interval(a[0], b[0]) : interval(b[0], a[0]);
        y = a[1] <= b[1] ? interval(a[1], b[1]) : interval(b[1], a[1]);
        z = a[2] <= b[2] ? interval(a[2], b[2]) : interval(b[2], a[2]);
    }
    aabb(const aabb& p, const aabb& q) : x(p.x, q.x), y(p.y, q.y), z(p.z, q.z) { pad_to_minimums(); }
    const interval& axis_interval(int n) const { return n == 1 ? y : (n == 2 ? z : x); }
    int longest_axis() const {
        if (x.size() > y.size()) return x.size() > z.size() ? 0 : 2;
        return y.size() > z.size() ? 1 : 2;
    }
    static const aabb empty, universe;

private:
    void pad_to_minimums() {
        const double delta = 0.0001;
        if (x.size() < delta) x = x.expand(delta);
        if (y.size() < delta) y = y.expand(delta);
        if (z.size() < delta) z = z.expand(delta);
    }
};
inline const aabb aabb::empty = aabb(interval::empty, interval::empty, interval::empty);
inline const aabb aabb::universe = aabb(interval::universe, interval::universe, interval::universe);
inline aabb operator+(const aabb& b, const vec3& o) { return aabb(b.x + o.x(), b.y + o.y(), b.z + o.z()); }
inline aabb operator+(const vec3& o, const aabb& b) { return b + o; }

#endif  // RTK_MATH_H
