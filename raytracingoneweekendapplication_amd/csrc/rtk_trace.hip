// rtk_trace.hip -- the per-pixel sample loop of camera::render() as one CDNA4
// (gfx950) kernel family.  Hand-written HIP, wave64, no MFMA (branchy traversal,
// not a contraction), no CUDA idioms.
//
// Replaces, on the device:
//   render_rows λ   Camera.txt:65-93     rtk_render_kernel (one lane = one pixel of an 8x8 tile = one wave)
//   get_ray         Camera.txt:177-200   make_primary_ray
//   ray_color       Camera.txt:203-238   the bounce loop in trace_sample (iterative: radiance = Σ throughput·emission)
//   world.hit       bvh.h:64-72 &c.      closest_hit: executes the linear traversal program (rtk_device_layout.h)
//   *.hit           sphere.h:32-58, quad.h:29-73, triangle.h:65-122, constant_medium.h:20-53
//   scatter/emitted material.h           shade
//   texture::value  texture.h, perlin.h  texture_value, perlin_noise
//
// Parity rules kept in this file (the CPU oracle is compared bit-for-bit where the
// maths allows): same operation order as the reference inside every formula,
// v/t computed as (1/t)*v (vec3.h:91-93), no FMA contraction (the file is built
// with -ffp-contract=off), RNG draws in the reference's order including g++'s
// right-to-left argument evaluation, and the reference's visiting order in the
// BVH (left then right, both always).  The hit record is DEFERRED: traversal
// keeps only (t, winning op); position, normal and uv are computed once per
// segment from the winning primitive, which gives the same values because they
// are pure functions of (ray, t, primitive).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "rtk.h"
#include "rtk_device_layout.h"
#include "rtk_trace.h"

namespace rtk {

// ------------------------------------------------------------------ math -----
template <typename real>
struct V3 {
    real x, y, z;
};
#define RTK_DEV __device__ __forceinline__

template <typename real> RTK_DEV V3<real> mk(real a, real b, real c) { return V3<real>{a, b, c}; }
template <typename real> RTK_DEV V3<real> ld3(const real* p) { return V3<real>{p[0], p[1], p[2]}; }
template <typename real> RTK_DEV V3<real> operator-(V3<real> a) { return V3<real>{-a.x, -a.y, -a.z}; }
template <typename real> RTK_DEV V3<real> operator+(V3<real> a, V3<real> b) { return V3<real>{a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename real> RTK_DEV V3<real> operator-(V3<real> a, V3<real> b) { return V3<real>{a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename real> RTK_DEV V3<real> operator*(V3<real> a, V3<real> b) { return V3<real>{a.x * b.x, a.y * b.y, a.z * b.z}; }
template <typename real> RTK_DEV V3<real> scale(real t, V3<real> a) { return V3<real>{t * a.x, t * a.y, t * a.z}; }
template <typename real> RTK_DEV V3<real> divide(V3<real> a, real t) { return scale(real(1) / t, a); }  // vec3.h:91-93
template <typename real> RTK_DEV real dot(V3<real> a, V3<real> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename real> RTK_DEV V3<real> cross(V3<real> a, V3<real> b) {
    return V3<real>{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename real> RTK_DEV real length_squared(V3<real> a) { return a.x * a.x + a.y * a.y + a.z * a.z; }

RTK_DEV double rt_sqrt(double x) { return __builtin_sqrt(x); }
RTK_DEV float rt_sqrt(float x) { return __builtin_sqrtf(x); }
RTK_DEV double rt_fabs(double x) { return __builtin_fabs(x); }
RTK_DEV float rt_fabs(float x) { return __builtin_fabsf(x); }
RTK_DEV double rt_fmin(double a, double b) { return __builtin_fmin(a, b); }
RTK_DEV float rt_fmin(float a, float b) { return __builtin_fminf(a, b); }
RTK_DEV double rt_fmax(double a, double b) { return __builtin_fmax(a, b); }
RTK_DEV float rt_fmax(float a, float b) { return __builtin_fmaxf(a, b); }
RTK_DEV double rt_floor(double x) { return __builtin_floor(x); }
RTK_DEV float rt_floor(float x) { return __builtin_floorf(x); }
RTK_DEV double rt_round(double x) { return __builtin_round(x); }
RTK_DEV float rt_round(float x) { return __builtin_roundf(x); }
RTK_DEV double rt_pow(double x, double y) { return pow(x, y); }
RTK_DEV float rt_pow(float x, float y) { return powf(x, y); }
RTK_DEV double rt_log(double x) { return log(x); }
RTK_DEV float rt_log(float x) { return logf(x); }
RTK_DEV double rt_sin(double x) { return sin(x); }
RTK_DEV float rt_sin(float x) { return sinf(x); }
RTK_DEV double rt_acos(double x) { return acos(x); }
RTK_DEV float rt_acos(float x) { return acosf(x); }
RTK_DEV double rt_atan2(double y, double x) { return atan2(y, x); }
RTK_DEV float rt_atan2(float y, float x) { return atan2f(y, x); }

template <typename real> RTK_DEV real real_inf() { return real(__builtin_huge_val()); }
// Round 3: libm-heavy f64 code kept OUT OF LINE, one macro each (see sphere_uv below for the reasoning and the first two).
// Measured and left inline: perlin::noise (43.7 vs 34.0 ms) and texture::value as a whole (864 B of scratch).
#ifndef RTK_COLD_POW
#define RTK_COLD_POW 1   // pow (the specular material): C5 38.87 -> 38.25 ms at 32 spp
#endif
#ifndef RTK_COLD_LOG
#define RTK_COLD_LOG 1   // log (constant_medium's scatter distance, twice per segment in book 2): 38.87 -> 35.72; both: 33.98
#endif
#ifndef RTK_COLD_PERLIN
#define RTK_COLD_PERLIN 0
#endif
#ifndef RTK_COLD_TEX
#define RTK_COLD_TEX 0
#endif
#define RTK_NOINLINE __device__ __attribute__((noinline))
#if RTK_COLD_POW
template <typename real> RTK_NOINLINE real call_pow(real x, real y) { return rt_pow(x, y); }
#else
template <typename real> RTK_DEV real call_pow(real x, real y) { return rt_pow(x, y); }
#endif
#if RTK_COLD_LOG
template <typename real> RTK_NOINLINE real call_log(real x) { return rt_log(x); }
#else
template <typename real> RTK_DEV real call_log(real x) { return rt_log(x); }
#endif

// v_min/v_max as single instructions.  __builtin_fmin/fmax are IEEE minnum/maxnum, for which the
// compiler first canonicalises every operand it cannot prove quiet (an extra v_max x,x per use).
// Operands here are results of arithmetic or interval ends, never signalling NaNs; a quiet NaN
// operand is dropped exactly as minnum/maxnum drop it.
RTK_DEV double raw_min(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RTK_DEV double raw_max(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RTK_DEV float raw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RTK_DEV float raw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RTK_DEV float raw_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
RTK_DEV float raw_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
template <typename real> RTK_DEV V3<real> unit_vector(V3<real> a) { return divide(a, rt_sqrt(length_squared(a))); }
template <typename real> RTK_DEV bool near_zero(V3<real> a) {
    const real s = real(1e-8);
    return rt_fabs(a.x) < s && rt_fabs(a.y) < s && rt_fabs(a.z) < s;
}
// x^5 for Schlick's approximation (material.h:73: pow((1 - cosine), 5)).  double: the correctly rounded value, from products
// carried as (value, rounding error) pairs -- x^2 = x2 + e2 exactly (one fused multiply-add recovers the error), x^4 = x4 + e4,
// x^5 = x5 + e5 to ~2^-100, so the final sum rounds like the exact power (0 mismatches against exact rational arithmetic in
// 2e5 arguments, tests/test_schlick_power.py).  float (statistical parity only): three plain products.
RTK_DEV double pow5(double x) {
    const double x2 = x * x, e2 = __builtin_fma(x, x, -x2);
    const double x4 = x2 * x2, e4 = __builtin_fma(x2, x2, -x4) + 2.0 * (x2 * e2);
    const double x5 = x4 * x, e5 = __builtin_fma(x4, x, -x5) + e4 * x;
    return x5 + e5;
}
RTK_DEV float pow5(float x) {
    const float x2 = x * x;
    return x2 * x2 * x;
}
template <typename real> RTK_DEV V3<real> reflect(V3<real> v, V3<real> n) { return v - scale(real(2) * dot(v, n), n); }  // vec3.h:125-127
template <typename real> RTK_DEV V3<real> refract(V3<real> uv, V3<real> n, real eta) {                                  // vec3.h:128-133
    real cos_theta = rt_fmin(dot(-uv, n), real(1));
    V3<real> perp = scale(eta, uv + scale(cos_theta, n));
    V3<real> par = scale(-rt_sqrt(rt_fabs(real(1) - length_squared(perp))), n);
    return perp + par;
}

// ------------------------------------------------------------------ counters --
template <bool COUNT>
struct Counters {
    uint32_t c[C_COUNT];
    RTK_DEV void clear() {
#pragma unroll
        for (int k = 0; k < C_COUNT; k++) c[k] = 0;
    }
    RTK_DEV void inc(int slot, uint32_t n = 1) { c[slot] += n; }
};
template <>
struct Counters<false> {
    RTK_DEV void clear() {}
    RTK_DEV void inc(int, uint32_t = 1) {}
};

// ------------------------------------------------------------------ RNG -------
// Per-lane PCG-RXS-M-XS-32: 32-bit state, one v_mul_lo_u32 for the LCG step and
// one for the output permutation, no 64-bit multiply (gfx950 has no fast one).
// The stream of a sample is seeded by hashing (seed, sample, pixel), so it does
// not depend on lane, tile, launch geometry or GPU count.
RTK_DEV uint32_t pcg_hash(uint32_t v) {
    uint32_t st = v * 747796405u + 2891336453u;
    uint32_t w = ((st >> ((st >> 28u) + 4u)) ^ st) * 277803737u;
    return (w >> 22u) ^ w;
}
template <typename real, bool COUNT>
RTK_DEV real rnd(uint32_t& s, Counters<COUNT>& cnt) {
    uint32_t old = s;
    s = old * 747796405u + 2891336453u;
    uint32_t w = ((old >> ((old >> 28u) + 4u)) ^ old) * 277803737u;
    cnt.inc(C_RNG);
    return real(((w >> 22u) ^ w) >> 8) * real(1.0 / 16777216.0);
}
// random_double(-1, 1) = -1 + 2 * random_double() (rtweekend.h) of the next draw, without a rounding anywhere: a draw is
// v * 2^-24 with v a 24-bit integer, so 2 * draw and -1 + 2 * draw are exact (multiples of 2^-23 in [-1, 1)), and the same
// number is (v - 2^23) * 2^-23 -- an integer subtraction, a conversion and one exact scaling instead of a conversion and
// three floating-point operations.  Same bits in double and in float (24 significant bits suffice), zero included (+0).
template <typename real, bool COUNT>
RTK_DEV real rnd_pm1(uint32_t& s, Counters<COUNT>& cnt) {
    uint32_t old = s;
    s = old * 747796405u + 2891336453u;
    uint32_t w = ((old >> ((old >> 28u) + 4u)) ^ old) * 277803737u;
    cnt.inc(C_RNG);
    return real(int32_t(((w >> 22u) ^ w) >> 8) - 8388608) * real(1.0 / 8388608.0);
}
// vec3.h:107-115 -- always accepts; components drawn z, y, x.
template <typename real, bool COUNT>
RTK_DEV V3<real> random_unit_vector(uint32_t& s, Counters<COUNT>& cnt) {
    real c = rnd_pm1<real>(s, cnt);
    real b = rnd_pm1<real>(s, cnt);
    real a = rnd_pm1<real>(s, cnt);
    V3<real> p = mk(a, b, c);
    return divide(p, rt_sqrt(length_squared(p)));
}

// ------------------------------------------------------------------ rays ------
// Object-space ray for a chain of instance transforms (hittable.h:46-49,101-116).
#ifndef RTK_CHAIN_PREFETCH
#define RTK_CHAIN_PREFETCH 1
#endif
template <uint32_t FEAT>
constexpr bool kChainPrefetch = (FEAT & ~uint32_t(F_FMA_BOX | F_F32_BOX | F_LDS_BOXES)) == kFeatAll || (FEAT & ~uint32_t(F_FMA_BOX | F_F32_BOX | F_MATTE)) == kFeatQuadBox;
// The constants of the first two steps of a chain are requested together, in front of the branches that pick them: walking
// the record step by step costs a dependent read per branch (count, is_rotate[k], that step's constants), and a chain
// switch is nothing but those reads and two dozen operations.  Longer chains (kMaxChain = 4) finish in the loop.
// PREFETCH: the full-feature kernels (C5 44.4 -> 43.6 ms) and, since round 3, the quad/box kernels: with the deep hierarchy of
// round 2 the 36 B of scratch it costs them outweighed it (C3 26.2 -> 26.5 ms at 100 spp); on the flattened programs a chain
// switch is a larger share of the frame and it pays (21.70 -> 21.25 ms).  (Reading the staged chains through a pointer typed
// as LDS -- ds_read instead of the flat loads a generic pointer compiles to, 3.6 per Cornell sample -- was measured in round 3
// as well: 21.51 without the prefetch, 21.70 with it, C5 29.46 -> 29.80 ms: the register allocator's answer costs more than
// the cheaper loads save; removed.)
template <bool PREFETCH = false, typename real>
RTK_DEV void apply_chain(const ChainRec<real>* __restrict__ chains, uint32_t chain, V3<real> wo, V3<real> wd, V3<real>& o, V3<real>& d) {
    o = wo;
    d = wd;
    if (chain == 0) return;
    const ChainRec<real>& ch = chains[chain];
    const int n = ch.count;
    auto step = [&](int rotate, real a, real b, real c) {
        if (rotate) {
            const real sn = a, cs = b;
            o = mk((cs * o.x) - (sn * o.z), o.y, (sn * o.x) + (cs * o.z));
            d = mk((cs * d.x) - (sn * d.z), d.y, (sn * d.x) + (cs * d.z));
        } else {
            o = o - mk(a, b, c);
        }
    };
    int first = 0;
    if constexpr (PREFETCH && RTK_CHAIN_PREFETCH) {
        const int r0 = ch.is_rotate[0], r1 = ch.is_rotate[1];
        const real a0 = ch.a[0], b0 = ch.b[0], c0 = ch.c[0], a1 = ch.a[1], b1 = ch.b[1], c1 = ch.c[1];
        if (n > 0) step(r0, a0, b0, c0);
        if (n > 1) step(r1, a1, b1, c1);
        first = 2;
    }
    for (int k = first; k < n; k++) step(ch.is_rotate[k], ch.a[k], ch.b[k], ch.c[k]);
}
// Hit point and normal back to world space (hittable.h:55,122-134), innermost first.
template <bool PREFETCH = false, typename real>
RTK_DEV void unapply_chain(const ChainRec<real>* __restrict__ chains, uint32_t chain, V3<real>& p, V3<real>& n) {
    if (chain == 0) return;
    const ChainRec<real>& ch = chains[chain];
    const int count = ch.count;
    auto step = [&](int rotate, real a, real b, real c) {
        if (rotate) {
            const real sn = a, cs = b;
            p = mk((cs * p.x) + (sn * p.z), p.y, (-sn * p.x) + (cs * p.z));
            n = mk((cs * n.x) + (sn * n.z), n.y, (-sn * n.x) + (cs * n.z));
        } else {
            p = p + mk(a, b, c);
        }
    };
    if constexpr (PREFETCH && RTK_CHAIN_PREFETCH) {
        const int r0 = ch.is_rotate[0], r1 = ch.is_rotate[1];
        const real a0 = ch.a[0], b0 = ch.b[0], c0 = ch.c[0], a1 = ch.a[1], b1 = ch.b[1], c1 = ch.c[1];
        for (int k = count - 1; k >= 2; k--) step(ch.is_rotate[k], ch.a[k], ch.b[k], ch.c[k]);
        if (count > 1) step(r1, a1, b1, c1);
        if (count > 0) step(r0, a0, b0, c0);
    } else {
        for (int k = count - 1; k >= 0; k--) step(ch.is_rotate[k], ch.a[k], ch.b[k], ch.c[k]);
    }
}

// ------------------------------------------------------------------ traversal --
// aabb::hit (aabb.h:61-85).  The reference's per-axis early-outs reduce to one
// final comparison because tmin only grows and tmax only shrinks.
//
// EXACT_NAN = true is the literal form: per axis `t0 < t1` selects near/far (a NaN
// from 0*inf makes the comparison false, i.e. the reference's else-branch), and
// max/min then drop a NaN operand exactly as its `>`/`<` updates do.
// EXACT_NAN = false is for rays whose 1/d is finite and non-zero on every axis
// (Lane::box_kind == OP_BOX): then t0 and t1 cannot be NaN (bounds and origin are finite), and
// near = min(t0,t1), far = max(t0,t1) are the same values as the select -- five
// instructions fewer per axis.  Rays with a zero or infinite direction component
// take the exact form (the aabb known-answer vectors exercise them).
RTK_DEV double rt_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
RTK_DEV float rt_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// Conservative-culling form, only for hierarchies whose boxes were grown by rtk_scene_optimize (F_FMA_BOX kernels,
// rtk_scene_upload_fast): t = b*inv - o*inv with one fused multiply-add per plane (o*inv hoisted per segment),
// 18 instead of 24 f64 operations per box.  Against (b - o)*inv the result moves by at most ~2^-52 |o*inv|, i.e.
// the plane by ~2^-52 |o| -- far inside the 2^-40 x scene-extent margin those boxes carry, so every box the exact
// test enters is entered here too and the closest hit (computed by the unchanged primitive tests) is the same.
// Rays with a zero or infinite direction component never come here (they take the literal form).
template <typename real>
RTK_DEV bool slab_test_fma(const Slot<real>& b, V3<real> oi, V3<real> inv, real tmin, real tmax) {
    const real t0x = rt_fma(b.v[0], inv.x, -oi.x), t1x = rt_fma(b.v[1], inv.x, -oi.x);
    const real t0y = rt_fma(b.v[2], inv.y, -oi.y), t1y = rt_fma(b.v[3], inv.y, -oi.y);
    const real t0z = rt_fma(b.v[4], inv.z, -oi.z), t1z = rt_fma(b.v[5], inv.z, -oi.z);
    const real nx = raw_min(t0x, t1x), fx = raw_max(t0x, t1x);
    const real ny = raw_min(t0y, t1y), fy = raw_max(t0y, t1y);
    const real nz = raw_min(t0z, t1z), fz = raw_max(t0z, t1z);
    tmin = raw_max(nz, raw_max(ny, raw_max(nx, tmin)));
    tmax = raw_min(fz, raw_min(fy, raw_min(fx, tmax)));
    return tmax > tmin;
}

template <bool EXACT_NAN, typename real>
RTK_DEV bool slab_test(const Slot<real>& b, V3<real> o, V3<real> inv, real tmin, real tmax) {
    const real t0x = (b.v[0] - o.x) * inv.x, t1x = (b.v[1] - o.x) * inv.x;
    const real t0y = (b.v[2] - o.y) * inv.y, t1y = (b.v[3] - o.y) * inv.y;
    const real t0z = (b.v[4] - o.z) * inv.z, t1z = (b.v[5] - o.z) * inv.z;
    real nx, fx, ny, fy, nz, fz;
    if constexpr (EXACT_NAN) {
        const bool lx = t0x < t1x, ly = t0y < t1y, lz = t0z < t1z;
        nx = lx ? t0x : t1x; fx = lx ? t1x : t0x;
        ny = ly ? t0y : t1y; fy = ly ? t1y : t0y;
        nz = lz ? t0z : t1z; fz = lz ? t1z : t0z;
    } else {
        nx = raw_min(t0x, t1x); fx = raw_max(t0x, t1x);
        ny = raw_min(t0y, t1y); fy = raw_max(t0y, t1y);
        nz = raw_min(t0z, t1z); fz = raw_max(t0z, t1z);
    }
    tmin = raw_max(nz, raw_max(ny, raw_max(nx, tmin)));
    tmax = raw_min(fz, raw_min(fy, raw_min(fx, tmax)));
    return tmax > tmin;
}

// Element e of a primitive whose reals are packed over consecutive slots ...
template <typename real, int E>
RTK_DEV real packed(const Slot<real>* rec) {
    return rec[E / Slot<real>::kReals].v[E % Slot<real>::kReals];
}
// ... or over the 16-byte units of a COMPACT record: elements 0..2 in the head, the header at bytes 24..31, element
// e >= 3 at byte 32 + 8 (e - 3) (rtk_device_layout.h).
template <typename real, int E>
RTK_DEV real packed(const Unit16* rec) {
    return real(reinterpret_cast<const double*>(rec)[E < 3 ? E : E + 1]);
}
// ... or over a quad record already in registers (COLD kernels: the whole 144-byte record is requested from memory at
// once, ahead of quad::hit's early-outs -- read field by field behind them it costs a second round trip to L2).
struct QuadRegs {
    double q[18];  // the record's nine units as doubles; q[3] holds the header's bits
};
template <typename real, int E>
RTK_DEV real packed(const QuadRegs* rec) {
    return real(rec->q[E < 3 ? E : E + 1]);
}
template <typename real, int E, typename Rec>
RTK_DEV V3<real> packed3(const Rec* rec) {
    return V3<real>{packed<real, E>(rec), packed<real, E + 1>(rec), packed<real, E + 2>(rec)};
}
// Header words, record length and the few fields whose position differs between the two layouts.
template <typename real> RTK_DEV uint32_t rec_kind_payload(const Slot<real>* rec) { return rec->kind_payload; }
template <typename real> RTK_DEV uint32_t rec_kind_payload(const Unit16* rec) { return rec[1].w[2]; }
template <typename real> RTK_DEV uint32_t rec_aux(const Slot<real>* rec) { return rec->aux; }
template <typename real> RTK_DEV uint32_t rec_aux(const Unit16* rec) { return rec[1].w[3]; }
template <typename real> RTK_DEV uint32_t rec_units(const Slot<real>*, uint32_t kind) { return uint32_t(slots_of<real>(kind)); }
template <typename real> RTK_DEV uint32_t rec_units(const Unit16*, uint32_t kind) { return uint32_t(compact_units(kind)); }
template <typename real> RTK_DEV V3<real> moving_dir(const Slot<real>* rec) { return mk(rec[1].v[0], rec[1].v[1], rec[1].v[2]); }   // centre2 - centre1 of a moving sphere
template <typename real> RTK_DEV V3<real> moving_dir(const Unit16* rec) { return packed3<real, 5>(rec); }

// n / a, correctly rounded, from y = RN(1/a) (computed once per segment by a true division): q0 = RN(n*y) is within
// one ulp of the quotient, r = n - a*q0 is exact in one fused multiply-add, and q0 + r*y rounds to RN(n/a)
// (Markstein's division step, the same correction the hardware division expansion ends with) -- three instructions
// instead of the ~25 of a full f64 division.  tests/test_division_identity.py checks the identity against `/` on
// 4e8 random operand pairs including all-ones significands.  (If q0 underflows the result is a denormal on either
// path and fails the `tmin < root` test below either way.)
template <typename real>
RTK_DEV real divide_by(real n, real a, real inv_a) {
    const real q0 = n * inv_a;
    const real r = rt_fma(-a, q0, n);
    return rt_fma(r, inv_a, q0);
}

// ---- exact ties (fast-order kernels) ---------------------------------------------------------------------------------
// Two primitives hit at EXACTLY the same distance: the reference resolves it by its visiting order -- a sphere accepts a
// root only strictly inside (tmin, closest so far) (sphere.h:44-48), so of two spheres the EARLIER one stays; a quad or a
// triangle accepts t == closest so far (quad.h:39, triangle.h:91), so the LATER one replaces whatever was there.  In a
// re-grouped hierarchy the visiting order is another one; the records carry their rank in the reference's order
// (SceneView::tie_rank, from rtk_node.c) and the same outcome is reproduced from the ranks: a quad/triangle beats a sphere,
// the higher rank wins among quads/triangles, the lower rank among spheres.  The tables are touched only when a tie
// actually occurs.  Unknown ranks (0, or no table) leave the decision to the visiting order, as before.
RTK_DEV bool is_strict_kind(uint32_t k) { return k == OP_SPHERE || k == OP_SPHERE_MOVING; }
RTK_DEV bool is_inclusive_kind(uint32_t k) { return k == OP_QUAD || k == OP_TRI; }
// a sphere's root equals the closest distance so far: does the sphere take the hit over?
RTK_DEV bool tie_sphere_wins(const uint32_t* __restrict__ ranks, uint32_t pc, uint32_t best_pc, uint32_t best_kind) {
    if (!ranks || best_pc == kNoHit || !is_strict_kind(best_kind)) return false;
    const uint32_t mine = ranks[pc], theirs = ranks[best_pc];
    return mine != 0 && theirs != 0 && mine < theirs;
}
// a quad's / triangle's t equals the closest distance so far (which its inclusive test admits): does it take the hit over?
RTK_DEV bool tie_inclusive_wins(const uint32_t* __restrict__ ranks, uint32_t pc, uint32_t best_pc, uint32_t best_kind) {
    if (!ranks || best_pc == kNoHit || !is_inclusive_kind(best_kind)) return true;
    const uint32_t mine = ranks[pc], theirs = ranks[best_pc];
    return mine == 0 || theirs == 0 || mine > theirs;
}

// sphere::hit up to the accepted root (sphere.h:32-49); cc = center.at(r.time()); a = d.d, inv_a = 1/a.
// TIE kernels: the interval test admits r == tmax as well -- the same two comparisons -- and only a root that was admitted
// is then checked for being that tie, which `wins_tie()` (tie_sphere_wins on the lane's state) decides; a tie that loses is
// a miss, exactly as in the reference, whose second root then lies beyond tmax.  The common path costs nothing extra.
template <bool TIE = false, typename real, typename F>
RTK_DEV bool sphere_root(V3<real> cc, real radius, V3<real> o, V3<real> d, real a, real inv_a, real tmin, real tmax, real& root, F&& wins_tie) {
    V3<real> oc = cc - o;
    real h = dot(d, oc);
    real c = length_squared(oc) - radius * radius;
    real disc = h * h - a * c;
    if (disc < real(0)) return false;
    real sq = rt_sqrt(disc);
    real r = divide_by(h - sq, a, inv_a);
    if constexpr (TIE) {
        if (!(tmin < r && r <= tmax)) {
            r = divide_by(h + sq, a, inv_a);
            if (!(tmin < r && r <= tmax)) return false;
        }
        if (r == tmax && !wins_tie()) return false;
    } else {
        if (!(tmin < r && r < tmax)) {
            r = divide_by(h + sq, a, inv_a);
            if (!(tmin < r && r < tmax)) return false;
        }
    }
    root = r;
    return true;
}
template <typename real>
RTK_DEV bool sphere_root(V3<real> cc, real radius, V3<real> o, V3<real> d, real a, real inv_a, real tmin, real tmax, real& root) {
    return sphere_root<false>(cc, radius, o, d, a, inv_a, tmin, tmax, root, [] { return false; });
}

// quad::hit (quad.h:29-73) on the packed record n(3),D,Q(3),w(3),v(3),u(3).
template <typename real, typename Rec>
RTK_DEV bool quad_test(const Rec* rec, V3<real> o, V3<real> d, real tmin, real tmax, real& t_out, real& alpha, real& beta) {
    V3<real> n = packed3<real, 0>(rec);
    real denom = dot(n, d);
    if (rt_fabs(denom) < real(1e-8)) return false;
    real t = (packed<real, 3>(rec) - dot(n, o)) / denom;
    if (!(tmin <= t && t <= tmax)) return false;
    V3<real> P = o + scale(t, d);
    V3<real> planar = P - packed3<real, 4>(rec);
    V3<real> w = packed3<real, 7>(rec);
    alpha = dot(w, cross(planar, packed3<real, 10>(rec)));
    beta = dot(w, cross(packed3<real, 13>(rec), planar));
    if (!(real(0) <= alpha && alpha <= real(1)) || !(real(0) <= beta && beta <= real(1))) return false;
    t_out = t;
    return true;
}

// triangle::hit (triangle.h:65-122): Moeller-Trumbore with the reference's float
// determinant (triangle.h:72,77) and float barycentrics (triangle.h:96-98), on
// the packed record e2(3),e1(3),p0(3).
template <typename real, typename Rec>
RTK_DEV bool tri_test(const Rec* rec, V3<real> o, V3<real> d, real tmin, real tmax, real& t_out, float& fa, float& fb, float& fg) {
    V3<real> e2 = packed3<real, 0>(rec), e1 = packed3<real, 3>(rec);
    V3<real> pvec = cross(d, e2);
    float det = float(dot(e1, pvec));
    if (__builtin_fabsf(det) < real(1e-8)) return false;
    float inv_det = 1.0f / det;
    V3<real> tvec = o - packed3<real, 6>(rec);
    real u = dot(tvec, pvec) * real(inv_det);
    if (u < real(0) || u > real(1)) return false;
    V3<real> qvec = cross(tvec, e1);
    real v = dot(d, qvec) * real(inv_det);
    if (v < real(0) || u + v > real(1)) return false;
    real t = dot(e2, qvec) * real(inv_det);
    if (t < tmin || t > tmax) return false;
    fa = float(real(1) - u - v);
    fb = float(u);
    fg = float(v);
    real da = real(fa), db = real(fb);
    if (!(real(0) <= da && da <= real(1)) || !(real(0) <= db && db <= real(1))) return false;
    t_out = t;
    return true;
}

// ------------------------------------------------------------------ lane state --
// Everything a lane carries between scheduler iterations.  One lane = one pixel;
// it walks that pixel's samples in order, each sample's segments in order, and
// each segment's traversal program in order -- the same sequence of arithmetic as
// the reference's recursion for that pixel, merely interleaved with other lanes.
template <typename real>
struct Lane {
    V3<real> ro, rd;         // world-space ray of the current segment (ray_color's `r`)
    V3<real> o, d, inv;      // the ray in the current chain's object space, and 1/d (aabb.h:67, hoisted: same value per box)
    V3<real> oi;             // o * inv, for the fused slab test of F_FMA_BOX kernels (dead, hence free, in the others)
    // F_F32_BOX kernels: the ray as the f32 culling boxes see it -- 1/d and o/d in float, the query interval rounded
    // outward.  inv and oi above are dead there.
    V3<float> inv32, oi32;
#if RTK_SIGNED_SLAB
    V3<float> oi32_lo;  // oi32 holds the UPPER bracket of o/d, this the lower one (see begin_culling32)
#endif
    float tmin32, tmax32;
    V3<float> inv32s, oi32s; // lean MIXED kernel (RTK_CH_SCALED): inv32 and oi32 times 1 / (end of the current interval), see rescale32
    float m2slack32;         // COMPACT kernels with centre / half-extent boxes: -2 x the ray's slack (slab_test32_che)
    real a, inv_a, tm;       // d.d (sphere.h:35, hoisted likewise) and 1/(d.d) for divide_by; ray time
    real tmin, best_t;       // current query interval: (tmin, closest so far)
    real sv_tmin, sv_best_t, rec1_t;  // constant_medium::hit nests two closest-hit queries of its boundary
                                      // (constant_medium.h:23,26); the outer query is parked here meanwhile
    V3<real> throughput, radiance, sum;
    uint32_t pc, kind, best_pc, sv_best_pc, rng;  // kind = record kind at pc (kept in a register so the vote needs no LDS read)
    int depth, s;
    uint32_t segs;           // segments traced for the current (pixel, chunk): the tile-cost estimate
    uint32_t box_kind;       // OP_BOX when 1/d is finite and non-zero on all three axes (slab tests cannot produce NaNs: the lane
                             // joins the box loop); kIrregularBox otherwise (its boxes go through step_other's literal slab test)
};

constexpr uint32_t kIrregularBox = 0xFFu;  // matches no record kind

template <typename real>
RTK_DEV bool regular_direction(V3<real> inv) {
    const real inf = real_inf<real>();
    return rt_fabs(inv.x) > real(0) && rt_fabs(inv.x) < inf && rt_fabs(inv.y) > real(0) && rt_fabs(inv.y) < inf && rt_fabs(inv.z) > real(0) &&
           rt_fabs(inv.z) < inf;
}

// The ray the records are tested against: the object-space ray when the scene has instance transforms, else the
// world-space ray itself (no second copy is kept in registers).
template <bool XF, typename real> RTK_DEV const V3<real>& ray_o(const Lane<real>& L) { if constexpr (XF) return L.o; else return L.ro; }
template <bool XF, typename real> RTK_DEV const V3<real>& ray_d(const Lane<real>& L) { if constexpr (XF) return L.d; else return L.rd; }

// The f64 query interval (tmin, best_t) as the f32 slab test uses it: rounded outward, so nothing the exact interval
// admits is rejected.  A float conversion rounds to nearest (error <= 2^-24 relative); scaling by 1 -/+ 2^-22 moves
// the bound safely past the exact value; infinities stay infinities.
RTK_DEV float below(double x) { const float f = float(x); return f - __builtin_fabsf(f) * 2.3841858e-07f; }
RTK_DEV float above(double x) { const float f = float(x); return f + __builtin_fabsf(f) * 2.3841858e-07f; }
RTK_DEV float below(float x) { return x; }
RTK_DEV float above(float x) { return x; }

template <typename real>
RTK_DEV void sync_interval32(Lane<real>& L) {
    L.tmin32 = below(L.tmin);
    L.tmax32 = above(L.best_t);
}

// F_F32_BOX: 1/d and o/d in float for the culling boxes (one v_rcp_f32 per axis instead of an f64 division), and
// whether the f32 test may be used at all: direction components neither zero nor beyond float range, origin inside
// the coordinate bound the box margin was sized for (rtk_api.cpp build_mixed_program).
template <int CH = 0, typename real>
RTK_DEV void begin_culling32(Lane<real>& L, V3<real> o, V3<real> d, float extent) {
    const V3<float> d32 = V3<float>{float(d.x), float(d.y), float(d.z)};
    L.inv32 = V3<float>{__builtin_amdgcn_rcpf(d32.x), __builtin_amdgcn_rcpf(d32.y), __builtin_amdgcn_rcpf(d32.z)};
    L.oi32 = V3<float>{float(o.x) * L.inv32.x, float(o.y) * L.inv32.y, float(o.z) * L.inv32.z};
#if RTK_SIGNED_SLAB
    if constexpr (CH >= 2) {
        // centre / half-extent boxes that carry only their own share: the origin's -- o/d off by < (2.5 + 1) 2^-23 |o/d| per
        // axis, the two final roundings of the test included -- becomes ONE slack for all axes, 2^-20 of the largest |o/d|
        const float big_oi = raw_max3(__builtin_fabsf(L.oi32.x), __builtin_fabsf(L.oi32.y), __builtin_fabsf(L.oi32.z));
        L.m2slack32 = -2.0f * (big_oi * (CH == 3 ? 1.9073486e-06f : 9.5367432e-07f));  // (3: oi32 becomes the bracket's upper end below -- twice the slack covers that)
    }
    if constexpr (CH == 0 || CH == 3)  // (centre / half-extent boxes: no bracket; 3 = a counting kernel that serves both record forms)
    // The origin's share of the float error travels with the RAY: o/d is bracketed, oi32_lo <= o/d <= oi32 -- the float
    // product is off by < 2^-21.9 relative (float(o) 2^-24, v_rcp_f32 1 ulp on float(d) 2^-24, the product 2^-24), and the
    // slab test's own final rounding adds 2^-24 of it -- so the boxes only have to carry their OWN share, 2^-21 of their own
    // coordinates (rtk_api.cpp), not 2^-19 of the largest coordinate in the scene.
    {
        const float k = 4.7683716e-07f;  // 2^-21
        const V3<float> e = V3<float>{__builtin_fabsf(L.oi32.x) * k, __builtin_fabsf(L.oi32.y) * k, __builtin_fabsf(L.oi32.z) * k};
        L.oi32_lo = V3<float>{L.oi32.x - e.x, L.oi32.y - e.y, L.oi32.z - e.z};
        L.oi32 = V3<float>{L.oi32.x + e.x, L.oi32.y + e.y, L.oi32.z + e.z};
    }
#endif
    // (RTK_CH_SCALED, lean MIXED kernel: the constants are multiplied once more by 1 / (end of the interval) <= 1, down to
    // 2^-42 / extent -- direction components within 2^-40 .. 2^40 keep every product a normal float; anything else takes
    // the exact path like a zero component does)
    const float big = (CH == 1 && RTK_CH_SCALED) ? 1.0e12f : 3.0e38f, tiny = (CH == 1 && RTK_CH_SCALED) ? 1.0e-12f : 1.0e-30f;
    const bool ok = __builtin_fabsf(L.inv32.x) < big && __builtin_fabsf(L.inv32.y) < big && __builtin_fabsf(L.inv32.z) < big &&
                    __builtin_fabsf(d32.x) < big && __builtin_fabsf(d32.y) < big && __builtin_fabsf(d32.z) < big &&
                    __builtin_fabsf(d32.x) > tiny && __builtin_fabsf(d32.y) > tiny && __builtin_fabsf(d32.z) > tiny &&
                    __builtin_fabsf(float(o.x)) <= extent && __builtin_fabsf(float(o.y)) <= extent && __builtin_fabsf(float(o.z)) <= extent;
    L.box_kind = ok ? uint32_t(OP_BOX) : kIrregularBox;
}

// world.hit(r, interval(0.001, inf), rec) (Camera.txt:211) starts here.
template <bool XF, bool MIXED = false, int CH = 0, typename real, bool COUNT>
RTK_DEV void begin_segment(Lane<real>& L, Counters<COUNT>& cnt, float extent = 0.0f) {
    cnt.inc(C_SEGMENTS);
    L.segs += 1;
    if constexpr (XF) {
        L.o = L.ro;
        L.d = L.rd;
    }
    L.a = length_squared(L.rd);
    L.inv_a = real(1) / L.a;
    L.tmin = real(0.001);
    L.best_t = real_inf<real>();
    if constexpr (MIXED) {
        begin_culling32<CH>(L, L.ro, L.rd, extent);
        sync_interval32(L);
        if constexpr (CH == 1 && RTK_CH_SCALED) rescale32(L, extent);
    } else {
        L.inv = mk(real(1) / L.rd.x, real(1) / L.rd.y, real(1) / L.rd.z);
        L.box_kind = regular_direction(L.inv) ? uint32_t(OP_BOX) : kIrregularBox;
        L.oi = L.ro * L.inv;
    }
    L.best_pc = kNoHit;
    L.pc = 0;
}

// What a primitive test needs to resolve an exact tie (see "exact ties" above): the rank table of the program the kernel
// executes and a way to learn the kind of the record at a pc (the current winner's).  TieCtx<false, ...> compiles to nothing.
template <bool TIE, typename KindOf>
struct TieCtx {
    static constexpr bool enabled = TIE;
    const uint32_t* ranks;
    KindOf kind_of;
    uint32_t shift = 0;  // a pc is (rank index << shift): 5 in the lean MIXED kernel, whose pcs count bytes
    RTK_DEV uint32_t at(uint32_t pc) const { return pc == kNoHit ? kNoHit : pc >> shift; }
    RTK_DEV bool sphere_wins(uint32_t pc, uint32_t best_pc) const { return tie_sphere_wins(ranks, at(pc), at(best_pc), best_pc == kNoHit ? 0u : kind_of(best_pc)); }
    RTK_DEV bool inclusive_wins(uint32_t pc, uint32_t best_pc) const { return tie_inclusive_wins(ranks, at(pc), at(best_pc), best_pc == kNoHit ? 0u : kind_of(best_pc)); }
};
struct NoTie {
    static constexpr bool enabled = false;
    RTK_DEV bool sphere_wins(uint32_t, uint32_t) const { return false; }
    RTK_DEV bool inclusive_wins(uint32_t, uint32_t) const { return true; }
};

// RTK_CH_SCALED (lean MIXED kernel): the box test works in t' = t * s with s = 1 / (end of the current interval), so that
// the interval is [~0, 1] and its two clamps are the output clamp of v_max3 / v_min3.  The end is tmax32 or, while there is
// no hit yet, a bound no hit can exceed: origin and every box lie in [-extent, extent]^3, so a hit has
// t <= 2 sqrt(3) extent / |d| <= 4 extent |1/d_x|.  s is rounded DOWN (v_rcp_f32 is good to an ulp; times 1 - 2^-22), so 1
// maps to a t at or beyond the end; the lower clamp sits at t = 0 instead of tmin = 0.001 -- both on the permissive side.
// The two extra roundings (inv * s, oi * s: 2^-24 each) are inside the margins the half-extents carry (rtk_api.cpp:
// 3 x 2^-23 of the 4 x 2^-23 budgeted for the box's own share, 4.5 x 2^-23 of the 8 x 2^-23 for the origin's).
template <typename real>
RTK_DEV void rescale32(Lane<real>& L, float extent) {
    const float bound = 4.0f * extent * __builtin_fabsf(L.inv32.x);
    const float end = raw_min(L.tmax32, bound);
    const float s = __builtin_amdgcn_rcpf(end) * 0.99999976f;
    L.inv32s = V3<float>{L.inv32.x * s, L.inv32.y * s, L.inv32.z * s};
    L.oi32s = V3<float>{L.oi32.x * s, L.oi32.y * s, L.oi32.z * s};
}
// A primitive test accepted t: it is the closest hit so far (hittable_list.h:27-31 / bvh.h:69 shrink the interval).
template <bool MIXED, bool SCALED = false, typename real>
RTK_DEV void take_hit(Lane<real>& L, real t, float extent = 0.0f) {
    L.best_t = t;
    L.best_pc = L.pc;
    if constexpr (MIXED) L.tmax32 = above(t);
    if constexpr (SCALED) rescale32(L, extent);
}
// The three primitive tests against the lane's current ray and interval, shared by every program layout: `units` is the
// record's length in that layout; MIXED kernels keep the float copy of the interval in step.
template <bool XF, bool MIXED, bool SCALED = false, typename real, bool COUNT, typename Tie>
RTK_DEV void hit_sphere(Lane<real>& L, V3<real> cc, real radius, uint32_t units, Counters<COUNT>& cnt, const Tie& tie, float extent = 0.0f) {
    cnt.inc(C_SPHERE);
    real r;
    if (sphere_root<Tie::enabled>(cc, radius, ray_o<XF>(L), ray_d<XF>(L), L.a, L.inv_a, L.tmin, L.best_t, r, [&] { return tie.sphere_wins(L.pc, L.best_pc); }))
        take_hit<MIXED, SCALED>(L, r, extent);
    L.pc += units;
}
template <bool XF, bool MIXED, typename real, bool COUNT, typename Rec, typename Tie>
RTK_DEV void hit_quad(Lane<real>& L, const Rec* __restrict__ rec, uint32_t units, Counters<COUNT>& cnt, const Tie& tie) {
    cnt.inc(C_QUAD);
    real t, al, be;
    if (quad_test(rec, ray_o<XF>(L), ray_d<XF>(L), L.tmin, L.best_t, t, al, be)) {
        bool ok = true;
        if constexpr (Tie::enabled) {
            if (t == L.best_t) ok = tie.inclusive_wins(L.pc, L.best_pc);
        }
        if (ok) take_hit<MIXED>(L, t);
    }
    L.pc += units;
}
template <bool XF, bool MIXED, typename real, bool COUNT, typename Rec, typename Tie>
RTK_DEV void hit_tri(Lane<real>& L, const Rec* __restrict__ rec, uint32_t units, Counters<COUNT>& cnt, const Tie& tie) {
    cnt.inc(C_TRI);
    real t;
    float fa, fb, fg;
    if (tri_test(rec, ray_o<XF>(L), ray_d<XF>(L), L.tmin, L.best_t, t, fa, fb, fg)) {
        bool ok = true;
        if constexpr (Tie::enabled) {
            if (t == L.best_t) ok = tie.inclusive_wins(L.pc, L.best_pc);
        }
        if (ok) take_hit<MIXED>(L, t);
    }
    L.pc += units;
}

// ---- F_F32_BOX: the steps on the MIXED / COMPACT programs (rtk_device_layout.h) -------------------------------
// Conservative slab test in float: the same min/max structure as slab_test_fma on bounds that were rounded outward
// and grown for exactly this arithmetic.  v_max3/v_min3 fold the reduction.
RTK_DEV bool slab_test32(const MixedHead& b, V3<float> oi, V3<float> inv, float tmin, float tmax) {
    const float t0x = __builtin_fmaf(b.f(0), inv.x, -oi.x), t1x = __builtin_fmaf(b.f(1), inv.x, -oi.x);
    const float t0y = __builtin_fmaf(b.f(2), inv.y, -oi.y), t1y = __builtin_fmaf(b.f(3), inv.y, -oi.y);
    const float t0z = __builtin_fmaf(b.f(4), inv.z, -oi.z), t1z = __builtin_fmaf(b.f(5), inv.z, -oi.z);
    const float nx = raw_min(t0x, t1x), fx = raw_max(t0x, t1x);
    const float ny = raw_min(t0y, t1y), fy = raw_max(t0y, t1y);
    const float nz = raw_min(t0z, t1z), fz = raw_max(t0z, t1z);
    const float near = raw_max(raw_max3(nx, ny, nz), tmin);
    const float far = raw_min(raw_min3(fx, fy, fz), tmax);
    return far >= near;  // >= : a tie is let through (conservative)
}
// UNITS = length of a box record: 1 in the MIXED program (32-byte units), 2 in the COMPACT one (16-byte units)
#if RTK_SIGNED_SLAB
// The sign-selected form: which bound of an axis is the near plane depends only on the sign of that direction component,
// known per ray -- three wave-level masks (SGPR pairs, taken once at the loop's entry: no lane changes its ray inside it)
// pick it with six v_cndmask instead of six min/max, so near and far planes can use the two ends of the o/d bracket:
// near = b_near/d - (o/d)_upper, far = b_far/d - (o/d)_lower.  Same instruction count as slab_test32.
struct SignMasks {
    unsigned long long x, y, z;  // bit l: lane l's direction component is negative
};
RTK_DEV float pick(float a, float b, unsigned long long mask) {  // lane-wise: mask bit set ? b : a
    float r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
}
RTK_DEV bool slab_test32_signed(const MixedHead& b, V3<float> oi_hi, V3<float> oi_lo, V3<float> inv, float tmin, float tmax, const SignMasks& m) {
    const float nx = __builtin_fmaf(pick(b.f(0), b.f(1), m.x), inv.x, -oi_hi.x), fx = __builtin_fmaf(pick(b.f(1), b.f(0), m.x), inv.x, -oi_lo.x);
    const float ny = __builtin_fmaf(pick(b.f(2), b.f(3), m.y), inv.y, -oi_hi.y), fy = __builtin_fmaf(pick(b.f(3), b.f(2), m.y), inv.y, -oi_lo.y);
    const float nz = __builtin_fmaf(pick(b.f(4), b.f(5), m.z), inv.z, -oi_hi.z), fz = __builtin_fmaf(pick(b.f(5), b.f(4), m.z), inv.z, -oi_lo.z);
    const float near = raw_max(raw_max3(nx, ny, nz), tmin);
    const float far = raw_min(raw_min3(fx, fy, fz), tmax);
    return far >= near;
}
template <uint32_t UNITS = 1, typename real, bool COUNT>
RTK_DEV void step_box32(Lane<real>& L, const MixedHead& rec, Counters<COUNT>& cnt, const SignMasks& m) {
    cnt.inc(C_BOX);
    const bool hit = slab_test32_signed(rec, L.oi32, L.oi32_lo, L.inv32, L.tmin32, L.tmax32, m);
    L.pc = hit ? L.pc + UNITS : rec.aux;
}
#else
struct SignMasks {};
template <uint32_t UNITS = 1, typename real, bool COUNT>
RTK_DEV void step_box32(Lane<real>& L, const MixedHead& rec, Counters<COUNT>& cnt, const SignMasks&) {
    cnt.inc(C_BOX);
    const bool hit = slab_test32(rec, L.oi32, L.inv32, L.tmin32, L.tmax32);
    L.pc = hit ? L.pc + UNITS : rec.aux;
}
#endif
// The lean MIXED kernel's program counters count BYTES (a unit is 32 of them) when its boxes are centre / half-extent
// records: the box step then needs no shift to turn its pc into an LDS address -- box step 19 -> 18 instructions.
constexpr uint32_t kChPcUnit = RTK_CH_BYTE_PC ? 32u : 1u;
// RTK_CH_BOX (MIXED program): the box as centre c = f[0..2] and half-extent h = f[3..5]; per axis
// tc = c/d - o/d, near = tc - h/|d|, far = tc + h/|d| -- which plane is the near one never has to be asked.  The
// half-extent was grown for exactly this arithmetic (rtk_api.cpp build_mixed_program), so the test is conservative.
RTK_DEV bool slab_test32_ch(const MixedHead& b, V3<float> oi, V3<float> inv, float tmin, float tmax) {
    const float tcx = __builtin_fmaf(b.f(0), inv.x, -oi.x), tcy = __builtin_fmaf(b.f(1), inv.y, -oi.y), tcz = __builtin_fmaf(b.f(2), inv.z, -oi.z);
    const float ax = __builtin_fabsf(inv.x), ay = __builtin_fabsf(inv.y), az = __builtin_fabsf(inv.z);  // source modifiers, no instructions
    const float nx = __builtin_fmaf(-b.f(3), ax, tcx), fx = __builtin_fmaf(b.f(3), ax, tcx);
    const float ny = __builtin_fmaf(-b.f(4), ay, tcy), fy = __builtin_fmaf(b.f(4), ay, tcy);
    const float nz = __builtin_fmaf(-b.f(5), az, tcz), fz = __builtin_fmaf(b.f(5), az, tcz);
    const float near = raw_max(raw_max3(nx, ny, nz), tmin);
    const float far = raw_min(raw_min3(fx, fy, fz), tmax);
    return far >= near;
}
// ... with the ray's slack (COMPACT programs): the planes may each be off by `slack` = -m2slack / 2 in t, so the box is
// passed when min(far, tmax) - max(near, tmin) >= -2 slack -- at the interval's ends a touch more permissive than
// clamping the widened planes, never less.
RTK_DEV bool slab_test32_che(const MixedHead& b, V3<float> oi, V3<float> inv, float tmin, float tmax, float m2slack) {
    const float tcx = __builtin_fmaf(b.f(0), inv.x, -oi.x), tcy = __builtin_fmaf(b.f(1), inv.y, -oi.y), tcz = __builtin_fmaf(b.f(2), inv.z, -oi.z);
    const float ax = __builtin_fabsf(inv.x), ay = __builtin_fabsf(inv.y), az = __builtin_fabsf(inv.z);
    const float nx = __builtin_fmaf(-b.f(3), ax, tcx), fx = __builtin_fmaf(b.f(3), ax, tcx);
    const float ny = __builtin_fmaf(-b.f(4), ay, tcy), fy = __builtin_fmaf(b.f(4), ay, tcy);
    const float nz = __builtin_fmaf(-b.f(5), az, tcz), fz = __builtin_fmaf(b.f(5), az, tcz);
    const float near = raw_max(raw_max3(nx, ny, nz), tmin);
    const float far = raw_min(raw_min3(fx, fy, fz), tmax);
    return far - near >= m2slack;
}
template <typename real, bool COUNT>
RTK_DEV void step_box32_che(Lane<real>& L, const MixedHead& rec, Counters<COUNT>& cnt) {
    cnt.inc(C_BOX);
    const bool hit = slab_test32_che(rec, L.oi32, L.inv32, L.tmin32, L.tmax32, L.m2slack32);
    L.pc = hit ? L.pc + 2u : rec.aux;
}
// ... in the ray parameter scaled by rescale32: the interval is [0, 1], its clamps are output modifiers.  Strict compare: a
// box has a positive extent along the ray (far' - near' = 2 h |inv'| > 0), so far' == near' after clamping means both were
// clamped to the same end -- the box lies wholly before 0 or wholly beyond 1.
RTK_DEV float max3_clamp01(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
RTK_DEV float min3_clamp01(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// (Packed f32 arithmetic for this test was measured in round 3 and removed: five v_pk_fma_f32 with op_sel / neg modifiers
// (inline asm on aligned register pairs: no moves) instead of nine v_fma_f32 -- 12 instead of 16 VALU instructions per box
// step in the ISA, same bits, same image -- ran C2 at 19.35 instead of 18.50 ms: the packed instruction issues at 0.24 per
// cycle per SIMD against 0.45 for v_fma_f32 (csrc/rtk_microbench.hip, aggregate rates: two operations per lane at half the
// rate -- five packed cost what ten plain ones do) and lengthens the step's dependent chain.  Left to the compiler
// (ext_vector_type(2) fma) the same test cost 9-19 extra moves per step: 22.6 ms.)
RTK_DEV bool slab_test32_chs(const MixedHead& b, V3<float> oi, V3<float> inv) {
    const float tcx = __builtin_fmaf(b.f(0), inv.x, -oi.x), tcy = __builtin_fmaf(b.f(1), inv.y, -oi.y), tcz = __builtin_fmaf(b.f(2), inv.z, -oi.z);
    const float ax = __builtin_fabsf(inv.x), ay = __builtin_fabsf(inv.y), az = __builtin_fabsf(inv.z);
    const float nx = __builtin_fmaf(-b.f(3), ax, tcx), fx = __builtin_fmaf(b.f(3), ax, tcx);
    const float ny = __builtin_fmaf(-b.f(4), ay, tcy), fy = __builtin_fmaf(b.f(4), ay, tcy);
    const float nz = __builtin_fmaf(-b.f(5), az, tcz), fz = __builtin_fmaf(b.f(5), az, tcz);
    return min3_clamp01(fx, fy, fz) > max3_clamp01(nx, ny, nz);
}
template <typename real, bool COUNT>
RTK_DEV void step_box32_ch(Lane<real>& L, const MixedHead& rec, Counters<COUNT>& cnt) {
    cnt.inc(C_BOX);
#if RTK_CH_SCALED
    const bool hit = slab_test32_chs(rec, L.oi32s, L.inv32s);
#else
    const bool hit = slab_test32_ch(rec, L.oi32, L.inv32, L.tmin32, L.tmax32);
#endif
    L.pc = hit ? L.pc + kChPcUnit : rec.aux;  // (pcs of this kernel count bytes, and so does the link)
}
// A box of those programs for a ray the float test must not judge (zero / out-of-range direction component, origin
// outside the sized bound): aabb::hit's literal form in f64 on the (outward-rounded, hence still enclosing) bounds.
template <bool XF = false, uint32_t UNITS = 1, uint32_t PCU = 1, typename real, bool COUNT>
RTK_DEV void step_box_mixed_exact(Lane<real>& L, const MixedHead& rec, Counters<COUNT>& cnt, bool compact_ch = false) {
    cnt.inc(C_BOX);
    Slot<real> b;
    if ((RTK_CH_BOX && UNITS == 1) || (UNITS == 2 && compact_ch)) {  // centre / half-extent records (the MIXED program; COMPACT programs of the mesh family)
        for (int k = 0; k < 3; k++) {
            b.v[2 * k] = real(rec.f(k)) - real(rec.f(3 + k));
            b.v[2 * k + 1] = real(rec.f(k)) + real(rec.f(3 + k));
        }
    } else {
        for (int k = 0; k < 6; k++) b.v[k] = real(rec.f(k));
    }
    const V3<real> d = ray_d<XF>(L);
    const V3<real> inv = mk(real(1) / d.x, real(1) / d.y, real(1) / d.z);
    const bool hit = slab_test<true>(b, ray_o<XF>(L), inv, L.tmin, L.best_t);
    L.pc = hit ? L.pc + UNITS * PCU : rec.aux;
}
// sphere::hit on a MIXED record: centre in the head unit, radius in the next one.
template <uint32_t PCU = 1, typename real, bool COUNT, typename Tie>
RTK_DEV void step_sphere_mixed(Lane<real>& L, const MixedHead& head, const MixedHead* __restrict__ rec, Counters<COUNT>& cnt, const Tie& tie, float extent) {
    const double radius = reinterpret_cast<const double*>(rec + 1)[0];
    hit_sphere<false, true, RTK_CH_BOX && RTK_CH_SCALED>(L, mk(real(head.d(0)), real(head.d(1)), real(head.d(2))), real(radius), 2u * PCU, cnt, tie, extent);
}
// ... and on a COMPACT record (3 units): the head (centre) is usually in registers already, the radius follows it.
template <bool XF, typename real, bool COUNT, typename Tie>
RTK_DEV void step_sphere_compact(Lane<real>& L, const MixedHead& head, const Unit16* __restrict__ rec, Counters<COUNT>& cnt, const Tie& tie) {
    hit_sphere<XF, true>(L, mk(real(head.d(0)), real(head.d(1)), real(head.d(2))), packed<real, 3>(rec), 3u, cnt, tie);
}
// The remaining record kinds of a sphere-only program: a moving sphere, or a box for an irregular ray.
template <uint32_t PCU = 1, typename real, bool COUNT, typename Tie>
RTK_DEV void step_other_mixed(Lane<real>& L, const MixedHead* __restrict__ rec, Counters<COUNT>& cnt, const Tie& tie, float extent) {
    const uint32_t kind = rec->kind_payload & 15u;
    if (kind == OP_BOX) {
        step_box_mixed_exact<false, 1, PCU>(L, *rec, cnt);
    } else if (kind == OP_SPHERE_MOVING) {
        const double* cont = reinterpret_cast<const double*>(rec + 1);
        const V3<real> cc = mk(real(rec->d(0)), real(rec->d(1)), real(rec->d(2))) + scale(L.tm, mk(real(cont[2]), real(cont[3]), real(cont[4])));
        hit_sphere<false, true, RTK_CH_BOX && RTK_CH_SCALED>(L, cc, real(cont[0]), 3u * PCU, cnt, tie, extent);
    } else {
        L.pc += uint32_t(mixed_units(kind)) * PCU;  // unreachable for a validated sphere-only program
    }
}
// bvh_node::hit's box test (bvh.h:65): on a miss skip the whole subtree.
template <bool EXACT_NAN, bool XF, bool FMA = false, typename real, bool COUNT>
RTK_DEV void step_box(Lane<real>& L, const Slot<real>& rec, Counters<COUNT>& cnt) {
    cnt.inc(C_BOX);
    bool hit;
    if constexpr (FMA && !EXACT_NAN) hit = slab_test_fma(rec, L.oi, L.inv, L.tmin, L.best_t);
    else hit = slab_test<EXACT_NAN>(rec, ray_o<XF>(L), L.inv, L.tmin, L.best_t);
    L.pc = hit ? L.pc + 1 : rec.aux;
}

// sphere::hit of a stationary sphere.
template <bool XF, typename real, bool COUNT, typename Tie>
RTK_DEV void step_sphere(Lane<real>& L, const Slot<real>& rec, Counters<COUNT>& cnt, const Tie& tie) {
    hit_sphere<XF, false>(L, mk(rec.v[0], rec.v[1], rec.v[2]), rec.v[3], 1u, cnt, tie);
}

// Which COMPACT programs hold centre / half-extent box records (RTK_CH_COMPACT): those of the mesh family -- the same rule
// in rtk_api.cpp, which also sets SceneView::compact_ch for the counting kernel, the one kernel that serves every family.
// (Measured: C4 62.3 -> 59.7 ms; the Cornell box and book 2, thin axis-aligned quads at coordinates in the hundreds
// and thousands, lose 2-4 % to the all-axes slack and keep the sign-selected test with its per-axis bracket.)
template <uint32_t FEAT>
constexpr bool kCompactChStatic = RTK_CH_COMPACT && (FEAT & F_F32_BOX) != 0 && (FEAT & ~uint32_t(F_FMA_BOX | F_F32_BOX | F_MATTE | F_LDS_BOXES)) == kFeatMesh;
template <uint32_t FEAT, bool COUNT, typename real>
RTK_DEV bool compact_ch_records(const SceneView<real>& sc) {
    if constexpr (kCompactChStatic<FEAT>) return true;
    else if constexpr (RTK_CH_COMPACT && COUNT) return sc.compact_ch != 0;
    else return false;
}
// Every other record kind (moving sphere, quad, triangle, chain switch, the three medium ops, a box met by a ray the
// box loop does not take).  `rec` points at the record in the program (LDS or global), in the slot layout (Slot<real>)
// or -- F_F32_BOX kernels of the non-lean families -- the COMPACT one (Unit16).
template <typename real, uint32_t FEAT, bool BRACKET = true, bool COUNT, typename Rec, typename Tie>
RTK_DEV void step_other(Lane<real>& L, const Rec* __restrict__ rec, const SceneView<real>& sc, Counters<COUNT>& cnt, const Tie& tie, float extent = 0.0f) {
    constexpr bool XF = (FEAT & F_XFORM) != 0;
    constexpr bool MIXED = (FEAT & F_F32_BOX) != 0;  // here: the COMPACT program (f32 culling boxes; the float interval follows every change)
    const uint32_t kp = rec_kind_payload<real>(rec);
    const uint32_t aux = rec_aux<real>(rec);
    const uint32_t kind = kp & 15u;
    if (kind == OP_BOX) {  // only for rays with a zero or infinite direction component: the literal, NaN-exact slab test
        if constexpr (MIXED) step_box_mixed_exact<XF, 2>(L, *reinterpret_cast<const MixedHead*>(rec), cnt, compact_ch_records<FEAT, COUNT>(sc));
        else step_box<true, XF>(L, *reinterpret_cast<const Slot<real>*>(rec), cnt);
    } else if (kind == OP_SPHERE_MOVING) {
        const V3<real> cc = packed3<real, 0>(rec) + scale(L.tm, moving_dir<real>(rec));
        hit_sphere<XF, MIXED>(L, cc, packed<real, 3>(rec), rec_units<real>(rec, kind), cnt, tie);
    } else if ((FEAT & F_QUAD) && kind == OP_QUAD) {
        hit_quad<XF, MIXED>(L, rec, rec_units<real>(rec, kind), cnt, tie);
    } else if ((FEAT & F_TRI) && kind == OP_TRI) {
        hit_tri<XF, MIXED>(L, rec, rec_units<real>(rec, kind), cnt, tie);
    } else if ((FEAT & F_XFORM) && kind == OP_CHAIN) {
        cnt.inc(C_XFORM, aux);
        apply_chain<kChainPrefetch<FEAT>>(sc.chains, kp >> 4, L.ro, L.rd, L.o, L.d);
        L.a = length_squared(L.d);
        L.inv_a = real(1) / L.a;
        if constexpr (MIXED) {
            begin_culling32(L, L.o, L.d, extent);  // (programs with instance chains keep the sign-selected test) 1/d, o/d in float for the boxes of this space; the interval (a ray parameter) is unchanged
        } else {
            L.inv = mk(real(1) / L.d.x, real(1) / L.d.y, real(1) / L.d.z);
            L.box_kind = regular_direction(L.inv) ? uint32_t(OP_BOX) : kIrregularBox;
            L.oi = L.o * L.inv;
        }
        L.pc += rec_units<real>(rec, kind);
    } else if ((FEAT & F_MEDIA) && kind == OP_MED_SPHERE) {
        // constant_medium::hit (constant_medium.h:20-53) with a stationary sphere as the boundary, in one step: the two
        // boundary->hit calls (universe, then (t1 + 0.0001, inf)) are two sphere::hit calls on the same sphere and ray, then the
        // clamp to the ray's interval, one random_double() and the scatter distance.  The lane's own interval and closest hit
        // are only touched by the final acceptance, so nothing is parked meanwhile.
        cnt.inc(C_MEDIUM);
        const V3<real> cc = packed3<real, 0>(rec);
        const real radius = packed<real, 3>(rec);
        real r1, r2;
        cnt.inc(C_SPHERE);
        if (sphere_root(cc, radius, ray_o<XF>(L), ray_d<XF>(L), L.a, L.inv_a, -real_inf<real>(), real_inf<real>(), r1)) {  // constant_medium.h:23
            cnt.inc(C_SPHERE);
            if (sphere_root(cc, radius, ray_o<XF>(L), ray_d<XF>(L), L.a, L.inv_a, r1 + real(0.0001), real_inf<real>(), r2)) {  // constant_medium.h:26
                if (r1 < L.tmin) r1 = L.tmin;
                if (r2 > L.best_t) r2 = L.best_t;
                if (r1 < r2) {
                    if (r1 < real(0)) r1 = real(0);
                    const real ray_length = rt_sqrt(L.a);
                    const real inside = (r2 - r1) * ray_length;
                    const real hit_distance = packed<real, 4>(rec) * call_log(rnd<real>(L.rng, cnt));
                    if (!(hit_distance > inside)) take_hit<MIXED>(L, r1 + hit_distance / ray_length);
                }
            }
        }
        L.pc += rec_units<real>(rec, kind);
    } else if ((FEAT & F_MEDIA) && BRACKET && kind == OP_MED_BEGIN) {
        cnt.inc(C_MEDIUM);
        L.sv_tmin = L.tmin;
        L.sv_best_t = L.best_t;
        L.sv_best_pc = L.best_pc;
        L.tmin = -real_inf<real>();
        L.best_t = real_inf<real>();
        L.best_pc = kNoHit;
        L.pc += rec_units<real>(rec, kind);
        if constexpr (MIXED) sync_interval32(L);
    } else if ((FEAT & F_MEDIA) && BRACKET && kind == OP_MED_MID) {
        if (L.best_pc == kNoHit) {  // constant_medium.h:23-24
            L.tmin = L.sv_tmin;
            L.best_t = L.sv_best_t;
            L.best_pc = L.sv_best_pc;
            L.pc = aux;
        } else {
            L.rec1_t = L.best_t;
            L.tmin = L.rec1_t + real(0.0001);  // constant_medium.h:26
            L.best_t = real_inf<real>();
            L.best_pc = kNoHit;
            L.pc += rec_units<real>(rec, kind);
        }
        if constexpr (MIXED) sync_interval32(L);
    } else if ((FEAT & F_MEDIA) && BRACKET && kind == OP_MED_END) {
        const bool hit2 = L.best_pc != kNoHit;
        real r2 = L.best_t;
        L.tmin = L.sv_tmin;
        L.best_t = L.sv_best_t;
        L.best_pc = L.sv_best_pc;
        if (hit2) {  // constant_medium.h:29-50
            real r1 = L.rec1_t;
            if (r1 < L.tmin) r1 = L.tmin;
            if (r2 > L.best_t) r2 = L.best_t;
            if (r1 < r2) {
                if (r1 < real(0)) r1 = real(0);
                const real ray_length = rt_sqrt(L.a);
                const real inside = (r2 - r1) * ray_length;
                const real hit_distance = packed<real, 0>(rec) * call_log(rnd<real>(L.rng, cnt));
                if (!(hit_distance > inside)) {
                    L.best_t = r1 + hit_distance / ray_length;
                    L.best_pc = L.pc;
                }
            }
        }
        L.pc += rec_units<real>(rec, kind);
        if constexpr (MIXED) sync_interval32(L);
    } else {
        L.pc += rec_units<real>(rec, kind);  // unreachable for a validated program
    }
}

// ------------------------------------------------------------------ textures --
// perlin::noise (perlin.h:14-37,72-89).
#if RTK_COLD_PERLIN
#define RTK_PERLIN_FN RTK_NOINLINE
#else
#define RTK_PERLIN_FN RTK_DEV
#endif
template <typename real, bool COUNT>
RTK_PERLIN_FN real perlin_noise(const PerlinRec<real>& pn, V3<real> p, Counters<COUNT>& cnt) {
    cnt.inc(C_NOISE);
    const real fx = rt_floor(p.x), fy = rt_floor(p.y), fz = rt_floor(p.z);
    const real u = p.x - fx, v = p.y - fy, w = p.z - fz;
    const int i = int(fx), j = int(fy), k = int(fz);
    const real uu = u * u * (real(3) - real(2) * u);
    const real vv = v * v * (real(3) - real(2) * v);
    const real ww = w * w * (real(3) - real(2) * w);
    real accum = real(0);
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int idx = pn.perm_x[(i + a) & 255] ^ pn.perm_y[(j + b) & 255] ^ pn.perm_z[(k + c) & 255];
                const V3<real> g = ld3(pn.randvec[idx]);
                const V3<real> wv = mk(u - real(a), v - real(b), w - real(c));
                accum += (real(a) * uu + real(1 - a) * (real(1) - uu)) * (real(b) * vv + real(1 - b) * (real(1) - vv)) *
                         (real(c) * ww + real(1 - c) * (real(1) - ww)) * dot(g, wv);
            }
    return accum;
}

// get_sphere_uv (sphere.h:67-73) and noise_texture::value's tail (texture.h:114-116: 1 + sin(...)) are libm-heavy f64 code
// (acos, atan2, sin: long polynomial expansions) that few lanes ever reach; inlined into the shade step they sit on top of
// everything live there and account for most of the full-feature kernel's scratch (compile-only ablation,
// tools/kernel_resources.py: without the two 248 VGPRs and no scratch at the 2-wave bound, with them 256 + 292 B).  Kept OUT
// OF LINE (RTK_COLD_UV / RTK_COLD_NOISE) a call saves and restores around itself on the cold path only: 248 VGPRs + 32 B at
// two waves per SIMD, 168 + 352 B at three (it was 168 + 640 B) -- and three waves then win: C5 at 32 spp 42.05 ms (inline,
// 2 waves) -> 40.5 (out of line, 2 waves) -> 38.85 (out of line, 3 waves); out of line at 3 waves with only the uv function
// moved: 47.3.  Same image.  (Moving more -- pow, perlin::turb, the image lookup, get_lighting, the media's log -- was
// measured too and lost: 43.0-46.7 ms at three waves.)
#ifndef RTK_COLD_UV
#define RTK_COLD_UV 1
#endif
#ifndef RTK_COLD_NOISE
#define RTK_COLD_NOISE 1
#endif
#if RTK_COLD_UV
#define RTK_UV_FN __device__ __attribute__((noinline))
#else
#define RTK_UV_FN RTK_DEV
#endif
#if RTK_COLD_NOISE
#define RTK_NOISE_FN __device__ __attribute__((noinline))
#else
#define RTK_NOISE_FN RTK_DEV
#endif
template <typename real>
RTK_UV_FN void sphere_uv(real ox, real oy, real oz, real* __restrict__ uv) {  // sphere.h:67-73 on the outward unit normal
    const real pi = real(3.1415926535897932385);
    const real theta = rt_acos(-oy);
    const real phi = rt_atan2(-oz, ox) + pi;
    uv[0] = phi / (real(2) * pi);
    uv[1] = theta / pi;
}
template <typename real>
RTK_NOISE_FN real noise_tail(real arg) { return real(1) + rt_sin(arg); }  // texture.h:115: 1 + sin(scale * p.z + 10 * turb)

// texture::value (texture.h:20-120); checker nesting is followed iteratively.
#if RTK_COLD_TEX
#define RTK_TEX_FN RTK_NOINLINE
#else
#define RTK_TEX_FN RTK_DEV
#endif
template <typename real, bool COUNT>
RTK_TEX_FN V3<real> texture_value(const SceneView<real>& sc, int tex, real u, real v, V3<real> p, Counters<COUNT>& cnt) {
    for (;;) {
        const TextureRec<real>& t = sc.textures[tex];
        if (t.kind == RTK_TEX_SOLID) return ld3(t.color);
        if (t.kind == RTK_TEX_CHECKER) {  // texture.h:42-50
            const int xi = int(rt_floor(t.param * p.x));
            const int yi = int(rt_floor(t.param * p.y));
            const int zi = int(rt_floor(t.param * p.z));
            tex = ((xi + yi + zi) % 2 == 0) ? t.even : t.odd;
            continue;
        }
        if (t.kind == RTK_TEX_CHECKER_TRI) {  // texture.h:66-76
            v = real(1) - v;
            const int ui = int(rt_round(t.param * u * real(10)));
            const int vi = int(rt_round(t.param * v * real(10)));
            tex = ((ui + vi) % 2 == 0) ? t.even : t.odd;
            continue;
        }
        if (t.kind == RTK_TEX_IMAGE) {  // texture.h:90-104, rtw_stb_image.h:71-81
            const ImageRec im = sc.images[t.image];
            if (im.width <= 0 || im.height <= 0) return mk(real(0), real(1), real(1));
            u = u < real(0) ? real(0) : (u > real(1) ? real(1) : u);
            const real vc = v < real(0) ? real(0) : (v > real(1) ? real(1) : v);
            v = real(1) - vc;
            int i = int(u * real(im.width));
            int j = int(v * real(im.height));
            i = i < 0 ? 0 : (i < im.width ? i : im.width - 1);
            j = j < 0 ? 0 : (j < im.height ? j : im.height - 1);
            cnt.inc(C_TEXEL);
            const uint8_t* px = sc.texels + im.texel_offset + (int64_t(j) * im.width + i) * 3;
            const real cs = real(1) / real(255);
            return mk(cs * real(px[0]), cs * real(px[1]), cs * real(px[2]));
        }
        // noise_texture (texture.h:114-116) with perlin::turb(p, 7) (perlin.h:38-50)
        const PerlinRec<real>& pn = sc.perlins[t.image];
        real accum = real(0), weight = real(1);
        V3<real> q = p;
        for (int k = 0; k < 7; k++) {
            accum += weight * perlin_noise(pn, q, cnt);
            weight *= real(0.5);
            q = mk(q.x * real(2), q.y * real(2), q.z * real(2));
        }
        const real s = noise_tail(t.param * p.z + real(10) * rt_fabs(accum));
        return mk(s * real(0.5), s * real(0.5), s * real(0.5));
    }
}

template <typename real, uint32_t FEAT, bool COUNT>
RTK_DEV V3<real> material_color(const SceneView<real>& sc, const MaterialRec<real>& m, real u, real v, V3<real> p, Counters<COUNT>& cnt) {
    if (!(FEAT & F_TEXTURE) || m.tex < 0) return ld3(m.albedo);  // solid colours are folded into the material at upload
    return texture_value(sc, m.tex, u, v, p, cnt);
}

#ifdef RTK_PROFILE
// Profile build: shade's sub-phases, accumulated per wave (flushed to counters[kShadeProfBase + 2 k] cycles / [.. + 1]
// calls: k = 0 the deferred hit record, 1 materials and textures, 2 the miss path in front of them, 3 the scattered ray).
struct ShadeProf {
    unsigned long long t[4] = {0, 0, 0, 0}, n[4] = {0, 0, 0, 0};
};
constexpr int kShadeProfBase = 32 + 4 * 4096;
#define RTK_SHADE_PROF_PARAM , ShadeProf& sprof
#define RTK_SHADE_PROF_ARG , sprof
#define RTK_SHADE_PROF_BEGIN unsigned long long sp_prev_ = __builtin_amdgcn_s_memtime();
#define RTK_SHADE_PROF(k)                                             \
    {                                                                 \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        sprof.t[k] += now_ - sp_prev_;                                \
        sprof.n[k] += 1;                                              \
        sp_prev_ = now_;                                              \
    }
#else
#define RTK_SHADE_PROF_PARAM
#define RTK_SHADE_PROF_ARG
#define RTK_SHADE_PROF_BEGIN
#define RTK_SHADE_PROF(k)
#endif

// ------------------------------------------------------------------ shading ---
template <typename real>
struct Surface {  // hit_record (hittable.h:11-27)
    V3<real> p, normal;
    real u, v;
    int material;
    bool front_face;
};

// The deferred hit record of a MIXED program's winner (sphere.h:50-56; solid colours only: no uv).
template <typename real>
RTK_DEV void make_surface_mixed(const MixedHead* __restrict__ prog, uint32_t best_pc, real t, V3<real> wo, V3<real> wd, real tm, Surface<real>& sf) {
    const MixedHead* rec = prog + best_pc;
    const uint32_t kind = rec->kind_payload & 15u;
    const double* cont = reinterpret_cast<const double*>(rec + 1);
    sf.material = int(rec->aux >> 8);
    sf.p = wo + scale(t, wd);
    sf.u = real(0);
    sf.v = real(0);
    V3<real> cc = mk(real(rec->d(0)), real(rec->d(1)), real(rec->d(2)));
    if (kind == OP_SPHERE_MOVING) cc = cc + scale(tm, mk(real(cont[2]), real(cont[3]), real(cont[4])));
    const V3<real> outward = scale(real(cont[1]), sf.p - cc);
    sf.front_face = dot(wd, outward) < real(0);
    sf.normal = sf.front_face ? outward : -outward;
}


// The barycentrics of the winning triangle for its UVs (triangle.h:96-110): triangle::hit once more on the winner, with an
// open interval.  RTK_COLD_TRI_UV keeps it out of line (A/B: the mesh kernels at four waves per SIMD).
#ifndef RTK_COLD_TRI_UV
#define RTK_COLD_TRI_UV 0
#endif
#if RTK_COLD_TRI_UV
#define RTK_TRI_UV_FN RTK_NOINLINE
#else
#define RTK_TRI_UV_FN RTK_DEV
#endif
template <typename real, typename Rec>
RTK_TRI_UV_FN void tri_barycentrics(const Rec* __restrict__ rec, real ox, real oy, real oz, real dx, real dy, real dz, float* fa, float* fb, float* fg) {
    real tt;
    float a = 0, b = 0, g = 0;
    tri_test(rec, mk(ox, oy, oz), mk(dx, dy, dz), -real_inf<real>(), real_inf<real>(), tt, a, b, g);
    *fa = a;
    *fb = b;
    *fg = g;
}

// Build the hit record of the winning record (the deferred half of *.hit).  Sphere
// and quad geometry is read from the program record itself (LDS when staged); only
// triangles go to their side record for the normal and the UVs.  `prog` is the slot
// program or the COMPACT one (same payload element numbering in both).
template <typename real, uint32_t FEAT, typename Rec>
RTK_DEV void make_surface(const Rec* __restrict__ prog, const SceneView<real>& sc, const MaterialRec<real>* __restrict__ mats, uint32_t best_pc, real t,
                          V3<real> wo, V3<real> wd, real tm, Surface<real>& sf, bool force_uv = false) {
    const Rec* rec = prog + best_pc;
    const uint32_t kp = rec_kind_payload<real>(rec), aux = rec_aux<real>(rec);
    const uint32_t kind = kp & 15u;
    const uint32_t idx = kp >> 4;
    const uint32_t chain = (FEAT & F_XFORM) ? (aux & 255u) : 0u;
    sf.material = int(aux >> 8);
    V3<real> o = wo, d = wd;
    if (FEAT & F_XFORM) apply_chain<kChainPrefetch<FEAT>>(sc.chains, chain, wo, wd, o, d);
    sf.p = o + scale(t, d);
    sf.u = real(0);
    sf.v = real(0);
    V3<real> outward;
    bool face_from_ray = true;
    if (kind == OP_SPHERE || kind == OP_SPHERE_MOVING) {  // sphere.h:50-56,67-73
        V3<real> cc = packed3<real, 0>(rec);
        if (kind == OP_SPHERE_MOVING) cc = cc + scale(tm, moving_dir<real>(rec));
        outward = scale(packed<real, 4>(rec), sf.p - cc);  // (p - center) / radius = (1/radius) * (p - center) (vec3.h:91-93); element 4 = 1/radius from the upload
        if ((FEAT & F_TEXTURE) && (force_uv || mats[sf.material].needs_uv)) {
            real uv[2];
            sphere_uv(outward.x, outward.y, outward.z, uv);
            sf.u = uv[0];
            sf.v = uv[1];
        }
    } else if ((FEAT & F_QUAD) && kind == OP_QUAD) {  // quad.h:44-57; record = n(3),D,Q(3),w(3),v(3),u(3)
        V3<real> planar = sf.p - packed3<real, 4>(rec);
        V3<real> w = packed3<real, 7>(rec);
        sf.u = dot(w, cross(planar, packed3<real, 10>(rec)));
        sf.v = dot(w, cross(packed3<real, 13>(rec), planar));
        outward = packed3<real, 0>(rec);
    } else if ((FEAT & F_TRI) && kind == OP_TRI) {  // triangle.h:96-110
        const TriRec<real>& tr = sc.tris[idx];
        real tt;
        float fa = 0, fb = 0, fg = 0;
        tri_barycentrics(rec, o.x, o.y, o.z, d.x, d.y, d.z, &fa, &fb, &fg);
        sf.u = real(fa * tr.uv0[0] + fb * tr.uv1[0] + fg * tr.uv2[0]);
        sf.v = real(fa * tr.uv0[1] + fb * tr.uv1[1] + fg * tr.uv2[1]);
        outward = ld3(tr.n);
    } else {  // OP_MED_END: constant_medium.h:45-50
        outward = mk(real(1), real(0), real(0));
        face_from_ray = false;
    }
    if (face_from_ray) {  // hittable.h:23-26
        sf.front_face = dot(d, outward) < real(0);
        sf.normal = sf.front_face ? outward : -outward;
    } else {
        sf.front_face = true;
        sf.normal = outward;
    }
    if (FEAT & F_XFORM) unapply_chain<kChainPrefetch<FEAT>>(sc.chains, chain, sf.p, sf.normal);
}

// get_lighting (Camera.txt:240-272).
template <typename real>
RTK_DEV V3<real> point_lighting(const SceneView<real>& sc, V3<real> p, V3<real> normal) {
    V3<real> result = mk(real(0), real(0), real(0));
    for (int i = 0; i < sc.n_lights; i++) {
        const LightRec<real>& L = sc.lights[i];
        V3<real> dir = ld3(L.position) - p;
        real dist2 = length_squared(dir);
        dir = unit_vector(dir);
        real dn = dot(normal, dir);
        real diffuse = dn > real(0) ? dn : real(0);
        real radius_effect = L.size * real(0.1);
        if (dist2 <= L.size * L.size) {
            result = result + scale(diffuse, ld3(L.intensity));
        } else {
            real att = real(1) / (dist2 + radius_effect);
            result = result + scale(diffuse, scale(att, ld3(L.intensity)));
        }
    }
    return result;
}

// get_ray (Camera.txt:177-200) for sample L.s of pixel (i, j): seeds the sample's
// RNG stream, draws sample_square (y then x, g++ order), the lens, then the time.
template <typename real, bool COUNT>
RTK_DEV void begin_sample(Lane<real>& L, const CameraRec<real>& cam, int i, int j, uint32_t seed_hash, Counters<COUNT>& cnt) {
    cnt.inc(C_SAMPLES);
    L.rng = pcg_hash(uint32_t(j * cam.width + i) + pcg_hash(uint32_t(L.s) + seed_hash));
    const real oy = rnd<real>(L.rng, cnt) - real(0.5);
    const real ox = rnd<real>(L.rng, cnt) - real(0.5);
    const V3<real> pixel_sample = ld3(cam.pixel00) + scale(real(i) + ox, ld3(cam.du)) + scale(real(j) + oy, ld3(cam.dv));
    V3<real> ro = ld3(cam.center);
    if (cam.defocus_angle > real(0)) {  // vec3.h:135-142
        real px, py;
        for (;;) {
            py = rnd_pm1<real>(L.rng, cnt);
            px = rnd_pm1<real>(L.rng, cnt);
            if (px * px + py * py < real(1)) break;  // (the reference adds z * z = 0 * 0: x + 0 == x for x >= +0)
        }
        ro = ld3(cam.center) + scale(px, ld3(cam.disk_u)) + scale(py, ld3(cam.disk_v));
    }
    L.ro = ro;
    L.rd = pixel_sample - ro;
    L.tm = rnd<real>(L.rng, cnt);
    L.radiance = mk(real(0), real(0), real(0));
    L.throughput = mk(real(1), real(1), real(1));
    L.depth = cam.max_depth;
}

// Kernels without point lights and emissive materials: a path gathers radiance only at the miss that ends it, so
// `radiance` is +0 until then and `pixel_color += radiance` (Camera.txt:72) adds either +0 (sum + 0 == sum: the sum is
// never -0) or 0 + throughput * background == throughput * background.  Those kernels keep no radiance registers
// (lean MIXED kernel: 119 -> 113 VGPRs, C2 20.95 -> 20.84 ms, same framebuffer).
// (round 3) The same holds with emissive materials as long as there are no point lights: a diffuse_light ends the path it
// emits into (material.h:99-101,116-118), so `radiance` is +0 until then as well and 0 + throughput * emitted == throughput *
// emitted.  Only get_lighting (Camera.txt:228) adds radiance in the middle of a path.  The quad/box and mesh families (area
// lights, no point lights) drop six registers of lane state with it: C3's kernel 128 VGPRs + 16 B -> 125 + 0, C4's 128 + 44 B
// -> 126 + 0 (53.0 -> 51.8 ms).
template <uint32_t FEAT>
constexpr bool kNoRadianceState = (FEAT & F_LIGHTS) == 0;

// material::scatter / emitted on a finished hit record (material.h:22-172) and the rest of ray_color's body
// (Camera.txt:216-237): what shade() runs after it has built the record -- and what the known-answer entry point
// rtk_debug_scatter runs on caller-supplied records.  Returns true when the sample's path has ended.
template <typename real, uint32_t FEAT, int FORCE_KIND = -1, bool COUNT>
RTK_DEV bool shade_surface(Lane<real>& L, const Surface<real>& sf, const SceneView<real>& sc, const MaterialRec<real>* __restrict__ mats,
                           Counters<COUNT>& cnt RTK_SHADE_PROF_PARAM) {
    RTK_SHADE_PROF_BEGIN
    const MaterialRec<real>& m = mats[sf.material];
    const int mkind = FORCE_KIND >= 0 ? FORCE_KIND : m.kind;  // (FORCE_KIND: the instruction-cost probes compile one material's path alone, tools/isa_costs.py)
    const V3<real> rd = L.rd;

    V3<real> attenuation, next_d;
    bool scattered = true;
    // Code shared between material branches is hoisted in front of them, so that a wave whose lanes hold different
    // materials executes it once instead of once per branch:
    //  - lambertian and metal both start by drawing random_unit_vector() (material.h:30,84): three draws, a square
    //    root, a division;
    //  - metal normalises the reflected direction (material.h:83), dielectric and specular the incoming one
    //    (material.h:52,146): a square root and a division.
    // The hoisted values stay live across the branches; the kernels with instance transforms are already short of
    // registers (they spill), so those keep one copy per branch (kHoist = false).
    constexpr bool kHoist = (FEAT & F_XFORM) == 0;
    V3<real> ruv = mk(real(0), real(0), real(0)), unit_in = rd;
    if constexpr (kHoist) {
        if (mkind == RTK_MAT_LAMBERTIAN || mkind == RTK_MAT_METAL) ruv = random_unit_vector<real>(L.rng, cnt);
        if (mkind == RTK_MAT_METAL) unit_in = reflect(rd, sf.normal);
        if (mkind == RTK_MAT_METAL || mkind == RTK_MAT_DIELECTRIC || ((FEAT & F_EXOTIC_MAT) && mkind == RTK_MAT_SPECULAR)) unit_in = unit_vector(unit_in);
    }
    if (mkind == RTK_MAT_LAMBERTIAN) {  // material.h:29-38
        if constexpr (!kHoist) ruv = random_unit_vector<real>(L.rng, cnt);
        V3<real> dir = sf.normal + ruv;
        if (near_zero(dir)) dir = sf.normal;
        next_d = dir;
        attenuation = material_color<real, FEAT>(sc, m, sf.u, sf.v, sf.p, cnt);
    } else if (!(FEAT & F_MATTE) && mkind == RTK_MAT_METAL) {  // material.h:82-88
        if constexpr (!kHoist) {
            ruv = random_unit_vector<real>(L.rng, cnt);
            unit_in = unit_vector(reflect(rd, sf.normal));
        }
        V3<real> fuzz = scale(m.param, ruv);
        next_d = unit_in + fuzz;
        attenuation = ld3(m.albedo);
        scattered = dot(next_d, sf.normal) > real(0);
    } else if (!(FEAT & F_MATTE) && mkind == RTK_MAT_DIELECTRIC) {  // material.h:47-65
        attenuation = mk(real(1), real(1), real(1));
        // 1/refraction_index and Schlick's r0^2 for both faces are per-material constants: computed once at upload,
        // in double, by the same expressions (material.h:50,71-72)
        const real ri = sf.front_face ? m.albedo[0] : m.param;
        if constexpr (!kHoist) unit_in = unit_vector(rd);
        const V3<real> unit_d = unit_in;
        const real cos_theta = rt_fmin(dot(-unit_d, sf.normal), real(1));
        const real sin_theta = rt_sqrt(real(1) - cos_theta * cos_theta);
        bool reflect_it = ri * sin_theta > real(1);
        if (!reflect_it) {  // Schlick (material.h:69-74); the draw is skipped on total internal reflection
            const real r0 = sf.front_face ? m.albedo[1] : m.albedo[2];
            // Schlick's (1 - cos)^5 (material.h:73 calls pow) by multiplication, ~200 instructions less than pow in the
            // dielectric branch -- in f64 CORRECTLY ROUNDED (pow5: the products carried with their rounding errors), which is
            // what glibc's pow returns in 99.92 % of cases (tests/test_schlick_power.py; the plain ((x^2)^2)*x of rounds 1-2
            // agreed with it in 49 %).  The value only feeds the comparison with a 24-bit uniform below: a different outcome
            // needs pow's own misrounding AND the reflectance within an ulp of a multiple of 2^-24 -- ~1e-12 per dielectric
            // interaction, i.e. ~1e-4 per full C2 frame (it was ~1e-9 and 0.1-1 per frame).
            const real omc = real(1) - cos_theta;
            const real refl = r0 + (real(1) - r0) * pow5(omc);
            reflect_it = refl > rnd<real>(L.rng, cnt);
        }
        next_d = reflect_it ? reflect(unit_d, sf.normal) : refract(unit_d, sf.normal, ri);
    } else if ((FEAT & F_EXOTIC_MAT) && !(FEAT & F_MATTE) && mkind == RTK_MAT_ISOTROPIC) {  // material.h:129-134
        next_d = random_unit_vector<real>(L.rng, cnt);
        attenuation = material_color<real, FEAT>(sc, m, sf.u, sf.v, sf.p, cnt);
    } else if ((FEAT & F_EXOTIC_MAT) && !(FEAT & F_MATTE) && mkind == RTK_MAT_SPECULAR) {  // material.h:145-167
        if constexpr (!kHoist) unit_in = unit_vector(rd);
        const V3<real> unit_d = unit_in;
        const V3<real> refl = reflect(unit_d, sf.normal);
        V3<real> diffuse = random_unit_vector<real>(L.rng, cnt);  // random_on_hemisphere, vec3.h:116-124
        if (!(dot(diffuse, sf.normal) > real(0))) diffuse = -diffuse;
        const real f = call_pow(real(1) - dot(refl, unit_d), m.param);
        V3<real> dir = scale(f, refl) + scale(real(1) - f, diffuse);
        if (near_zero(dir)) dir = sf.normal;
        next_d = dir;
        attenuation = ld3(m.albedo);
    } else {  // diffuse_light / emissive_light: emits, never scatters (material.h:99-101,116-118)
        if (FEAT & F_EXOTIC_MAT) {
            V3<real> emitted = material_color<real, FEAT>(sc, m, sf.u, sf.v, sf.p, cnt);
            if constexpr (kNoRadianceState<FEAT>) L.sum = L.sum + L.throughput * emitted;  // = sum + (0 + throughput * emitted), the same bits
            else L.radiance = L.radiance + L.throughput * emitted;
        }
        return true;
    }
    RTK_SHADE_PROF(1)
    if (!scattered) return true;  // Camera.txt:223-225 (emission of a scattering material is zero)
    if ((FEAT & F_LIGHTS) && sc.n_lights > 0) {  // Camera.txt:228
        V3<real> lighting = attenuation * point_lighting(sc, sf.p, sf.normal);
        L.radiance = L.radiance + L.throughput * lighting;
    }
    L.throughput = L.throughput * attenuation;
    L.ro = sf.p;
    L.rd = next_d;
    L.depth -= 1;
    RTK_SHADE_PROF(3)
    return L.depth <= 0;  // Camera.txt:205-206: the next ray_color call returns black
}

// The traversal program ran to OP_END: the body of ray_color after world.hit
// (Camera.txt:211-237), iteratively (radiance = sum of throughput * emission).
// Returns true when the sample's path has ended.
// `hit_rec` = the record of the closest hit (program + L.best_pc; anything when there is none).
template <typename real, uint32_t FEAT, bool COUNT, typename ProgT>
RTK_DEV bool shade(Lane<real>& L, const ProgT* __restrict__ hit_rec, const SceneView<real>& sc, const MaterialRec<real>* __restrict__ mats,
                   const CameraRec<real>& cam, Counters<COUNT>& cnt RTK_SHADE_PROF_PARAM) {
    RTK_SHADE_PROF_BEGIN
    if (L.best_pc == kNoHit) {  // Camera.txt:211-213
        if constexpr (kNoRadianceState<FEAT>) L.sum = L.sum + L.throughput * ld3(cam.background);  // = sum + (0 + throughput * background), the same bits
        else L.radiance = L.radiance + L.throughput * ld3(cam.background);
        return true;
    }
    cnt.inc(C_SURFACE);
    RTK_SHADE_PROF(2)
    Surface<real> sf;
    if constexpr (std::is_same_v<ProgT, MixedHead>) make_surface_mixed(hit_rec, 0u, L.best_t, L.ro, L.rd, L.tm, sf);  // lean MIXED program
    else make_surface<real, FEAT>(hit_rec, sc, mats, 0u, L.best_t, L.ro, L.rd, L.tm, sf);                             // slot or COMPACT program
    RTK_SHADE_PROF(0)
    return shade_surface<real, FEAT>(L, sf, sc, mats, cnt RTK_SHADE_PROF_ARG);
}

RTK_DEV uint8_t to_byte(double x) {  // Camera.txt:29-34,77-83
    double g = x > 0 ? __builtin_sqrt(x) : 0.0;
    g = g < 0.000 ? 0.000 : (g > 0.999 ? 0.999 : g);
    return uint8_t(int(255.999 * g));
}

// ------------------------------------------------------------------ kernel ----
// Persistent waves + a wave-level ballot scheduler.
//
// Each workgroup first stages the traversal program from HBM into LDS with
// coalesced 16-byte loads (when IN_LDS: the host checked it fits); then every
// wave repeatedly pulls the next work item -- an 8x8 tile of this rank x a chunk of
// 8 samples -- from an atomic counter.  A lane owns one pixel of the item and walks
// that chunk's samples in order; the resolve kernel adds a pixel's chunks in index
// order.  The sum is therefore a fixed function of (scene, camera, seed, pixel) --
// the image does not depend on which wave, workgroup or GPU rendered what -- but it
// is NOT the reference's left-to-right sum over all samples (Camera.txt:70-73): with
// chunked partial sums and the iterative throughput product the linear image agrees
// with the reference to ~1e-17 (asserted < 1e-12), and only the u8 bytes are equal.
//
// Paths have 1..max_depth segments and rays visit 1..100+ program records, so a
// lock-step "all lanes trace, then all lanes shade" loop leaves ~90 % of the
// lanes idle.  Instead each lane is a small state machine (Lane<real>) and in
// every iteration the wave VOTES (v_cmp + s_bcnt1 on the 64-bit ballots) on which
// kind of work has the most lanes ready -- box test, sphere test, another record
// kind, or shade/regenerate -- and runs only that, under a wave-uniform branch.
// Lanes waiting for a different kind keep their state and join a later vote; a
// lane whose path ends starts its next sample in the same shade step, so nobody
// waits for the longest path in the wave.  Scheduling changes only the
// interleaving between lanes, never a lane's own arithmetic, so the image is
// bit-identical to a lock-step execution.
enum Want : int { W_DONE = 0, W_BOX = 1, W_SPHERE = 2, W_OTHER = 3, W_SHADE = 4, W_QUAD = 5, W_TRI = 6 };

// Lane count of a ballot as a 32-bit scalar, one s_bcnt1_i32_b64.  With __builtin_popcountll the compiler
// carries the count as i64 and performs the vote's comparisons on the vector unit (v_cmp_lt_u64).
// A wave-uniform condition as a scalar branch: the compiler cannot always prove uniformity and would otherwise turn
// the branch into exec-mask bookkeeping.
RTK_DEV bool uniform(bool c) { return __builtin_amdgcn_readfirstlane(int(c)) != 0; }

RTK_DEV int popcount64(unsigned long long m) {
    int n;
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(n) : "s"(m) : "scc");
    return n;
}

// Diagnostic build only (tools/profile_phases.py compiles this file with -DRTK_PROFILE into a separate
// library): s_memtime stamps at the scheduler's phase boundaries; per phase the wave adds its cycles,
// step count and active-lane count to counters[3*phase .. 3*phase+2].  The product build has none of it.
#ifdef RTK_PROFILE
#define RTK_PROF_DECL unsigned long long prof_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, prof_n[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, prof_l[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
    unsigned long long prof_prev = __builtin_amdgcn_s_memtime();                                                                       \
    const unsigned long long prof_wall0 = wall_clock64();                                                                              \
    unsigned long long prof_wall_empty = 0, prof_chunk_t0 = 0;                                                                         \
    ShadeProf sprof;
#define RTK_PROF_CHUNK_BEGIN prof_chunk_t0 = wall_clock64();
#define RTK_PROF_CHUNK_END /* (pixel, chunk)s that ended after the work queue ran dry and took over 300 us: [31] count, then {begin, duration, segments, slot << 8 | pixel} from [32] on */ \
    {                                                                                                                                  \
        const unsigned long long dur_ = wall_clock64() - prof_chunk_t0;                                                                \
        if (prof_wall_empty != 0 && dur_ > 30000ull) {                                                                                                       \
            const unsigned long long i_ = atomicAdd(&counters[31], 1ull);                                                              \
            if (i_ < 4096ull) {                                                                                                        \
                counters[32 + 4 * i_] = prof_chunk_t0;                                                                                 \
                counters[33 + 4 * i_] = dur_;                                                                                          \
                counters[34 + 4 * i_] = (unsigned long long)(L.segs);                                                                  \
                counters[35 + 4 * i_] = ((unsigned long long)(my_slot) << 8) | (unsigned long long)(my_pix);                           \
            }                                                                                                                          \
        }                                                                                                                              \
    }
#define RTK_PROF_QUEUE_EMPTY \
    if (prof_wall_empty == 0) prof_wall_empty = wall_clock64();
#define RTK_PROF_MARK(phase, steps, lanes)                           \
    {                                                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        prof_t[phase] += now_ - prof_prev;                           \
        prof_n[phase] += (steps);                                    \
        prof_l[phase] += (lanes);                                    \
        prof_prev = now_;                                            \
    }
#define RTK_PROF_FLUSH                                                        \
    if (lane == 0)                                                            \
        for (int k_ = 0; k_ < 4; k_++) {                                      \
            atomicAdd(&counters[kShadeProfBase + 2 * k_], sprof.t[k_]);       \
            atomicAdd(&counters[kShadeProfBase + 2 * k_ + 1], sprof.n[k_]);   \
        }                                                                     \
    if (lane == 0)                                                            \
        for (int ph_ = 0; ph_ < 10; ph_++) { /* phases 6, 7 (parts of the shade step) live at [25..30], 8 and 9 behind the shade sub-phases */ \
            const int at_ = ph_ < 6 ? 3 * ph_ : (ph_ < 8 ? 25 + 3 * (ph_ - 6) : kShadeProfBase + 8 + 3 * (ph_ - 8)); \
            atomicAdd(&counters[at_], prof_t[ph_]);                           \
            atomicAdd(&counters[at_ + 1], prof_n[ph_]);                       \
            atomicAdd(&counters[at_ + 2], prof_l[ph_]);                       \
        }                                                                     \
    if (lane == 0) { /* wave lifetimes on the 100 MHz wall clock: [18] sum, [19] max, [20] sum of the time after the queue ran dry, [21] its max, [22] waves, [23] 2^62 - first start, [24] last end */ \
        const unsigned long long end_ = wall_clock64();                       \
        if (prof_wall_empty == 0) prof_wall_empty = end_;                     \
        atomicAdd(&counters[18], end_ - prof_wall0);                          \
        atomicMax(&counters[19], end_ - prof_wall0);                          \
        atomicAdd(&counters[20], end_ - prof_wall_empty);                     \
        atomicMax(&counters[21], end_ - prof_wall_empty);                     \
        atomicAdd(&counters[22], 1ull);                                       \
        atomicMax(&counters[23], (1ull << 62) - prof_wall0);                  \
        atomicMax(&counters[24], end_);                                       \
    }
#else
#define RTK_PROF_DECL
#define RTK_PROF_QUEUE_EMPTY
#define RTK_PROF_CHUNK_BEGIN
#define RTK_PROF_CHUNK_END
#define RTK_PROF_MARK(phase, steps, lanes)
#define RTK_PROF_FLUSH
#endif

// A finished (pixel, chunk): its sum of ray_color values goes to partial[item][3][64].
template <typename real>
RTK_DEV void store_partial(real* __restrict__ partial, int item, int pix, V3<real> sum) {
    real* base = partial + size_t(item) * 192 + pix;
    base[0] = sum.x;
    base[64] = sum.y;
    base[128] = sum.z;
}

#ifndef RTK_SPLIT_MATERIALS_IN_LDS
#define RTK_SPLIT_MATERIALS_IN_LDS 1
#endif
#ifndef RTK_THREADS_ALL_F64
#define RTK_THREADS_ALL_F64 768   // workgroup bound of the full-feature f64 kernels: 3 waves per SIMD, 168 VGPRs (A/B builds: 512 = 2 waves, 1024 = 4)
#endif
// Workgroup-size bound = register budget: 1024 threads -> 4 waves per SIMD, 128 VGPRs; 768 -> 3 waves, 168 VGPRs;
// 512 -> 2 waves, 256 VGPRs.  Measured on C2 (lean f64 kernel): 2 waves 93.8 ms, 3 waves 73.5 ms, 4 waves 68.5 ms
// per frame (the 4-wave build spills a few values in the shade path).  Same-box A/B for the other f64 kernels
// (tools/ab/run_ab.sh): quad/box subset on C3 43.3 ms at 4 waves vs 48.7 at 3; mesh subset on C4 no difference;
// the full-feature kernel, which needs far more registers (~560 B/lane of spills at 168), is fastest at 2 waves.
#ifndef RTK_THREADS_MESH_F64
#define RTK_THREADS_MESH_F64 1024   // workgroup bound of the f64 mesh kernels: 4 waves per SIMD, 128 VGPRs (C4 60.1 -> 53.0 ms; 768 = 3 waves)
#endif
// (round 3) the f64 mesh kernels run at four waves per SIMD when the whole program is staged in LDS (C4 60.1 -> 53.0 ms, 44 B of
// scratch); with the program, or all but its boxes, in memory a fourth wave only adds pressure on L2 (C4, program in global
// memory: 23.2 ms at three waves, 33.6 at four), so those keep 768 threads.
template <typename real>
constexpr int max_threads_of(uint32_t feat, bool in_lds) {
    const uint32_t scene_feat = feat & ~uint32_t(F_FMA_BOX | F_F32_BOX | F_MATTE | F_LDS_BOXES | F_SPHERE_MEDIA_ONLY);
    // ... and the full-feature f64 kernels run at three waves where the traversal data is in LDS -- the whole program, or the
    // hot part of a COMPACT program (F_LDS_BOXES | F_F32_BOX: C5 at 16 spp 21.1 -> 17.7 ms) -- and stay at two where records
    // come from memory (program in global memory 25.5 vs 29.1 ms at three; slot program with its boxes in LDS 25.6 vs 29.0).
    const bool hot_cold = (feat & F_LDS_BOXES) != 0 && (feat & F_F32_BOX) != 0;
    if (sizeof(real) == 8)
        return scene_feat == kFeatAll ? ((in_lds || hot_cold) ? RTK_THREADS_ALL_F64 : 512)
                                      : ((scene_feat == kFeatLean || scene_feat == kFeatQuadBox) ? 1024 : (in_lds ? RTK_THREADS_MESH_F64 : 768));
    return 768;
}
template <typename real, uint32_t FEAT, bool IN_LDS>
constexpr int max_threads() {
    return max_threads_of<real>(FEAT, IN_LDS);
}

template <typename real, uint32_t FEAT_ALL, bool COUNT, bool IN_LDS>
__global__ __launch_bounds__((max_threads<real, FEAT_ALL, IN_LDS>())) void rtk_render_kernel(SceneView<real> sc, const CameraRec<real>* __restrict__ cam_ptr, TileMap tmap, uint32_t seed,
                                                          real* __restrict__ partial, unsigned long long* __restrict__ counters,
                                                          unsigned int* __restrict__ tile_counter, const int32_t* __restrict__ tile_order,
                                                          unsigned int* __restrict__ tile_cost, uint32_t diag) {
    extern __shared__ __align__(16) unsigned char lds_program[];
#ifndef RTK_CH_LDS_ABS
#define RTK_CH_LDS_ABS 1
#endif
    if constexpr (RTK_CH_BOX && RTK_CH_BYTE_PC && RTK_CH_LDS_ABS && IN_LDS) {
        // (lean MIXED kernel: byte pcs are used as LDS addresses, see rec_at) -- render nothing rather than garbage otherwise
        typedef const unsigned char __attribute__((address_space(3))) * lds_bytes_t;
        if (uint32_t(size_t((lds_bytes_t)lds_program)) != 0u) return;
    }
#ifndef RTK_DEV_MASK_OFF
#define RTK_DEV_MASK_OFF 0u   // register-pressure experiments (tools/kernel_resources.py): feature bits compiled out of every kernel
#endif
    constexpr bool BRACKET = (FEAT_ALL & F_SPHERE_MEDIA_ONLY) == 0;  // generic media (OP_MED_BEGIN / MID / END) may occur in the program
    // (measured and not done: also compiling get_lighting and the radiance registers out of the restricted variant -- 80 instead
    // of 96 B of scratch, and C5 34.6 instead of 32.0 ms at 32 spp)
    constexpr uint32_t FEAT = FEAT_ALL & ~uint32_t(F_LDS_BOXES | F_SPHERE_MEDIA_ONLY) & ~uint32_t(RTK_DEV_MASK_OFF);
    constexpr bool LDS_PART = (FEAT_ALL & F_LDS_BOXES) != 0;  // a program larger than LDS, part of it staged there
    static_assert(!LDS_PART || !IN_LDS, "F_LDS_BOXES: for programs that do not fit LDS");
    constexpr bool MIXED = (FEAT & F_F32_BOX) != 0;  // f32 culling boxes + exact primitives (f64 kernels, fast order): ...
    // ... the MIXED program of sphere-only scenes (32-byte units) or, for every other family, the COMPACT program (16-byte units)
    constexpr bool COMPACT = MIXED && (FEAT & ~uint32_t(F_F32_BOX | F_MATTE)) != kFeatLean;
    static_assert(!MIXED || sizeof(real) == 8, "F_F32_BOX: f64 kernels only");
    // ... which part depends on the layout.  Slot program (SPLIT): the box slots, the kind nibbles and a rank table.  COMPACT
    // program (COLD): the whole "hot" program -- f32 culling boxes, spheres, every rare record -- in which each run of quads
    // or triangles is ONE two-unit record {kind, count, first unit in the cold array}; the quads and triangles themselves
    // ("cold": 144 / 80 bytes each, tested a few times per sample) stay in memory (SceneView::program_cold).
    constexpr bool CH = RTK_CH_BOX && MIXED && !COMPACT;  // the MIXED program's boxes are centre / half-extent records
    constexpr bool CHE = COMPACT && kCompactChStatic<FEAT>;  // ... and so are the COMPACT programs' of the mesh family, with a per-ray slack
    constexpr bool CHE_RT = RTK_CH_COMPACT && COMPACT && COUNT && !CHE;  // the counting kernel on any family's COMPACT program: asks the scene
    [[maybe_unused]] const bool che_rt = CHE_RT && sc.compact_ch != 0;
    constexpr int kCulling = CH ? 1 : (CHE ? 2 : (CHE_RT ? 3 : 0));
    constexpr bool SPLIT = LDS_PART && !COMPACT;
    constexpr bool COLD = LDS_PART && COMPACT;
    static_assert(!LDS_PART || !MIXED || COMPACT, "F_LDS_BOXES with f32 boxes: the COMPACT program");
    constexpr bool XF = (FEAT & F_XFORM) != 0;
    using ProgRec = std::conditional_t<COMPACT, Unit16, std::conditional_t<MIXED, MixedHead, Slot<real>>>;  // what pc counts
    using CurRec = std::conditional_t<MIXED, MixedHead, Slot<real>>;                                      // the record head a step holds in registers
    constexpr uint32_t kBoxUnits = COMPACT ? 2u : 1u, kSphereUnits = COMPACT ? 3u : (MIXED ? 2u : 1u), kQuadUnits = COMPACT ? 9u : 3u, kTriUnits = COMPACT ? 5u : 2u;
    const ProgRec* prog;
    if constexpr (COLD) prog = sc.program_hot;
    else if constexpr (COMPACT) prog = sc.program_compact;
    else if constexpr (MIXED) prog = sc.program_mixed;
    else prog = sc.program;
    const int n_records = COLD ? sc.n_hot_units : (COMPACT ? sc.n_units16 : (MIXED ? sc.n_units : sc.n_slots));  // units of sizeof(ProgRec)
    [[maybe_unused]] const Unit16* cold = nullptr;
    if constexpr (COLD) cold = sc.program_cold;
    const MaterialRec<real>* mats = sc.materials;
    if constexpr (IN_LDS || COLD) {  // program (COLD: its hot part), then the material table, both as 16-byte words
        const int n_prog16 = n_records * int(sizeof(ProgRec) / 16);
        const int n_mat16 = sc.n_materials * int(sizeof(MaterialRec<real>) / 16);
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(prog);
        const uint4* __restrict__ msrc = reinterpret_cast<const uint4*>(sc.materials);
        uint4* dst = reinterpret_cast<uint4*>(lds_program);
        for (int k = threadIdx.x; k < n_prog16; k += blockDim.x) dst[k] = src[k];
        for (int k = threadIdx.x; k < n_mat16; k += blockDim.x) dst[n_prog16 + k] = msrc[k];
        if constexpr (COLD && (FEAT & F_TEXTURE) != 0) {  // the Perlin tables as well when the launcher found room
            if (tmap.perlin_lds_offset > 0) {
                const int n16 = sc.n_perlins * int(sizeof(PerlinRec<real>) / 16);
                const uint4* __restrict__ psrc = reinterpret_cast<const uint4*>(sc.perlins);
                uint4* pdst = reinterpret_cast<uint4*>(lds_program + tmap.perlin_lds_offset);
                for (int k = threadIdx.x; k < n16; k += blockDim.x) pdst[k] = psrc[k];
                sc.perlins = reinterpret_cast<const PerlinRec<real>*>(lds_program + tmap.perlin_lds_offset);
            }
        }
        __syncthreads();
        prog = reinterpret_cast<const ProgRec*>(lds_program);
        mats = reinterpret_cast<const MaterialRec<real>*>(lds_program + size_t(n_prog16) * 16);
    }
    using BoxCacheRec = BoxRec<real>;  // what F_LDS_BOXES keeps per box
    [[maybe_unused]] const BoxCacheRec* lds_boxes = nullptr;
    [[maybe_unused]] const uint32_t* lds_kinds = nullptr;
    [[maybe_unused]] const uint2* lds_rank = nullptr;
    if constexpr (SPLIT) {
        const BoxCacheRec* cache = sc.box_cache;
        const uint32_t* kind_words = sc.kind_words;
        const uint2* box_rank = sc.box_rank;
        const int n_boxes = sc.n_cached_boxes, n_kind_words = sc.n_kind_words, n_rank_words = sc.n_rank_words;
        const size_t box_bytes = size_t(n_boxes) * sizeof(BoxCacheRec);  // a multiple of 8
        const uint2* __restrict__ src = reinterpret_cast<const uint2*>(cache);
        uint2* dst = reinterpret_cast<uint2*>(lds_program);
        for (int k = threadIdx.x; k < int(box_bytes / 8); k += blockDim.x) dst[k] = src[k];
        uint32_t* kdst = reinterpret_cast<uint32_t*>(lds_program + box_bytes);
        for (int k = threadIdx.x; k < n_kind_words; k += blockDim.x) kdst[k] = kind_words[k];
        uint2* rdst = reinterpret_cast<uint2*>(lds_program + box_bytes + ((size_t(n_kind_words) * 4 + 7) & ~size_t(7)));
        for (int k = threadIdx.x; k < n_rank_words; k += blockDim.x) rdst[k] = box_rank[k];
#if RTK_SPLIT_MATERIALS_IN_LDS
        // ... and the material table behind them when the launcher found room (tmap.mats_lds_offset > 0): every shade step reads it
        if (tmap.mats_lds_offset > 0) {
            const int n_mat16 = sc.n_materials * int(sizeof(MaterialRec<real>) / 16);
            const uint4* __restrict__ msrc = reinterpret_cast<const uint4*>(sc.materials);
            uint4* mdst = reinterpret_cast<uint4*>(lds_program + tmap.mats_lds_offset);
            for (int k = threadIdx.x; k < n_mat16; k += blockDim.x) mdst[k] = msrc[k];
            mats = reinterpret_cast<const MaterialRec<real>*>(lds_program + tmap.mats_lds_offset);
        }
#endif
        if constexpr ((FEAT & F_TEXTURE) != 0) {  // the Perlin tables too, when the launcher found room (tmap.perlin_lds_offset > 0)
            if (tmap.perlin_lds_offset > 0) {
                const int n16 = sc.n_perlins * int(sizeof(PerlinRec<real>) / 16);
                const uint4* __restrict__ psrc = reinterpret_cast<const uint4*>(sc.perlins);
                uint4* pdst = reinterpret_cast<uint4*>(lds_program + tmap.perlin_lds_offset);
                for (int k = threadIdx.x; k < n16; k += blockDim.x) pdst[k] = psrc[k];
                sc.perlins = reinterpret_cast<const PerlinRec<real>*>(lds_program + tmap.perlin_lds_offset);  // texture_value reads through sc
            }
        }
        __syncthreads();
        lds_boxes = reinterpret_cast<const BoxCacheRec*>(lds_program);
        lds_kinds = kdst;
        lds_rank = rdst;
    }
    // the head of the record that starts at pc, as a step holds it in registers
    // (COLD: a lane inside a run of cold primitives keeps its position in the run in the top byte of its pc)
    constexpr uint32_t kPcMask = COLD ? 0x00FFFFFFu : 0xFFFFFFFFu;
    constexpr uint32_t kPcUnit = CH ? kChPcUnit : 1u;  // what one unit of the program adds to a pc (CH: pcs count bytes)
    // the record that starts at pc
    auto rec_at = [&](uint32_t pc) -> const ProgRec* {
        if constexpr (CH && RTK_CH_BYTE_PC && IN_LDS && RTK_CH_LDS_ABS) {
            // The staged program starts at LDS address 0 -- this kernel declares no static LDS, so its dynamic segment does
            // (checked when the kernel starts, and on the code object by tests/test_abi_and_host.py) -- hence a byte pc IS
            // the LDS address: the box step's next read needs no address arithmetic at all (C2 19.35 -> 18.96 ms).
            typedef const unsigned char __attribute__((address_space(3))) * lds_bytes_t;
            return reinterpret_cast<const ProgRec*>((const unsigned char*)(lds_bytes_t)(size_t)pc);
        } else if constexpr (CH && RTK_CH_BYTE_PC) return reinterpret_cast<const ProgRec*>(reinterpret_cast<const unsigned char*>(prog) + pc);
        else return prog + pc;
    };
    auto head_at = [&](uint32_t pc) -> CurRec {
        if constexpr (COMPACT) return *reinterpret_cast<const MixedHead*>(prog + (pc & kPcMask));
        else return *rec_at(pc);
    };
    // kind of the record that starts at pc; the box record at pc (SPLIT: from the LDS copies)
    auto kind_of = [&](uint32_t pc) -> uint32_t {
        if constexpr (SPLIT) return (lds_kinds[pc >> 3] >> ((pc & 7u) * 4u)) & 15u;
        else if constexpr (COMPACT) return prog[(pc & kPcMask) + 1].w[2] & 15u;
        else return rec_at(pc)->kind_payload & 15u;
    };
    [[maybe_unused]] auto box_at = [&](uint32_t pc) -> CurRec {
        const uint2 e = lds_rank[pc >> 5];
        const uint32_t at = e.y + uint32_t(__builtin_popcount(e.x & ((1u << (pc & 31u)) - 1u)));
        if constexpr (MIXED) {
            return CurRec{};  // (the f32-box programs are only used when they fit LDS whole: no boxes-in-LDS kernel)
        } else {
            const BoxRec<real> b = lds_boxes[at];
            Slot<real> s;
#pragma unroll
            for (int k = 0; k < 6; k++) s.v[k] = b.v[k];
            s.kind_payload = OP_BOX;
            s.aux = b.aux;
            return s;
        }
    };
    // The record a closest hit names (L.best_pc): a pc of the program, or -- COLD -- n_records + the unit of a cold primitive.
    auto record_of = [&](uint32_t id) -> const ProgRec* {
        if constexpr (COLD) return id < uint32_t(n_records) ? prog + id : reinterpret_cast<const ProgRec*>(cold + (id - uint32_t(n_records)));
        else return rec_at(id);
    };
    [[maybe_unused]] auto kind_of_hit = [&](uint32_t id) -> uint32_t {  // for the tie rule: the kind of the current winner
        if constexpr (COLD) return id < uint32_t(n_records) ? kind_of(id) : (cold[id - uint32_t(n_records) + 1].w[2] & 15u);
        else return kind_of(id);
    };
    // exact ties between primitives are resolved by the reference's ranks in the kernels that run re-grouped hierarchies
    // (the quad/box subset kernel serves the fast order without a flag of its own)
    // f64 kernels only: the float kernels are not bit-exact against the reference anyway, and the lean one has no register to spare
// Wave priorities (s_setprio) by scheduler phase.  The SIMD's instruction arbiter serves the higher priority first (then the
// older wave); waves inside a traversal loop -- short dependent chains, an LDS round trip per step -- lose issue slots to
// waves that are in a shade step (long independent runs of arithmetic) exactly when a stall costs them most.  Measured on
// C2 (tools/ab, same box): no priorities 21.93-22.09 ms; box loop 1 / 2 / 3: 21.55 / 21.53 / 21.52; box 2 + sphere loop 1:
// 21.45-21.49 (kept); raising the shade step instead: 22.5.
#ifndef RTK_AB_BOX_PRIO
#define RTK_AB_BOX_PRIO 2
#endif
#ifndef RTK_AB_PRIM_PRIO
#define RTK_AB_PRIM_PRIO 1   // quad and triangle loops
#endif
#ifndef RTK_AB_SPH_PRIO
#define RTK_AB_SPH_PRIO 1
#endif
#ifndef RTK_AB_SHADE_PRIO
#define RTK_AB_SHADE_PRIO 0
#endif
#ifndef RTK_AB_NO_TIE
#define RTK_AB_NO_TIE 0   // tools/ab: what the tie rule costs
#endif
    constexpr bool TIE = !RTK_AB_NO_TIE && sizeof(real) == 8 && ((FEAT & (F_FMA_BOX | F_F32_BOX)) != 0 || (FEAT & ~uint32_t(F_MATTE)) == kFeatQuadBox);
    const TieCtx<TIE, decltype(kind_of_hit)> tie{TIE ? (COLD ? sc.tie_rank_hot : (MIXED ? sc.tie_rank : sc.tie_rank_slot)) : nullptr, kind_of_hit, (CH && RTK_CH_BYTE_PC) ? 5u : 0u};
    // The hand-out order of the tiles (learned from the previous frame) is staged behind the program when the host
    // found room for it (tmap.order_in_lds): a lookup per work item from LDS instead of a cold global load.
    if constexpr ((FEAT & F_XFORM) != 0 && (IN_LDS || LDS_PART)) {
        if (tmap.chains_lds_offset > 0) {  // instance-transform chains (a few hundred bytes: word by word)
            const int n4 = sc.n_chains * int(sizeof(ChainRec<real>) / 4);
            const uint32_t* __restrict__ csrc = reinterpret_cast<const uint32_t*>(sc.chains);
            uint32_t* cdst = reinterpret_cast<uint32_t*>(lds_program + tmap.chains_lds_offset);
            for (int k = threadIdx.x; k < n4; k += blockDim.x) cdst[k] = csrc[k];
            __syncthreads();
            sc.chains = reinterpret_cast<const ChainRec<real>*>(lds_program + tmap.chains_lds_offset);
        }
    }
    const int32_t* lds_order = nullptr;
    if (tmap.order_in_lds && tile_order) {
        int32_t* dst_order = reinterpret_cast<int32_t*>(lds_program + tmap.order_lds_offset);
        for (int k = threadIdx.x; k < tmap.n_tiles_local; k += blockDim.x) dst_order[k] = tile_order[k];
        __syncthreads();
        lds_order = dst_order;
    }
    // The camera lives in device memory and is read with scalar loads where it is
    // used (once per sample); as a by-value argument it would sit in registers for
    // the whole kernel.
    const CameraRec<real>& cam = *cam_ptr;
    const int lane = threadIdx.x & 63;
    const uint32_t seed_hash = pcg_hash(seed);
    const int n_tiles_total = tmap.tiles_x * tmap.tiles_y;
    const int width = cam.width, height = cam.height, spp = cam.spp;
    const uint32_t end_pc = uint32_t(n_records - (COMPACT ? 2 : 1)) * kPcUnit;  // OP_END: the last record (two units in the COMPACT layout)
    const float extent = sc.extent;
    Counters<COUNT> cnt;
    cnt.clear();

    // Wave-uniform hand-out state: pixels of work item `refill_item` from
    // `refill_next` on have not been given to a lane yet.
    const int n_items = tmap.n_tiles_local * tmap.n_chunks;
    int refill_item = 0, refill_next = 64;
    // The index of the NEXT work item is fetched one item ahead: the returning atomic is issued when an item is
    // taken and only waited for when its 64 pixels have been handed out, so its ~2 us round trip never stalls the wave.
    unsigned int prefetched_item = 0;
    if (lane == 0) prefetched_item = atomicAdd(tile_counter, 1u);
    bool exhausted = false;
    // tools/: A/B of the refill batch size through variant bits 14..16 (0 = default)
    constexpr int kRefillMinTable[8] = {4, 1, 4, 8, 16, 24, 32, 12};
    const int refill_min = kRefillMinTable[(diag >> 14) & 7];
    constexpr int kSphereMinTable[8] = {16, 65, 4, 8, 12, 16, 24, 32};  // variant bits 17..19; 65 = never (spheres only through the vote)
    const int sphere_sel = int(diag >> 17) & 7;
    // defaults per kernel family (tools/variant_sweep.py): MIXED 12 (24.5 vs 24.7 ms at 16); full-feature 4 (C5 297.3 -> 283.4 ms; 8: 285.7);
    // mesh subset 8 (C4 +1 %); 16 elsewhere
    constexpr uint32_t kFamily = FEAT & ~uint32_t(F_FMA_BOX | F_F32_BOX | F_MATTE);
    constexpr int kSphereMinDefault = (MIXED && !COMPACT) ? 12 : (kFamily == kFeatAll ? 4 : (kFamily == kFeatMesh ? 8 : 16));
    const int sphere_min = sphere_sel == 0 ? kSphereMinDefault : kSphereMinTable[sphere_sel];

    Lane<real> L;
    L.pc = end_pc;
    L.kind = OP_DEAD;                 // this lane owns no (pixel, chunk): it takes part in no vote and waits for a refill
    L.box_kind = OP_BOX;              // (never left undefined: the vote compares kind with it on every lane)
    L.sum = mk(real(0), real(0), real(0));
    L.s = 0;
    RTK_PROF_DECL
    int my_slot = 0, my_pix = 0, px_i = 0, px_j = 0, s_end = 0, cost_tile = -1;  // my_slot = chunk * n_tiles_local + local_tile: where the partial sum goes

    // COLD: one primitive of the run the lane sits on.  The run record (hot, two units) = {kind, count | aux = first unit
    // in the cold array}; the lane's position in the run rides in the top byte of its pc.  A hit is recorded under the id
    // n_records + cold unit (record_of / the rank table understand it); after the last primitive the lane walks on in
    // the hot program.
    [[maybe_unused]] auto cold_step = [&](auto is_quad, uint32_t& k) {
        if constexpr (COLD) {
            constexpr bool QUAD = decltype(is_quad)::value;
            constexpr uint32_t units = QUAD ? 9u : 5u;
            const uint32_t hot = L.pc & kPcMask, sub = L.pc >> 24;
            const MixedHead run = head_at(hot);
            const uint32_t at = run.aux + sub * units;
            L.pc = uint32_t(n_records) + at;
#ifndef RTK_COLD_QUAD_PREFETCH
#define RTK_COLD_QUAD_PREFETCH 1
#endif
            if constexpr (QUAD && RTK_COLD_QUAD_PREFETCH) {
                QuadRegs regs;
                const double* __restrict__ src = reinterpret_cast<const double*>(cold + at);
#pragma unroll
                for (int e = 0; e < 18; e++) regs.q[e] = src[e];
                hit_quad<XF, MIXED>(L, &regs, 0u, cnt, tie);
            } else if constexpr (QUAD) {
                hit_quad<XF, MIXED>(L, reinterpret_cast<const ProgRec*>(cold + at), 0u, cnt, tie);
            } else {
                hit_tri<XF, MIXED>(L, reinterpret_cast<const ProgRec*>(cold + at), 0u, cnt, tie);
            }
            const bool last = sub + 1u >= (run.kind_payload >> 4);
            L.pc = last ? hot + 2u : (hot | ((sub + 1u) << 24));
            k = last ? kind_of(L.pc) : uint32_t(QUAD ? OP_QUAD : OP_TRI);
            L.kind = k;
        }
    };
    // One refill round: idle lanes take the next pixels of the wave's current work item (a tile x a chunk of the
    // samples); when the item is used up the wave first pulls another one from the rank-wide counter.  A (pixel,
    // chunk) belongs to exactly one lane, which walks its samples in order.  Returns true when idle lanes remain that
    // were not offered a pixel (the item ran out first).  Everything that steers it is wave-uniform.
    // What is the same for all 64 pixels of a work item is computed when the item is taken, on the scalar unit: the
    // integer divisions that turn the item index into (tile, chunk, pixel origin) and the tile_order lookup used to be
    // redone -- and their latencies waited for -- in every refill round (~12 lanes each, 4.3 M rounds per C2 frame).
    int it_x0 = 0, it_y0 = 0, it_slot = 0, it_s_begin = 0, it_s_end = 0, it_cost_tile = -1;
    bool it_ok = false;
    auto hand_out = [&](unsigned long long m_idle) -> bool {
        if (refill_next >= 64) {
            const int t = int(__builtin_amdgcn_readfirstlane(prefetched_item));
            if (t >= n_items) {
                exhausted = true;
                RTK_PROF_QUEUE_EMPTY
                return false;
            }
            refill_item = t;
            refill_next = 0;
            if (lane == 0) prefetched_item = atomicAdd(tile_counter, 1u);
            // Items are handed out in `tile_order` when the host has one (most expensive tiles of the previous
            // frame first); which wave renders a tile, and when, never changes a pixel's value.
            const int position = t / tmap.n_chunks, chunk = t % tmap.n_chunks;
            int local_tile = position;
            if (tile_order) local_tile = lds_order ? int(__builtin_amdgcn_readfirstlane(lds_order[position])) : tile_order[position];
            const int tile = local_tile * tmap.n_ranks + tmap.rank;
            it_slot = chunk * tmap.n_tiles_local + local_tile;
            it_x0 = (tile % tmap.tiles_x) * 8;
            it_y0 = (tile / tmap.tiles_x) * 8;
            it_s_begin = tmap.chunk_start[chunk];
            it_s_end = tmap.chunk_start[chunk + 1];
            it_cost_tile = chunk == 0 ? local_tile : -1;  // chunk 0 of every pixel reports the tile's cost
            it_ok = tile < n_tiles_total && it_s_begin < it_s_end;
        }
        const int avail = 64 - refill_next;
        const int rank_in_idle = int(__builtin_amdgcn_mbcnt_hi(uint32_t(m_idle >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m_idle), 0u)));
        if (L.kind == OP_DEAD && rank_in_idle < avail) {
            my_pix = refill_next + rank_in_idle;
            my_slot = it_slot;
            px_i = it_x0 + (my_pix & 7);
            px_j = it_y0 + (my_pix >> 3);
            s_end = it_s_end;
            if (it_ok && px_i < width && px_j < height) {
                L.sum = mk(real(0), real(0), real(0));
                L.s = it_s_begin;
                L.segs = 0;
                cost_tile = it_cost_tile;
                RTK_PROF_CHUNK_BEGIN
                begin_sample(L, cam, px_i, px_j, seed_hash, cnt);
                if (L.depth > 0) begin_segment<(FEAT & F_XFORM) != 0, MIXED, kCulling>(L, cnt, extent);
                else L.pc = end_pc;  // max_depth == 0: ray_color returns black at once (Camera.txt:205-206)
                L.kind = kind_of(L.pc);
            }
        }
        const int n_idle = popcount64(m_idle);
        refill_next += n_idle < avail ? n_idle : avail;
        return n_idle > avail;
    };

    for (;;) {
        // ---- regeneration at pixel granularity, BATCHED: handing out one pixel costs a full begin_sample +
        // begin_segment (hashes, RNG draws, the lens rejection loop, three f64 divisions) executed by the whole wave,
        // so idle lanes wait until `refill_min` of them can be served at once (or nobody has work left).  Which lane
        // renders a pixel, and when, never changes its value.  At most two rounds: the rest of the current item,
        // then the head of the next one.
        {
            const unsigned long long m_idle = __ballot(L.kind == OP_DEAD);
            const int n_idle = popcount64(m_idle);
            if (uniform(!exhausted && (n_idle >= refill_min || n_idle == 64))) {
                RTK_PROF_MARK(0, 0, 0)
                if (uniform(hand_out(m_idle))) hand_out(__ballot(L.kind == OP_DEAD));
                RTK_PROF_MARK(4, 1, n_idle)  // profile build: refills are booked under "other op" (phase 4)
            }
        }

        // ---- vote (registers and scalar unit only).  A lane's record kind IS its vote; OP_DEAD lanes have none.
        const uint32_t kind = L.kind;
        const unsigned long long m_box = __ballot(kind == L.box_kind);  // irregular rays' boxes count as "other"
        const unsigned long long m_sph = __ballot(kind == OP_SPHERE);
        const unsigned long long m_shd = __ballot(kind == OP_END);
        const unsigned long long m_quad = (FEAT & F_QUAD) ? __ballot(kind == OP_QUAD) : 0ull;
        const unsigned long long m_tri = (FEAT & F_TRI) ? __ballot(kind == OP_TRI) : 0ull;
        const unsigned long long m_oth = __ballot(kind != OP_DEAD) & ~(m_box | m_sph | m_shd | m_quad | m_tri);
        if ((m_box | m_sph | m_oth | m_shd | m_quad | m_tri) == 0ull) {
            if (exhausted) break;
            continue;
        }
        const int n_box = popcount64(m_box), n_sph = popcount64(m_sph), n_oth = popcount64(m_oth), n_shd = popcount64(m_shd);
        const int n_quad = popcount64(m_quad), n_tri = popcount64(m_tri);
        RTK_PROF_MARK(0, 1, n_box + n_sph + n_oth + n_shd + n_quad + n_tri)
        // The kind with the most ready lanes wins; ties go to the cheaper kind (order of the tests).  Written as
        // nested comparisons on purpose: a running "pick/most" update chain compiled to a 9 % slower kernel
        // (tools/ab/run_ab.sh A/B on the same box: 78.5 ms vs 70.2 ms per C2 frame).
        // (round 3: a shortcut for the common case -- 32 or more lanes on box records are a majority, so the other five ballots need
        // not be counted -- left the pick unchanged and the lean kernel with 128 VGPRs + 28 B of scratch instead of 110 + 0
        // (compile-only, tools/kernel_resources.py): not built.  The vote's cost is the wait for the lanes' record kinds, not its
        // thirty instructions.)
        int pick;
        if (n_box >= n_sph && n_box >= n_quad && n_box >= n_tri && n_box >= n_shd && n_box >= n_oth) pick = W_BOX;
        else if (n_sph >= n_quad && n_sph >= n_tri && n_sph >= n_shd && n_sph >= n_oth) pick = W_SPHERE;
        else if (n_quad >= n_tri && n_quad >= n_shd && n_quad >= n_oth) pick = W_QUAD;
        else if (n_tri >= n_shd && n_tri >= n_oth) pick = W_TRI;
        else if (n_shd >= n_oth) pick = W_SHADE;
        else pick = W_OTHER;
        if (pick == W_BOX) {
            // Box tests dominate (about 100 per sample against a dozen sphere tests), so
            // the vote is amortised: keep stepping boxes -- one ballot and one branch per
            // step -- until fewer than `keep` lanes are still sitting on a box record.  A low
            // threshold wins: leaving the loop costs a vote plus an exposed LDS round trip,
            // which is worth more than the lanes that idle for a few extra steps.
            const int sel = int(diag >> 8) & 7;  // tools/: A/B of the loop-exit threshold, in eighths of the starters (0 = default)
            const int eighths = sel == 0 ? (kFamily == kFeatMesh ? 3 : 2) : sel;  // mesh subset: 3/8 (C4 42.4 -> 41.1 ms)  // measured on C2 (votes now cost ~2 box steps, spheres ride along): 2/8 of the starters 30.9 ms, 3/8 31.8, 4/8 32.4
            const int frac = (n_box * eighths) >> 3;
            const int keep = frac > 8 ? frac : 8;
            // Box steps per trip around the loop's scalar checks.  Measured per kernel (tools/ab): the MIXED sphere kernel
            // (long runs of cheap box steps) 27.8 / 26.3 / 25.5 / 24.8 / 24.6 ms at 1 / 2 / 4 / 6 / 8; the reference-order
            // kernels lose with any unrolling (C2 53.2 -> 54.8 at 4; C4 62.6 -> 64.2), the full-feature kernel gains 1 % at 2;
            // the fused-slab sphere kernels (f32 mode, f64 fallback) behave like the MIXED one (f32: 20.6 -> 18.8 ms at 8).
#ifndef RTK_UNROLL_MIXED
#define RTK_UNROLL_MIXED 8
#endif
                        // the boxes-in-LDS kernels: C4 49.9 / 48.2 / 47.8 / 49.7 ms at 1 / 2 / 4 / 8 (64 spp), C5 +0.8 % at 4
            // the subset kernels, now that their primitive tests ride inside the loop: quad/box (C3) 30.5 / 29.4 / 30.7 ms at
            // 1 / 2 / 4; mesh with the program in LDS (C4 f32) 28.8 / 25.5 / 25.3
            // the lean kernel with the exact slab test (sphere scenes in the reference order): C2 50.7 / 50.0 / 52.0 ms at 1 / 2 / 4
#ifndef RTK_UNROLL_LEAN
#define RTK_UNROLL_LEAN 2
#endif
#ifndef RTK_UNROLL_QUADBOX
#define RTK_UNROLL_QUADBOX 2
#endif
#ifndef RTK_UNROLL_MESH
#define RTK_UNROLL_MESH 4
#endif
#ifndef RTK_UNROLL_SPLIT
#define RTK_UNROLL_SPLIT 4
#endif
#ifndef RTK_UNROLL_COMPACT_QUADBOX
#define RTK_UNROLL_COMPACT_QUADBOX 2
#endif
#ifndef RTK_UNROLL_COMPACT_MESH
#define RTK_UNROLL_COMPACT_MESH 4
#endif
#ifndef RTK_UNROLL_COLD
#define RTK_UNROLL_COLD 4   // the hot/cold full-feature kernel: C5 45.8 / 42.9 / 42.2 / 42.5 / 42.8 ms at 1 / 2 / 4 / 6 / 8
#endif
            constexpr int kBoxUnroll = COLD ? RTK_UNROLL_COLD : (SPLIT && RTK_UNROLL_SPLIT > 0) ? RTK_UNROLL_SPLIT
                                       : (((MIXED && !COMPACT) || FEAT == (kFeatLean | uint32_t(F_FMA_BOX))) ? RTK_UNROLL_MIXED
                                          : (kFamily == kFeatAll ? 2 : (kFamily == kFeatQuadBox ? (COMPACT ? RTK_UNROLL_COMPACT_QUADBOX : RTK_UNROLL_QUADBOX)
                                                                       : (kFamily == kFeatMesh ? (COMPACT ? RTK_UNROLL_COMPACT_MESH : RTK_UNROLL_MESH) : RTK_UNROLL_LEAN))));
            CurRec cur;  // the record at L.pc (its first 32 bytes in the MIXED / COMPACT layouts), held in registers: one LDS round trip per step
            uint32_t k = kind;
            // the record at L.pc -> cur, its kind -> k.  SPLIT: the kind comes from the LDS nibble table and only boxes are
            // fetched (from the LDS copy); a primitive's record is read from memory by the step that tests it
            // (tried on C5: requesting a sphere's record from memory already here, when a lane lands on it -- 770 vs 839 Msamples/s:
            // `cur` is one set of registers for the whole wave, so the next box step waits for that load all the same)
            auto fetch = [&]() {
                if constexpr (SPLIT) {
                    k = kind_of(L.pc);
                    if (k == OP_BOX) cur = box_at(L.pc);
                } else {
                    cur = head_at(L.pc);
                    k = cur.kind_payload & 15u;  // (a header word that IS the kind, read without the mask: the allocator answers with 14 moves per step, 21.9 vs 19.4 ms)
                }
            };
            if constexpr (SPLIT) {
                if (k == OP_BOX) cur = box_at(L.pc);
            } else {
                cur = head_at(L.pc);
            }
            const uint32_t box_kind = L.box_kind;  // a lane with an irregular ray matches nothing here: it never steps in this loop
            int remaining;
            [[maybe_unused]] SignMasks signs;
#if RTK_SIGNED_SLAB
            if constexpr (MIXED && !CH && !CHE) {  // per-lane direction signs as wave masks; rays do not change inside this loop
                signs.x = __ballot(L.inv32.x < 0.0f);
                signs.y = __ballot(L.inv32.y < 0.0f);
                signs.z = __ballot(L.inv32.z < 0.0f);
            }
#endif
#if RTK_AB_BOX_PRIO
            __builtin_amdgcn_s_setprio(RTK_AB_BOX_PRIO);
#endif
            // Sphere tests ride along: whenever `sphere_min` lanes of the wave sit on a sphere record, they are
            // stepped here, inside the box loop, instead of waiting for the loop to drain and a vote to pick
            // them (a vote round costs about two box steps).  Those lanes then return to box records, which
            // also keeps the loop populated for longer.
            do {
                if (k == box_kind) {
                    if constexpr (CH) step_box32_ch(L, cur, cnt);
                    else if constexpr (CHE) step_box32_che(L, cur, cnt);
                    else if constexpr (CHE_RT) {
                        if (che_rt) step_box32_che(L, cur, cnt);
                        else step_box32<kBoxUnits>(L, cur, cnt, signs);
                    }
                    else if constexpr (MIXED) step_box32<kBoxUnits>(L, cur, cnt, signs);
                    else step_box<false, XF, (FEAT & F_FMA_BOX) != 0>(L, cur, cnt);
                    fetch();
                    L.kind = k;
                }
#pragma unroll
                for (int extra = 1; extra < kBoxUnroll; extra++) {
                    // further box steps before the loop's scalar checks (vote / sphere / exit): the checks are a
                    // dependent v_cmp -> s_bcnt1 -> s_cmp -> branch chain per step, and the kernel is latency-bound
                    if (k == box_kind) {
                        if constexpr (CH) step_box32_ch(L, cur, cnt);
                    else if constexpr (CHE) step_box32_che(L, cur, cnt);
                    else if constexpr (CHE_RT) {
                        if (che_rt) step_box32_che(L, cur, cnt);
                        else step_box32<kBoxUnits>(L, cur, cnt, signs);
                    }
                    else if constexpr (MIXED) step_box32<kBoxUnits>(L, cur, cnt, signs);
                        else step_box<false, XF, (FEAT & F_FMA_BOX) != 0>(L, cur, cnt);
                        fetch();
                        L.kind = k;
                    }
                }
                if ([[maybe_unused]] const int n_ride = popcount64(__ballot(k == OP_SPHERE)); n_ride >= sphere_min) {
                    if (k == OP_SPHERE) {
                        if constexpr (SPLIT) cur = head_at(L.pc);
                        if constexpr (COMPACT) step_sphere_compact<XF>(L, cur, prog + L.pc, cnt, tie);
                        else if constexpr (MIXED) step_sphere_mixed<kPcUnit>(L, cur, rec_at(L.pc), cnt, tie, extent);
                        else step_sphere<XF>(L, cur, cnt, tie);
                        fetch();
                        L.kind = k;
                    }
                    RTK_PROF_MARK(2, 1, n_ride)
                }
#ifndef RTK_TRI_RIDE
#define RTK_TRI_RIDE 16
#endif
#ifndef RTK_TRI_RIDE_ALL
#define RTK_TRI_RIDE_ALL 1
#endif
#ifndef RTK_QUAD_RIDE
#define RTK_QUAD_RIDE 16
#endif
                if constexpr ((FEAT & F_TRI) != 0 && (SPLIT || RTK_TRI_RIDE_ALL) && RTK_TRI_RIDE > 0) {
                    // triangle tests ride along as well (mesh scenes: a bvh leaf is a box and one or two triangles):
                    // C4 47.8 -> 42.5 ms at 16 lanes (8: 42.8, 24: 43.6)
                    if (popcount64(__ballot(k == OP_TRI)) >= RTK_TRI_RIDE) {
                        if (k == OP_TRI) {
                            if constexpr (COLD) {
                                cold_step(std::false_type{}, k);
                                cur = head_at(L.pc);
                            } else {
                                hit_tri<XF, MIXED>(L, prog + L.pc, kTriUnits, cnt, tie);
                                fetch();
                                L.kind = k;
                            }
                        }
                    }
                }
                // ... and quad tests in the quad/box subset kernels (C3 31.9 -> 30.4 ms at 16 lanes, 30.8 at 24); the
                // full-feature kernel has no registers for it (C5 298 -> 316 ms)
                if constexpr ((FEAT & ~uint32_t(F_FMA_BOX | F_F32_BOX | F_MATTE)) == kFeatQuadBox && RTK_QUAD_RIDE > 0) {
                    if (popcount64(__ballot(k == OP_QUAD)) >= RTK_QUAD_RIDE) {
                        if (k == OP_QUAD) {
                            hit_quad<XF, MIXED>(L, prog + L.pc, kQuadUnits, cnt, tie);
                            fetch();
                            L.kind = k;
                        }
                    }
                }
                remaining = popcount64(__ballot(k == box_kind));
                RTK_PROF_MARK(1, 1, remaining)
            } while (remaining >= keep);
#if RTK_AB_BOX_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        } else if (pick == W_SPHERE) {
            // A bvh leaf usually holds two spheres in a row: same amortisation, half the starters.
            const int ssel = int(diag >> 11) & 7;  // tools/: same for the sphere loop (0 = default)
            const int sfrac = (n_sph * (ssel == 0 ? 4 : ssel)) >> 3;
            const int keep = sfrac > 8 ? sfrac : 8;
            CurRec cur = head_at(L.pc);
            uint32_t k = kind;
            int remaining;
#if RTK_AB_SPH_PRIO
            __builtin_amdgcn_s_setprio(RTK_AB_SPH_PRIO);
#endif
            do {
                if (k == OP_SPHERE) {
                    if constexpr (COMPACT) step_sphere_compact<XF>(L, cur, prog + L.pc, cnt, tie);
                    else if constexpr (MIXED) step_sphere_mixed<kPcUnit>(L, cur, rec_at(L.pc), cnt, tie, extent);
                    else step_sphere<XF>(L, cur, cnt, tie);
                    if constexpr (SPLIT) {
                        k = kind_of(L.pc);
                        if (k == OP_SPHERE) cur = head_at(L.pc);
                    } else {
                        cur = head_at(L.pc);
                        k = cur.kind_payload & 15u;
                    }
                    L.kind = k;
                }
                remaining = popcount64(__ballot(k == OP_SPHERE));
                RTK_PROF_MARK(2, 1, remaining)
            } while (remaining >= keep);
#if RTK_AB_SPH_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        } else if ((FEAT & F_QUAD) && pick == W_QUAD) {
          if constexpr ((FEAT & F_QUAD) != 0) {
            // quad::hit.  A box() is six quads in a row (quad.h:86-108): stay while at least half the starters do.
            const int keep = (n_quad >> 1) > 8 ? (n_quad >> 1) : 8;
            uint32_t k = kind;
            int remaining;
#if RTK_AB_PRIM_PRIO
            __builtin_amdgcn_s_setprio(RTK_AB_PRIM_PRIO);
#endif
            do {
                if (k == OP_QUAD) {
                    if constexpr (COLD) {
                        cold_step(std::true_type{}, k);
                    } else {
                        hit_quad<XF, MIXED>(L, prog + L.pc, kQuadUnits, cnt, tie);
                        k = kind_of(L.pc);
                        L.kind = k;
                    }
                }
                remaining = popcount64(__ballot(k == OP_QUAD));
                RTK_PROF_MARK(5, 1, remaining)
            } while (remaining >= keep);
#if RTK_AB_PRIM_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
          }
        } else if ((FEAT & F_TRI) && pick == W_TRI) {
          if constexpr ((FEAT & F_TRI) != 0) {
            // triangle::hit; a bvh leaf holds one or two triangles.
            const int keep = (n_tri >> 1) > 8 ? (n_tri >> 1) : 8;
            uint32_t k = kind;
            int remaining;
#if RTK_AB_PRIM_PRIO
            __builtin_amdgcn_s_setprio(RTK_AB_PRIM_PRIO);
#endif
            do {
                if (k == OP_TRI) {
                    if constexpr (COLD) {
                        cold_step(std::false_type{}, k);
                    } else {
                        hit_tri<XF, MIXED>(L, prog + L.pc, kTriUnits, cnt, tie);
                        k = kind_of(L.pc);
                        L.kind = k;
                    }
                }
                remaining = popcount64(__ballot(k == OP_TRI));
                RTK_PROF_MARK(5, 1, remaining)
            } while (remaining >= keep);
#if RTK_AB_PRIM_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
          }
        } else if (pick == W_SHADE) {
            // 1. the body of ray_color for the lanes whose segment ended; a finished (pixel, chunk) is written out
            // (round 3, measured and removed: lanes whose closest hit carries a noise or image texture parked without a vote until
            // 4 - 16 of them had gathered -- C5 34.1 - 34.3 vs 31.9 ms at 32 spp for every threshold: the textures are 9 % of that
            // frame (tools/knockout.py), the gathering saved none of it and the extra ballot and spills cost 7 %.)
            bool finished = false, next_sample = false, alive = false;
#if RTK_AB_SHADE_PRIO
            __builtin_amdgcn_s_setprio(RTK_AB_SHADE_PRIO);
#endif
            RTK_PROF_MARK(8, 1, 0)   // profile build: from the vote to here (marks sit outside divergent code: the profile registers are lane 0's)
            if (kind == OP_END) {
                alive = true;
                const bool ended = L.depth <= 0 || shade<real, FEAT, COUNT>(L, record_of(L.best_pc), sc, mats, cam, cnt RTK_SHADE_PROF_ARG);
                if (ended) {  // pixel_color += ray_color(...) (Camera.txt:72)
                    if constexpr (!kNoRadianceState<FEAT>) L.sum = L.sum + L.radiance;
                    L.s += 1;
                    if (L.s < s_end) {
                        next_sample = true;
                    } else {
                        finished = true;
                        alive = false;
                        L.kind = OP_DEAD;
                        RTK_PROF_CHUNK_END
                        store_partial(partial, my_slot, my_pix, L.sum);
                        if (tile_cost && cost_tile >= 0) atomicAdd(&tile_cost[cost_tile], L.segs);  // no return value: fire and forget
                    }
                }
            }
            RTK_PROF_MARK(3, 1, n_shd)   // profile build: the shade step in three parts -- ray_color's body, begin_sample, begin_segment
            // 2. lanes that just finished take the next pixels of the wave's current work item right here, so that
            // their begin_sample / begin_segment is the code the continuing lanes execute anyway; a separate refill
            // round costs as much as a shade step and serves a dozen lanes.  Only what the current item still holds:
            // fetching the next item stays with the batched refill at the top of the loop.
            // (Not in the glossy quad/box subset kernel, which is short of registers: A/B on C3 34.5 vs 35.4 ms; its matte
            // variant has room: 32.3 -> 31.9.  C2 24.4 -> 23.0.)
            constexpr bool kRefillInShade = (FEAT & ~uint32_t(F_FMA_BOX | F_F32_BOX | F_MATTE)) != kFeatQuadBox || (FEAT & F_MATTE) != 0;
            const unsigned long long m_fin = kRefillInShade ? __ballot(finished) : 0ull;
            if (uniform(m_fin != 0ull && refill_next < 64)) {
                const int avail = 64 - refill_next;
                const int rank_in_fin = int(__builtin_amdgcn_mbcnt_hi(uint32_t(m_fin >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m_fin), 0u)));
                if (finished && rank_in_fin < avail) {
                    my_pix = refill_next + rank_in_fin;
                    my_slot = it_slot;
                    px_i = it_x0 + (my_pix & 7);
                    px_j = it_y0 + (my_pix >> 3);
                    s_end = it_s_end;
                    if (it_ok && px_i < width && px_j < height) {
                        L.sum = mk(real(0), real(0), real(0));
                        L.s = it_s_begin;
                        L.segs = 0;
                        cost_tile = it_cost_tile;
                        RTK_PROF_CHUNK_BEGIN
                        alive = true;
                        next_sample = true;
                    }
                }
                const int n_fin = popcount64(m_fin);
                refill_next += n_fin < avail ? n_fin : avail;
            }
            // 3. the next sample of the same or of the new pixel, then the next segment
            if (next_sample) begin_sample(L, cam, px_i, px_j, seed_hash, cnt);
            RTK_PROF_MARK(6, 1, popcount64(__ballot(next_sample)))
            if (alive) {
                if (L.depth > 0) begin_segment<(FEAT & F_XFORM) != 0, MIXED, kCulling>(L, cnt, extent);
                else L.pc = end_pc;
                L.kind = kind_of(L.pc);
            }
#if RTK_AB_SHADE_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            RTK_PROF_MARK(7, 1, popcount64(__ballot(alive)))
        } else {
            // (A short loop here -- stay while at least half of the starters still sit on a rare record: book 2's two media are
            // neighbours in the program, an instance is entered and left through two chain switches -- was measured and
            // removed in round 3: C5 42.55 vs 42.10 ms at 32 spp, C4 15.47 vs 15.38, and the quad/box kernel lost 27 % to it,
            // C3 33.6 vs 26.3 ms at 100 spp.)
            // (Also measured and removed in round 3: the two sphere-bounded media of book 2 -- neighbours in the program, met by
            // every segment -- evaluated side by side in ONE branch-free step so that their two chains of dependent f64
            // operations interleave, B's random number drawn speculatively from the state A leaves: same image, same
            // counters, C5 42.9 vs 41.9 ms at 32 spp.)
            if (m_oth >> lane & 1ull) {
                if constexpr (MIXED && !COMPACT) step_other_mixed<kPcUnit>(L, rec_at(L.pc), cnt, tie, extent);
                else step_other<real, FEAT, BRACKET>(L, prog + L.pc, sc, cnt, tie, extent);
                L.kind = kind_of(L.pc);
            }
            RTK_PROF_MARK(4, 1, n_oth)
        }
    }
    RTK_PROF_FLUSH
    if constexpr (COUNT) {
#pragma unroll
        for (int k = 0; k < C_COUNT; k++) {
            unsigned long long v = cnt.c[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0 && v) atomicAdd(&counters[k], v);
        }
    }
}

// Known-answer entry point (tests): hittable::hit(r, interval(tmin, tmax), rec) of the uploaded scene's
// root for caller-supplied rays, one lane per ray, lock-step over the traversal program.  Exercises the same
// step_* / make_surface code as the render kernel.  out[12] = hit, t, p(3), normal(3), front_face, u, v, material.
template <typename real>
__global__ __launch_bounds__(256) void rtk_debug_hit_kernel(SceneView<real> sc, int n, const double* __restrict__ rays, const uint32_t* __restrict__ keys,
                                                             double* __restrict__ out, unsigned long long* __restrict__ draws) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= n) return;
    const double* r = rays + size_t(gid) * 9;
    Counters<true> cnt;
    cnt.clear();
    Lane<real> L;
    L.ro = mk(real(r[0]), real(r[1]), real(r[2]));
    L.rd = mk(real(r[3]), real(r[4]), real(r[5]));
    L.tm = real(r[6]);
    L.rng = pcg_hash(keys[gid * 3 + 1] + pcg_hash(keys[gid * 3 + 2] + pcg_hash(keys[gid * 3])));
    L.sv_tmin = L.sv_best_t = L.rec1_t = real(0);
    L.sv_best_pc = kNoHit;
    begin_segment<true>(L, cnt);
    L.tmin = real(r[7]);
    L.best_t = real(r[8]);
    const Slot<real>* prog = sc.program;
    auto kind_of = [&](uint32_t pc) -> uint32_t { return prog[pc].kind_payload & 15u; };
    const TieCtx<true, decltype(kind_of)> tie{sc.tie_rank_slot, kind_of};  // inert (null table) for reference-order uploads
    for (;;) {
        const Slot<real>* rec = prog + L.pc;
        const uint32_t kind = rec->kind_payload & 15u;
        if (kind == OP_END) break;
        if (kind == OP_BOX) {
            if (L.box_kind == OP_BOX) step_box<false, true>(L, *rec, cnt);
            else step_box<true, true>(L, *rec, cnt);
        } else if (kind == OP_SPHERE) {
            step_sphere<true>(L, *rec, cnt, tie);
        } else {
            step_other<real, kFeatAll>(L, rec, sc, cnt, tie);
        }
    }
    double* o = out + size_t(gid) * 12;
    for (int k = 0; k < 12; k++) o[k] = 0.0;
    o[11] = -1.0;
    if (L.best_pc != kNoHit) {
        Surface<real> sf;
        make_surface<real, kFeatAll>(prog, sc, sc.materials, L.best_pc, L.best_t, L.ro, L.rd, L.tm, sf, true);
        o[0] = 1.0;
        o[1] = double(L.best_t);
        o[2] = double(sf.p.x); o[3] = double(sf.p.y); o[4] = double(sf.p.z);
        o[5] = double(sf.normal.x); o[6] = double(sf.normal.y); o[7] = double(sf.normal.z);
        o[8] = sf.front_face ? 1.0 : 0.0;
        o[9] = double(sf.u); o[10] = double(sf.v);
        o[11] = double(sf.material);
    }
    draws[gid] = cnt.c[C_RNG];
}

// Known-answer entry points for the shading side (tests): the very device functions the render kernel executes --
// shade_surface (material::scatter / emitted, material.h:22-172), texture_value (texture.h:20-120, perlin.h:14-50) and
// begin_sample (get_ray, Camera.txt:177-200) -- on caller-supplied inputs, one lane per case.
//   scatter: mat[n]; ray[n][7] = origin, direction, time; rec[n][11] = t, p(3), normal(3), front_face, u, v, -;
//            out[n][14] = scattered?, scattered ray origin(3) direction(3), attenuation(3), time, emitted(3)
template <typename real>
__global__ __launch_bounds__(256) void rtk_debug_scatter_kernel(SceneView<real> sc, int n, const int32_t* __restrict__ mat, const double* __restrict__ ray,
                                                                 const double* __restrict__ rec, const uint32_t* __restrict__ keys, double* __restrict__ out,
                                                                 unsigned long long* __restrict__ draws) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= n) return;
    sc.n_lights = 0;  // get_lighting (Camera.txt:228) belongs to ray_color, not to scatter()
    Counters<true> cnt;
    cnt.clear();
    const double* r = ray + size_t(gid) * 7;
    const double* h = rec + size_t(gid) * 11;
    Lane<real> L;
    L.ro = mk(real(r[0]), real(r[1]), real(r[2]));
    L.rd = mk(real(r[3]), real(r[4]), real(r[5]));
    L.tm = real(r[6]);
    L.rng = pcg_hash(keys[gid * 3 + 1] + pcg_hash(keys[gid * 3 + 2] + pcg_hash(keys[gid * 3])));
    L.throughput = mk(real(1), real(1), real(1));
    L.radiance = mk(real(0), real(0), real(0));
    L.sum = mk(real(0), real(0), real(0));
    L.depth = 1 << 20;
    Surface<real> sf;
    sf.p = mk(real(h[1]), real(h[2]), real(h[3]));
    sf.normal = mk(real(h[4]), real(h[5]), real(h[6]));
    sf.front_face = h[7] != 0.0;
    sf.u = real(h[8]);
    sf.v = real(h[9]);
    sf.material = mat[gid];
#ifdef RTK_PROFILE
    ShadeProf sprof;  // (the profile build's shade functions take their sub-phase accumulators)
#endif
    const bool ended = shade_surface<real, kFeatAll>(L, sf, sc, sc.materials, cnt RTK_SHADE_PROF_ARG);
    double* o = out + size_t(gid) * 14;
    o[0] = ended ? 0.0 : 1.0;
    o[1] = double(L.ro.x); o[2] = double(L.ro.y); o[3] = double(L.ro.z);
    o[4] = double(L.rd.x); o[5] = double(L.rd.y); o[6] = double(L.rd.z);
    o[7] = double(L.throughput.x); o[8] = double(L.throughput.y); o[9] = double(L.throughput.z);
    o[10] = double(L.tm);
    o[11] = double(L.radiance.x); o[12] = double(L.radiance.y); o[13] = double(L.radiance.z);
    draws[gid] = cnt.c[C_RNG];
}
//   texture: tex[n]; uvp[n][5] = u, v, p(3); out[n][3] = texture::value(u, v, p); work[n][2] = perlin::noise calls, texel fetches
template <typename real>
__global__ __launch_bounds__(256) void rtk_debug_texture_kernel(SceneView<real> sc, int n, const int32_t* __restrict__ tex, const double* __restrict__ uvp,
                                                                 double* __restrict__ out, unsigned long long* __restrict__ work) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= n) return;
    Counters<true> cnt;
    cnt.clear();
    const double* q = uvp + size_t(gid) * 5;
    const V3<real> c = texture_value(sc, tex[gid], real(q[0]), real(q[1]), mk(real(q[2]), real(q[3]), real(q[4])), cnt);
    out[size_t(gid) * 3] = double(c.x);
    out[size_t(gid) * 3 + 1] = double(c.y);
    out[size_t(gid) * 3 + 2] = double(c.z);
    work[size_t(gid) * 2] = cnt.c[C_NOISE];
    work[size_t(gid) * 2 + 1] = cnt.c[C_TEXEL];
}
//   get_ray: ijs[n][3] = pixel i, j, sample; out[n][7] = origin, direction, time
template <typename real>
__global__ __launch_bounds__(256) void rtk_debug_get_ray_kernel(CameraRec<real> cam, uint32_t seed, int n, const int32_t* __restrict__ ijs, double* __restrict__ out,
                                                                 unsigned long long* __restrict__ draws) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= n) return;
    Counters<true> cnt;
    cnt.clear();
    Lane<real> L;
    L.s = ijs[gid * 3 + 2];
    begin_sample(L, cam, ijs[gid * 3], ijs[gid * 3 + 1], pcg_hash(seed), cnt);
    double* o = out + size_t(gid) * 7;
    o[0] = double(L.ro.x); o[1] = double(L.ro.y); o[2] = double(L.ro.z);
    o[3] = double(L.rd.x); o[4] = double(L.rd.y); o[5] = double(L.rd.z);
    o[6] = double(L.tm);
    draws[gid] = cnt.c[C_RNG];
}

// Tile order for the NEXT frame: local tiles sorted by the cost measured in this frame, most expensive first
// (64 buckets on a scale relative to the maximum: a counting sort, one workgroup).  A frame cannot end before its
// slowest sample -- a 50-bounce path inside a glass sphere is one sequential ~2-3 ms chain on one lane -- so the
// expensive tiles must START early; otherwise every GPU idles ~2.5 ms at the end of its share of the frame.
// Clears the cost array for the next measurement.
__global__ __launch_bounds__(1024) void rtk_tile_order_kernel(unsigned int* __restrict__ cost, int n, int32_t* __restrict__ order) {
    __shared__ unsigned int s_max;
    __shared__ unsigned int s_count[64], s_base[64];
    const int tid = threadIdx.x;
    if (tid == 0) s_max = 0;
    if (tid < 64) s_count[tid] = 0;
    __syncthreads();
    unsigned int local_max = 0;
    for (int k = tid; k < n; k += 1024) local_max = cost[k] > local_max ? cost[k] : local_max;
    atomicMax(&s_max, local_max);
    __syncthreads();
    const unsigned long long top = (unsigned long long)s_max + 1ull;
    for (int k = tid; k < n; k += 1024) atomicAdd(&s_count[63 - int((unsigned long long)cost[k] * 64ull / top)], 1u);  // bucket 0 = most expensive
    __syncthreads();
    if (tid == 0) {
        unsigned int run = 0;
        for (int b = 0; b < 64; b++) {
            s_base[b] = run;
            run += s_count[b];
        }
    }
    __syncthreads();
    for (int k = tid; k < n; k += 1024) {
        const int b = 63 - int((unsigned long long)cost[k] * 64ull / top);
        order[atomicAdd(&s_base[b], 1u)] = k;
    }
    __syncthreads();
    for (int k = tid; k < n; k += 1024) cost[k] = 0;
}

hipError_t launch_tile_order(unsigned int* cost, int n, int32_t* order, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    rtk_tile_order_kernel<<<dim3(1), dim3(1024), 0, stream>>>(cost, n, order);
    return hipGetLastError();
}

// Partial sums -> pixels.  For every pixel of this rank: add its chunks in index
// order, scale by 1/spp (Camera.txt:74) and write either the row-major image
// (+ gamma/clamp/quantised bytes, Camera.txt:77-89) or this rank's compact tile
// buffer.  One thread per (local tile, pixel).
// A frame with more sample chunks than the workspace has planes (kMaxPlanesPerPass) is rendered in several passes over
// consecutive chunk ranges; `acc` [local tile][3][64] carries the running sum from pass to pass.  The additions happen in the
// same order as in one pass over all chunks -- c0, + c1, + c2, ... -- so the image does not depend on the number of passes.
template <typename real>
__global__ __launch_bounds__(256) void rtk_resolve_kernel(const real* __restrict__ partial, TileMap tmap, int width, int height, real samples_scale,
                                                           real* __restrict__ out_linear, uint8_t* __restrict__ out_rgb8, real* __restrict__ acc, int first_pass,
                                                           int last_pass) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int pix = int(gid & 63);
    const long long local_tile = gid >> 6;
    if (local_tile >= tmap.n_tiles_local) return;
    const long long tile = local_tile * tmap.n_ranks + tmap.rank;
    const int i = int(tile % tmap.tiles_x) * 8 + (pix & 7), j = int(tile / tmap.tiles_x) * 8 + (pix >> 3);
    const bool inside = tile < (long long)tmap.tiles_x * tmap.tiles_y && i < width && j < height;
    V3<real> sum = mk(real(0), real(0), real(0));
    if (inside) {
        const real* src = partial + size_t(local_tile) * 192 + pix;
        const size_t chunk_stride = size_t(tmap.n_tiles_local) * 192;
        int c = 0;
        if (first_pass) {
            sum = mk(src[0], src[64], src[128]);
            c = 1;
        } else {
            const real* a = acc + size_t(local_tile) * 192 + pix;
            sum = mk(a[0], a[64], a[128]);
        }
        for (; c < tmap.n_chunks; c++) {
            const real* q = src + size_t(c) * chunk_stride;
            sum = sum + mk(q[0], q[64], q[128]);
        }
        if (!last_pass) {
            real* a = acc + size_t(local_tile) * 192 + pix;
            a[0] = sum.x;
            a[64] = sum.y;
            a[128] = sum.z;
        }
        sum = scale(samples_scale, sum);
    }
    if (!last_pass) return;
    if (tmap.compact) {
        if (out_linear) {
            real* base = out_linear + size_t(local_tile) * 192 + pix;
            base[0] = sum.x;
            base[64] = sum.y;
            base[128] = sum.z;
        }
    } else if (inside) {
        const size_t idx = (size_t(j) * width + i) * 3;
        if (out_linear) {
            out_linear[idx] = sum.x;
            out_linear[idx + 1] = sum.y;
            out_linear[idx + 2] = sum.z;
        }
        if (out_rgb8) {
            out_rgb8[idx] = to_byte(double(sum.x));
            out_rgb8[idx + 1] = to_byte(double(sum.y));
            out_rgb8[idx + 2] = to_byte(double(sum.z));
        }
    }
}

// Gathered compact tiles -> row-major image (+ bytes).  One thread per pixel slot.
template <typename real>
__global__ __launch_bounds__(256) void rtk_unpermute_kernel(const real* __restrict__ gathered, int width, int height, int tiles_x, int n_tiles, int n_ranks,
                                                             long long tiles_per_rank, real* __restrict__ out_linear, uint8_t* __restrict__ out_rgb8) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int lane = int(gid & 63);
    const long long tile = gid >> 6;
    if (tile >= n_tiles) return;
    const int i = int(tile % tiles_x) * 8 + (lane & 7), j = int(tile / tiles_x) * 8 + (lane >> 3);
    if (i >= width || j >= height) return;
    const long long rank = tile % n_ranks, local_tile = tile / n_ranks;
    const real* src = gathered + (rank * tiles_per_rank + local_tile) * 192 + lane;
    const real r = src[0], g = src[64], b = src[128];
    const size_t idx = (size_t(j) * width + i) * 3;
    if (out_linear) {
        out_linear[idx] = r;
        out_linear[idx + 1] = g;
        out_linear[idx + 2] = b;
    }
    if (out_rgb8) {
        out_rgb8[idx] = to_byte(double(r));
        out_rgb8[idx + 1] = to_byte(double(g));
        out_rgb8[idx + 2] = to_byte(double(b));
    }
}

// ------------------------------------------------------------------ launchers --

// Geometry of a persistent launch: waves per workgroup and workgroups per CU so
// that (a) the register-limited wave count per CU is reached and (b) every
// resident workgroup's LDS copy of the program fits.
template <typename Kernel>
static hipError_t plan_launch(Kernel kernel, int kMaxWavesPerBlock, size_t lds_bytes, int n_tiles, int& blocks, int& threads) {
    int device = 0, cus = 256, waves_per_cu = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    if (lds_bytes > 64 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes));
        if (e != hipSuccess) return e;
    }
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&waves_per_cu, kernel, 64, 0);  // register-limited waves per CU
    if (e != hipSuccess) return e;
    if (waves_per_cu < 4) waves_per_cu = 4;
    if (waves_per_cu > 32) waves_per_cu = 32;
    int blocks_per_cu = (waves_per_cu + kMaxWavesPerBlock - 1) / kMaxWavesPerBlock;
    if (lds_bytes > 0) {
        const int by_lds = int(kLdsBytesPerCU / lds_bytes);
        if (by_lds < 1) return hipErrorInvalidValue;
        if (blocks_per_cu > by_lds) blocks_per_cu = by_lds;
    }
    int waves_per_block = waves_per_cu / blocks_per_cu;
    if (waves_per_block > kMaxWavesPerBlock) waves_per_block = kMaxWavesPerBlock;
    if (waves_per_block < 1) waves_per_block = 1;
    threads = waves_per_block * 64;
    blocks = cus * blocks_per_cu;
    const int needed = (n_tiles + waves_per_block - 1) / waves_per_block;
    if (blocks > needed) blocks = needed;
    if (blocks < 1) blocks = 1;
    if (getenv("RTK_DEBUG"))
        fprintf(stderr, "[rtk] launch plan: occupancy API %d waves/CU, %d workgroup(s)/CU x %d waves, grid %d x %d threads, LDS %zu B/workgroup\n",
                waves_per_cu, blocks_per_cu, waves_per_block, blocks, threads, lds_bytes);
    return hipSuccess;
}

// Bytes a workgroup stages in LDS: the traversal program (MIXED or slots) followed by the material table.
// `feat` = the kernel's FEAT word: F_F32_BOX selects the MIXED program (lean family) or the COMPACT one (the others).
static bool is_compact(uint32_t feat) { return (feat & F_F32_BOX) != 0 && (feat & ~uint32_t(F_F32_BOX | F_MATTE | F_LDS_BOXES)) != kFeatLean; }
template <typename real>
static size_t lds_image_bytes(const SceneView<real>& sc, uint32_t feat) {
    const size_t program = is_compact(feat) ? size_t(sc.n_units16) * sizeof(Unit16)
                                            : ((feat & F_F32_BOX) ? size_t(sc.n_units) * sizeof(MixedHead) : size_t(sc.n_slots) * sizeof(Slot<real>));
    return program + size_t(sc.n_materials) * sizeof(MaterialRec<real>);
}

// F_LDS_BOXES kernels: the box slots, the kind nibbles (padded to 8 bytes) and the rank table.
template <typename real>
static size_t split_lds_bytes(const SceneView<real>& sc, uint32_t feat) {
    if (is_compact(feat)) return size_t(sc.n_hot_units) * sizeof(Unit16) + size_t(sc.n_materials) * sizeof(MaterialRec<real>);  // COLD: hot program + materials
    return size_t(sc.n_cached_boxes) * sizeof(BoxRec<real>) + ((size_t(sc.n_kind_words) * 4 + 7) & ~size_t(7)) + size_t(sc.n_rank_words) * 8;
}
template <typename real, uint32_t FEAT, bool COUNT, bool IN_LDS>
static hipError_t launch_one(const SceneView<real>& sc, const CameraRec<real>* cam, const TileMap& tmap, uint32_t seed, void* partial,
                             unsigned long long* counters, unsigned int* tile_counter, const int32_t* tile_order, unsigned int* tile_cost, uint32_t diag,
                             hipStream_t stream) {
    const int n_items = tmap.n_tiles_local * tmap.n_chunks;
    if (n_items <= 0) return hipSuccess;
    auto kernel = rtk_render_kernel<real, FEAT, COUNT, IN_LDS>;
    size_t lds = IN_LDS ? lds_image_bytes(sc, FEAT) : ((FEAT & F_LDS_BOXES) ? split_lds_bytes(sc, FEAT) : 0);
    TileMap tm = tmap;
    tm.mats_lds_offset = 0;
    tm.perlin_lds_offset = 0;
    if constexpr ((FEAT & F_LDS_BOXES) != 0) {  // boxes-in-LDS kernels: the material table too, if it is small and there is room
        const size_t mat_bytes = size_t(sc.n_materials) * sizeof(MaterialRec<real>);
        const size_t at = (lds + 15) & ~size_t(15);
        if (!is_compact(FEAT) /* (COLD kernels always stage it, right behind the hot program) */ && RTK_SPLIT_MATERIALS_IN_LDS && mat_bytes <= 16 * 1024 &&
            at + mat_bytes + 64 <= size_t(kLdsBytesPerCU)) {
            tm.mats_lds_offset = int32_t(at);
            lds = at + mat_bytes;
        }
#ifndef RTK_SPLIT_PERLIN_IN_LDS
#define RTK_SPLIT_PERLIN_IN_LDS 1
#endif
        // ... and the Perlin tables (book-2's noise sphere: 7.8 % of the C5 frame went into perlin::turb's dependent gathers from memory)
        const size_t perlin_bytes = size_t(sc.n_perlins) * sizeof(PerlinRec<real>);
        const size_t pat = (lds + 15) & ~size_t(15);
        if (RTK_SPLIT_PERLIN_IN_LDS && (FEAT & F_TEXTURE) != 0 && perlin_bytes > 0 && pat + perlin_bytes + 64 <= size_t(kLdsBytesPerCU)) {
            tm.perlin_lds_offset = int32_t(pat);
            lds = pat + perlin_bytes;
        }
    }
#ifndef RTK_CHAINS_IN_LDS
#define RTK_CHAINS_IN_LDS 1
#endif
    tm.chains_lds_offset = 0;
    if constexpr ((FEAT & F_XFORM) != 0 && (IN_LDS || (FEAT & F_LDS_BOXES) != 0)) {
        const size_t chain_bytes = size_t(sc.n_chains) * sizeof(ChainRec<real>);
        const size_t cat = (lds + 15) & ~size_t(15);
        if (RTK_CHAINS_IN_LDS && sc.n_chains > 1 && chain_bytes <= 8192 && cat + chain_bytes + 64 <= size_t(kLdsBytesPerCU)) {
            tm.chains_lds_offset = int32_t(cat);
            lds = cat + chain_bytes;
        }
    }
    // room for the tile order behind the program?  (never at the price of a second resident workgroup's LDS)
    tm.order_in_lds = 0;
    tm.order_lds_offset = int32_t((lds + 15) & ~size_t(15));
    const size_t order_bytes = size_t(tmap.n_tiles_local) * sizeof(int32_t);
    if (tile_order && size_t(tm.order_lds_offset) + order_bytes <= size_t(kLdsBytesPerCU) &&
        (lds == 0 || kLdsBytesPerCU / lds == kLdsBytesPerCU / (size_t(tm.order_lds_offset) + order_bytes))) {
        tm.order_in_lds = 1;
        lds = size_t(tm.order_lds_offset) + order_bytes;
    }
    int blocks = 0, threads = 0;
    hipError_t e = plan_launch(kernel, max_threads<real, FEAT, IN_LDS>() / 64, lds, n_items, blocks, threads);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(tile_counter, 0, sizeof(unsigned int), stream);
    if (e != hipSuccess) return e;
    kernel<<<dim3(blocks), dim3(threads), lds, stream>>>(sc, cam, tm, seed, static_cast<real*>(partial), counters, tile_counter, tile_order, tile_cost, diag);
    return hipGetLastError();
}

// The f32-culling-box programs are used whenever the upload built one (f64, fast order): the MIXED program of a sphere-only
// scene, the COMPACT program of any other; variant bit 20 keeps the f64 boxes of the slot program instead (A/B, tests).
template <typename real>
static bool use_mixed_program(const SceneView<real>& sc, uint32_t diag) {
    return sizeof(real) == 8 && sc.program_mixed != nullptr && (diag & (1u << 20)) == 0;
}
template <typename real>
static bool use_compact_program(const SceneView<real>& sc, uint32_t diag) {
    return sizeof(real) == 8 && sc.program_compact != nullptr && (diag & (1u << 20)) == 0;
}

// Kernel instantiation for a scene: the leanest feature subset that covers it, with or without the fused slab test;
// `mixed` / `compact` = the scene has that program and the caller did not ask for the f64 boxes.
static uint32_t kernel_features_base(uint32_t features, bool count, bool mixed, bool compact);
static uint32_t kernel_features(uint32_t features, bool count, bool mixed, bool compact) {
    const uint32_t sphere_media = count ? 0u : (features & F_SPHERE_MEDIA_ONLY);  // (the counting kernels serve every scene: no restriction)
    features &= ~uint32_t(F_SPHERE_MEDIA_ONLY);
    const uint32_t base = kernel_features_base(features, count, mixed, compact);
    return base == (kFeatAll | uint32_t(F_F32_BOX)) ? (base | sphere_media) : base;  // the COMPACT full-feature kernels (what bench.py times on C5) have the restricted variant
}
static uint32_t kernel_features_base(uint32_t features, bool count, bool mixed, bool compact) {
    const uint32_t fma = features & F_FMA_BOX, matte = features & F_MATTE, scene = features & ~uint32_t(F_FMA_BOX | F_MATTE);
    const uint32_t f32box = uint32_t(F_F32_BOX);
    // counting instantiations: the MIXED program has its own (the kernel bench.py times on sphere-only scenes, with
    // counters), the COMPACT program counts with the full-feature kernel on that program, everything else with the
    // full-feature kernel on the slot program (exact boxes: these counters equal the oracle's)
    if (count) return (mixed && scene == kFeatLean) ? (kFeatLean | f32box) : (compact ? (kFeatAll | f32box) : (kFeatAll | fma));
    if (scene == kFeatLean) return mixed ? (kFeatLean | f32box) : (kFeatLean | fma);
    if ((scene & ~kFeatQuadBox) == 0) return kFeatQuadBox | matte | (compact ? f32box : 0u);  // slot program: no registers to spare for o*inv at 4 waves/SIMD (it spills): exact slab test, A/B on C3 40.7 vs 42.2 ms
    if ((scene & ~kFeatMesh) == 0) return kFeatMesh | matte | (compact ? f32box : fma);
    return kFeatAll | (compact ? f32box : fma);
}

// Which instantiation a launch uses -- decided in ONE place, for the launcher and for rtk_kernel_name alike.
struct KernelChoice {
    uint32_t feat;   // template FEAT word, F_LDS_BOXES included
    bool count, in_lds;
};
template <typename real>
static KernelChoice choose_kernel(const SceneView<real>& sc, uint32_t features, bool count, bool allow_lds, uint32_t diag) {
    const bool mixed = use_mixed_program(sc, diag);
    bool compact = use_compact_program(sc, diag);
    bool cold = false;
    if (compact) {
        // Only when it fits one CU's LDS.  A COMPACT program split between LDS (box heads) and memory (primitives) is slower
        // than the slot program split the same way, measured: C4 with 4 494 ops 175 ms against 95 ms, C5 714 against 838
        // Msamples/s (there the coordinates run into the thousands -- a fog boundary of radius 5000 -- and the 2^-19 x extent
        // margin exceeds a quad's own box thickness: +55 % quad tests).
        const size_t bytes = size_t(sc.n_units16) * sizeof(Unit16) + size_t(sc.n_materials) * sizeof(MaterialRec<real>);
        if (!(allow_lds && bytes <= size_t(kLdsBytesPerCU)) || (diag & (1u << 23)) != 0) {  // (variant bit 23: the hot/cold form although the whole would fit)
            // ... unless its hot part does (COLD kernels: quads and triangles stay in memory, everything else in LDS) -- full-feature family, timed kernels
            const size_t hot = size_t(sc.n_hot_units) * sizeof(Unit16) + size_t(sc.n_materials) * sizeof(MaterialRec<real>);
            const uint32_t scene = features & ~uint32_t(F_FMA_BOX | F_MATTE | F_SPHERE_MEDIA_ONLY);
            cold = allow_lds && !count && sc.program_hot != nullptr && hot + 64 <= size_t(kLdsBytesPerCU) && (diag & (1u << 21)) == 0 &&
                   (scene & ~kFeatQuadBox) != 0 && (scene & ~kFeatMesh) != 0;
            compact = cold;
        }
    }
    // the matte variants pay off in f64 only (C3: f64 34.8 -> 31.9 ms, f32 26.6 -> 36.5 ms at one more wave per SIMD)
    KernelChoice k{kernel_features(sizeof(real) == 8 ? features : (features & ~uint32_t(F_MATTE)), count, mixed, compact), count, false};
    const uint32_t scene = k.feat & ~uint32_t(F_FMA_BOX | F_F32_BOX | F_MATTE | F_SPHERE_MEDIA_ONLY);
    const bool fits = allow_lds && lds_image_bytes(sc, k.feat) <= size_t(kLdsBytesPerCU);
    if (count) {  // counting builds: only the MIXED one stages its program (it is the timed kernel with counters)
        k.in_lds = fits && (k.feat & F_F32_BOX) != 0 && !is_compact(k.feat);
        return k;
    }
    k.in_lds = fits;
    if (!fits && !cold) k.feat &= ~uint32_t(F_SPHERE_MEDIA_ONLY);  // (the restricted variant exists for the LDS-resident forms only)
    if (cold) {
        k.in_lds = false;
        k.feat |= uint32_t(F_LDS_BOXES);
        return k;
    }
    // a program larger than LDS whose box records are not: the boxes-in-LDS kernel (mesh and full-feature families)
    if (!fits && !(k.feat & F_F32_BOX) && sc.box_cache != nullptr && (scene == kFeatMesh || scene == kFeatAll) && (diag & (1u << 21)) == 0) k.feat |= uint32_t(F_LDS_BOXES);
    return k;
}

template <typename real, uint32_t FEAT>
static hipError_t launch_feat(const KernelChoice& k, const SceneView<real>& sc, const CameraRec<real>* cam, const TileMap& tmap, uint32_t seed, uint32_t diag,
                              void* partial, unsigned long long* counters, unsigned int* tile_counter, const int32_t* tile_order, unsigned int* tile_cost,
                              hipStream_t stream) {
#define RTK_GO(FF, C, L) return launch_one<real, FF, C, L>(sc, cam, tmap, seed, partial, counters, tile_counter, tile_order, tile_cost, diag, stream)
    if constexpr ((FEAT & ~uint32_t(F_FMA_BOX)) == kFeatAll) {
        if (k.count) RTK_GO(FEAT, true, false);
    }
    if constexpr (FEAT == (kFeatLean | uint32_t(F_F32_BOX))) {  // the timed sphere-scene kernel with work counters: same program, same steps, same LDS staging
        if (k.count) {
            if (k.in_lds) RTK_GO(FEAT, true, true);
            RTK_GO(FEAT, true, false);
        }
    }
    if constexpr (FEAT == (kFeatAll | uint32_t(F_F32_BOX))) {  // work counters on the COMPACT program (any family's scene)
        if (k.count) RTK_GO(FEAT, true, false);
        constexpr uint32_t SMO = uint32_t(F_SPHERE_MEDIA_ONLY);  // every medium of the scene is sphere-bounded: the variant without the generic bracket
        if (k.feat & F_LDS_BOXES) {  // COLD: hot program in LDS, quads / triangles in memory
            if (k.feat & SMO) RTK_GO(FEAT | uint32_t(F_LDS_BOXES) | SMO, false, false);
            RTK_GO(FEAT | uint32_t(F_LDS_BOXES), false, false);
        }
        if (k.in_lds && (k.feat & SMO)) RTK_GO(FEAT | SMO, false, true);
    }
    if constexpr ((FEAT & F_F32_BOX) == 0 && ((FEAT & ~uint32_t(F_FMA_BOX | F_MATTE)) == kFeatMesh || (FEAT & ~uint32_t(F_FMA_BOX | F_MATTE)) == kFeatAll)) {
        if (k.feat & F_LDS_BOXES) RTK_GO(FEAT | uint32_t(F_LDS_BOXES), false, false);  // (slot programs only: see choose_kernel)
    }
    if (k.in_lds) RTK_GO(FEAT, false, true);
    RTK_GO(FEAT, false, false);
#undef RTK_GO
}

template <typename real>
hipError_t launch_render(const SceneView<real>& sc, const CameraRec<real>* cam, const TileMap& tmap, uint32_t seed, uint32_t features, bool count,
                         bool allow_lds, uint32_t diag, void* partial, unsigned long long* counters, unsigned int* tile_counter,
                         const int32_t* tile_order, unsigned int* tile_cost, hipStream_t stream) {
    const KernelChoice k = choose_kernel(sc, features, count, allow_lds, diag);
#define RTK_LAUNCH_CASE(F) \
    case F: return launch_feat<real, F>(k, sc, cam, tmap, seed, diag, partial, counters, tile_counter, tile_order, tile_cost, stream);
    switch (k.feat & ~uint32_t(F_LDS_BOXES | F_SPHERE_MEDIA_ONLY)) {
#if !defined(RTK_DEV_ONLY_ALL)   // tools/kernel_resources.py -DRTK_DEV_ONLY_ALL: only the full-feature family (quick register experiments)
        RTK_LAUNCH_CASE(kFeatLean)
        RTK_LAUNCH_CASE(kFeatQuadBox)
        RTK_LAUNCH_CASE(kFeatQuadBox | F_MATTE)
        RTK_LAUNCH_CASE(kFeatMesh)
        RTK_LAUNCH_CASE(kFeatLean | F_FMA_BOX)
        RTK_LAUNCH_CASE(kFeatMesh | F_FMA_BOX)
        RTK_LAUNCH_CASE(kFeatMesh | F_MATTE)
        RTK_LAUNCH_CASE(kFeatMesh | F_FMA_BOX | F_MATTE)
#define RTK_LAUNCH_CASE_F64(F) \
    case F:                      \
        if constexpr (sizeof(real) == 8) return launch_feat<real, F>(k, sc, cam, tmap, seed, diag, partial, counters, tile_counter, tile_order, tile_cost, stream); \
        break;
        RTK_LAUNCH_CASE_F64(kFeatLean | F_F32_BOX)
        RTK_LAUNCH_CASE_F64(kFeatQuadBox | F_F32_BOX)
        RTK_LAUNCH_CASE_F64(kFeatQuadBox | F_MATTE | F_F32_BOX)
        RTK_LAUNCH_CASE_F64(kFeatMesh | F_F32_BOX)
        RTK_LAUNCH_CASE_F64(kFeatMesh | F_MATTE | F_F32_BOX)
#endif
        RTK_LAUNCH_CASE(kFeatAll)
        RTK_LAUNCH_CASE(kFeatAll | F_FMA_BOX)
#ifndef RTK_LAUNCH_CASE_F64
#define RTK_LAUNCH_CASE_F64(F) \
    case F:                      \
        if constexpr (sizeof(real) == 8) return launch_feat<real, F>(k, sc, cam, tmap, seed, diag, partial, counters, tile_counter, tile_order, tile_cost, stream); \
        break;
#endif
        RTK_LAUNCH_CASE_F64(kFeatAll | F_F32_BOX)
#undef RTK_LAUNCH_CASE_F64
    }
#undef RTK_LAUNCH_CASE
    return hipErrorInvalidValue;
}
template hipError_t launch_render<double>(const SceneView<double>&, const CameraRec<double>*, const TileMap&, uint32_t, uint32_t, bool, bool, uint32_t, void*,
                                          unsigned long long*, unsigned int*, const int32_t*, unsigned int*, hipStream_t);
template hipError_t launch_render<float>(const SceneView<float>&, const CameraRec<float>*, const TileMap&, uint32_t, uint32_t, bool, bool, uint32_t, void*,
                                         unsigned long long*, unsigned int*, const int32_t*, unsigned int*, hipStream_t);

template <typename real>
const char* render_kernel_name(const SceneView<real>& sc, uint32_t features, bool count, bool allow_lds, uint32_t diag) {
    static thread_local char name[96];
    const KernelChoice k = choose_kernel(sc, features, count, allow_lds, diag);
    snprintf(name, sizeof name, "rtk_render_kernel<%s, %uu, %s, %s>", sizeof(real) == 8 ? "double" : "float", k.feat, k.count ? "true" : "false", k.in_lds ? "true" : "false");
    return name;
}
template const char* render_kernel_name<double>(const SceneView<double>&, uint32_t, bool, bool, uint32_t);
template const char* render_kernel_name<float>(const SceneView<float>&, uint32_t, bool, bool, uint32_t);

template <typename real>
hipError_t launch_resolve(const void* partial, const TileMap& tmap, int width, int height, double samples_scale, void* out_linear, uint8_t* out_rgb8,
                          void* acc, bool first_pass, bool last_pass, hipStream_t stream) {
    const long long slots = (long long)tmap.n_tiles_local * 64;
    if (slots <= 0) return hipSuccess;
    rtk_resolve_kernel<real><<<dim3(int((slots + 255) / 256)), dim3(256), 0, stream>>>(static_cast<const real*>(partial), tmap, width, height,
                                                                                      real(samples_scale), static_cast<real*>(out_linear), out_rgb8,
                                                                                      static_cast<real*>(acc), first_pass ? 1 : 0, last_pass ? 1 : 0);
    return hipGetLastError();
}
template hipError_t launch_resolve<double>(const void*, const TileMap&, int, int, double, void*, uint8_t*, void*, bool, bool, hipStream_t);
template hipError_t launch_resolve<float>(const void*, const TileMap&, int, int, double, void*, uint8_t*, void*, bool, bool, hipStream_t);

template <typename real>
hipError_t launch_debug_hit(const SceneView<real>& sc, int n, const double* d_rays, const uint32_t* d_keys, double* d_out, unsigned long long* d_draws,
                            hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    rtk_debug_hit_kernel<real><<<dim3((n + 255) / 256), dim3(256), 0, stream>>>(sc, n, d_rays, d_keys, d_out, d_draws);
    return hipGetLastError();
}
template hipError_t launch_debug_hit<double>(const SceneView<double>&, int, const double*, const uint32_t*, double*, unsigned long long*, hipStream_t);
template hipError_t launch_debug_hit<float>(const SceneView<float>&, int, const double*, const uint32_t*, double*, unsigned long long*, hipStream_t);

template <typename real>
hipError_t launch_debug_scatter(const SceneView<real>& sc, int n, const int32_t* d_mat, const double* d_ray, const double* d_rec, const uint32_t* d_keys, double* d_out,
                                unsigned long long* d_draws, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    rtk_debug_scatter_kernel<real><<<dim3((n + 255) / 256), dim3(256), 0, stream>>>(sc, n, d_mat, d_ray, d_rec, d_keys, d_out, d_draws);
    return hipGetLastError();
}
template hipError_t launch_debug_scatter<double>(const SceneView<double>&, int, const int32_t*, const double*, const double*, const uint32_t*, double*, unsigned long long*, hipStream_t);
template hipError_t launch_debug_scatter<float>(const SceneView<float>&, int, const int32_t*, const double*, const double*, const uint32_t*, double*, unsigned long long*, hipStream_t);

template <typename real>
hipError_t launch_debug_texture(const SceneView<real>& sc, int n, const int32_t* d_tex, const double* d_uvp, double* d_out, unsigned long long* d_work, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    rtk_debug_texture_kernel<real><<<dim3((n + 255) / 256), dim3(256), 0, stream>>>(sc, n, d_tex, d_uvp, d_out, d_work);
    return hipGetLastError();
}
template hipError_t launch_debug_texture<double>(const SceneView<double>&, int, const int32_t*, const double*, double*, unsigned long long*, hipStream_t);
template hipError_t launch_debug_texture<float>(const SceneView<float>&, int, const int32_t*, const double*, double*, unsigned long long*, hipStream_t);

template <typename real>
hipError_t launch_debug_get_ray(const CameraRec<real>& cam, uint32_t seed, int n, const int32_t* d_ijs, double* d_out, unsigned long long* d_draws, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    rtk_debug_get_ray_kernel<real><<<dim3((n + 255) / 256), dim3(256), 0, stream>>>(cam, seed, n, d_ijs, d_out, d_draws);
    return hipGetLastError();
}
template hipError_t launch_debug_get_ray<double>(const CameraRec<double>&, uint32_t, int, const int32_t*, double*, unsigned long long*, hipStream_t);
template hipError_t launch_debug_get_ray<float>(const CameraRec<float>&, uint32_t, int, const int32_t*, double*, unsigned long long*, hipStream_t);

template <typename real>
hipError_t launch_unpermute(const void* gathered, int width, int height, int n_ranks, long long tiles_per_rank, void* out_linear, uint8_t* out_rgb8,
                            hipStream_t stream) {
    const int tiles_x = (width + 7) / 8, tiles_y = (height + 7) / 8;
    const long long slots = (long long)tiles_x * tiles_y * 64;
    const int blocks = int((slots + 255) / 256);
    rtk_unpermute_kernel<real><<<dim3(blocks), dim3(256), 0, stream>>>(static_cast<const real*>(gathered), width, height, tiles_x, tiles_x * tiles_y, n_ranks,
                                                                       tiles_per_rank, static_cast<real*>(out_linear), out_rgb8);
    return hipGetLastError();
}
template hipError_t launch_unpermute<double>(const void*, int, int, int, long long, void*, uint8_t*, hipStream_t);
template hipError_t launch_unpermute<float>(const void*, int, int, int, long long, void*, uint8_t*, hipStream_t);

#ifdef RTK_ISA_PROBES
// Instruction-cost probes (tools/isa_costs.py; never part of the product build): one kernel per unit of work of the lean
// MIXED kernel family (what bench.py times on C2), each loading a lane's state from memory, running ONE step through the
// product's own step function and storing the state back.  The tool counts the VALU instructions of each probe by class in the
// ISA and subtracts the empty probe: what is left is the instruction cost of that unit of work when a whole wave executes it
// once -- the per-unit figures behind bench.py's roofline.work_frac (work counters x these costs / 64 lanes / SIMD-cycles
// available: a fraction that executing MORE instructions cannot raise).
constexpr uint32_t kProbeFeat = kFeatLean | uint32_t(F_F32_BOX);
enum ProbeKind : int { PROBE_EMPTY, PROBE_BOX, PROBE_SPHERE, PROBE_SEGMENT, PROBE_SAMPLE, PROBE_MISS, PROBE_LAMBERTIAN, PROBE_METAL, PROBE_DIELECTRIC, PROBE_PARTIAL };
template <int KIND>
__global__ __launch_bounds__(1024) void rtk_isa_probe(Lane<double>* __restrict__ lanes, const MixedHead* __restrict__ prog, SceneView<double> sc,
                                                       const CameraRec<double>* __restrict__ cam_ptr, double* __restrict__ partial, float extent) {
    const int gid = blockIdx.x * 1024 + threadIdx.x;
    Lane<double> L = lanes[gid];
    Counters<false> cnt;
    const NoTie tie{};
    const MixedHead* rec = reinterpret_cast<const MixedHead*>(reinterpret_cast<const unsigned char*>(prog) + L.pc);
    if constexpr (KIND == PROBE_BOX) {  // one box step: the slab test on the record in registers, then the next record and its kind
        MixedHead cur = *rec;
        step_box32_ch(L, cur, cnt);
        cur = *reinterpret_cast<const MixedHead*>(reinterpret_cast<const unsigned char*>(prog) + L.pc);
        L.kind = cur.kind_payload & 15u;
        L.oi32_lo.x = cur.f(0);  // (keeps the fetched record alive, as the next step's slab test does)
    } else if constexpr (KIND == PROBE_SPHERE) {
        MixedHead cur = *rec;
        step_sphere_mixed<kChPcUnit>(L, cur, rec, cnt, tie, extent);
        cur = *reinterpret_cast<const MixedHead*>(reinterpret_cast<const unsigned char*>(prog) + L.pc);
        L.kind = cur.kind_payload & 15u;
        L.oi32_lo.x = cur.f(0);
    } else if constexpr (KIND == PROBE_SEGMENT) {
        begin_segment<false, true, 1>(L, cnt, extent);
        L.kind = rec->kind_payload & 15u;
    } else if constexpr (KIND == PROBE_SAMPLE) {
        begin_sample(L, *cam_ptr, int(L.segs), L.depth, pcg_hash(L.best_pc), cnt);
    } else if constexpr (KIND == PROBE_MISS) {
        L.sum = L.sum + L.throughput * ld3(cam_ptr->background);
    } else if constexpr (KIND == PROBE_LAMBERTIAN || KIND == PROBE_METAL || KIND == PROBE_DIELECTRIC) {
        Surface<double> sf;
        make_surface_mixed(rec, 0u, L.best_t, L.ro, L.rd, L.tm, sf);
        constexpr int kMat = KIND == PROBE_LAMBERTIAN ? int(RTK_MAT_LAMBERTIAN) : (KIND == PROBE_METAL ? int(RTK_MAT_METAL) : int(RTK_MAT_DIELECTRIC));
#ifdef RTK_PROFILE
        ShadeProf sprof;
#endif
        const bool ended = shade_surface<double, kProbeFeat, kMat>(L, sf, sc, sc.materials, cnt RTK_SHADE_PROF_ARG);
        L.kind = ended ? 1u : 0u;
    } else if constexpr (KIND == PROBE_PARTIAL) {
        store_partial(partial, int(L.segs), L.depth, L.sum);
    }
    lanes[gid] = L;
}
template __global__ void rtk_isa_probe<PROBE_EMPTY>(Lane<double>*, const MixedHead*, SceneView<double>, const CameraRec<double>*, double*, float);
template __global__ void rtk_isa_probe<PROBE_BOX>(Lane<double>*, const MixedHead*, SceneView<double>, const CameraRec<double>*, double*, float);
template __global__ void rtk_isa_probe<PROBE_SPHERE>(Lane<double>*, const MixedHead*, SceneView<double>, const CameraRec<double>*, double*, float);
template __global__ void rtk_isa_probe<PROBE_SEGMENT>(Lane<double>*, const MixedHead*, SceneView<double>, const CameraRec<double>*, double*, float);
template __global__ void rtk_isa_probe<PROBE_SAMPLE>(Lane<double>*, const MixedHead*, SceneView<double>, const CameraRec<double>*, double*, float);
template __global__ void rtk_isa_probe<PROBE_MISS>(Lane<double>*, const MixedHead*, SceneView<double>, const CameraRec<double>*, double*, float);
template __global__ void rtk_isa_probe<PROBE_LAMBERTIAN>(Lane<double>*, const MixedHead*, SceneView<double>, const CameraRec<double>*, double*, float);
template __global__ void rtk_isa_probe<PROBE_METAL>(Lane<double>*, const MixedHead*, SceneView<double>, const CameraRec<double>*, double*, float);
template __global__ void rtk_isa_probe<PROBE_DIELECTRIC>(Lane<double>*, const MixedHead*, SceneView<double>, const CameraRec<double>*, double*, float);
template __global__ void rtk_isa_probe<PROBE_PARTIAL>(Lane<double>*, const MixedHead*, SceneView<double>, const CameraRec<double>*, double*, float);
#endif  // RTK_ISA_PROBES

}  // namespace rtk
