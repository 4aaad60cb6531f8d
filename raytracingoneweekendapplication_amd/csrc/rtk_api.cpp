// rtk_api.cpp -- implementation of the C ABI in include/rtk.h (host side of
// librtk_hip.so; compiled by hipcc).  It validates an rtk_scene_desc, compiles
// the scene graph into the linear traversal program of rtk_device_layout.h,
// uploads f64 and f32 images of it to HBM and enqueues the kernels of
// rtk_trace.hip.  There is no host rendering path: without a gfx950 device every
// compute entry point fails with RTK_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "rtk.h"
#include "rtk_device_layout.h"
#include "rtk_internal.h"
#include "rtk_trace.h"

namespace rtk {

thread_local std::string g_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

}  // namespace rtk

namespace {

using rtk::fail;
using rtk::g_error;

#define RTK_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) return fail(RTK_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

using namespace rtk;

// ---------------------------------------------------------------------------
// Scene graph -> traversal program
// ---------------------------------------------------------------------------
struct ChainStep {
    bool rotate;
    int32_t index;  // into translates[] / rotates[]
    bool operator<(const ChainStep& o) const { return rotate != o.rotate ? rotate < o.rotate : index < o.index; }
};
using Chain = std::vector<ChainStep>;

struct Program {
    bool has_media_bracket = false;  // a constant_medium whose boundary is not one stationary sphere: OP_MED_BEGIN / MID / END records
    std::vector<Op> ops;
    std::vector<uint32_t> ranks;  // parallel to ops: rtk_node.c of a primitive op (1 + reference visiting rank; 0 = none)
    std::vector<uint32_t> extra;  // parallel to ops: OP_MED_SPHERE -> index of the boundary sphere
    std::vector<Chain> chains;
    std::map<Chain, uint32_t> chain_ids;
    size_t last_label = size_t(-1);  // op index some skip link points at
    uint32_t features = 0;
    int n_primitive_ops = 0;
    std::string error;
    int error_code = RTK_OK;
};

constexpr size_t kMaxOps = size_t(1) << 26;
constexpr int kMaxGraphDepth = 4096;

struct Compiler {
    const rtk_scene_desc& sc;
    Program& prog;

    bool bad(int code, const std::string& why) {
        if (prog.error_code == RTK_OK) {
            prog.error_code = code;
            prog.error = why;
        }
        return false;
    }
    uint32_t chain_id(const Chain& c) {
        auto it = prog.chain_ids.find(c);
        if (it != prog.chain_ids.end()) return it->second;
        uint32_t id = uint32_t(prog.chains.size());
        if (id > 255u) bad(RTK_ERR_UNSUPPORTED, "more than 255 distinct instance-transform chains");  // 8 bits of a record's aux word
        prog.chains.push_back(c);
        prog.chain_ids[c] = id;
        return id;
    }
    uint32_t push(uint32_t kind, uint32_t payload, uint32_t aux, uint32_t rank = 0) {
        prog.ops.push_back(Op{make_op(kind, payload), aux});
        prog.ranks.push_back(rank);
        prog.extra.push_back(0);
        return uint32_t(prog.ops.size()) - 1;
    }
    void label_here() { prog.last_label = prog.ops.size(); }
    // Consecutive chain switches collapse into one op unless a skip link lands
    // between them.
    void push_chain(uint32_t chain, uint32_t entries) {
        if (!prog.ops.empty() && (prog.ops.back().kind_payload & 15u) == OP_CHAIN && prog.last_label != prog.ops.size()) {
            Op& last = prog.ops.back();
            last.aux += entries;
            last.kind_payload = make_op(OP_CHAIN, chain);
            return;
        }
        push(OP_CHAIN, chain, entries);
    }

    bool emit(int32_t node, const Chain& chain, bool in_medium, int depth) {
        if (prog.error_code != RTK_OK) return false;
        if (node < 0 || node >= sc.n_nodes) return bad(RTK_ERR_INVALID, "node index out of range");
        if (depth > kMaxGraphDepth) return bad(RTK_ERR_INVALID, "scene graph too deep (cycle?)");
        if (prog.ops.size() > kMaxOps) return bad(RTK_ERR_UNSUPPORTED, "traversal program too large");
        const rtk_node& n = sc.nodes[node];
        const uint32_t cid = chain_id(chain);
        switch (n.kind) {
            case RTK_NODE_SPHERE:
                if (n.a < 0 || n.a >= sc.n_spheres) return bad(RTK_ERR_INVALID, "sphere index out of range");
                push(OP_SPHERE, uint32_t(n.a), cid, n.c > 0 ? uint32_t(n.c) : 0u);
                prog.n_primitive_ops++;
                return true;
            case RTK_NODE_QUAD:
                if (n.a < 0 || n.a >= sc.n_quads) return bad(RTK_ERR_INVALID, "quad index out of range");
                push(OP_QUAD, uint32_t(n.a), cid, n.c > 0 ? uint32_t(n.c) : 0u);
                prog.features |= F_QUAD;
                prog.n_primitive_ops++;
                return true;
            case RTK_NODE_TRIANGLE:
                if (n.a < 0 || n.a >= sc.n_triangles) return bad(RTK_ERR_INVALID, "triangle index out of range");
                push(OP_TRI, uint32_t(n.a), cid, n.c > 0 ? uint32_t(n.c) : 0u);
                prog.features |= F_TRI;
                prog.n_primitive_ops++;
                return true;
            case RTK_NODE_LIST:
                if (n.a < 0 || n.b < 0 || int64_t(n.a) + n.b > sc.n_list_children) return bad(RTK_ERR_INVALID, "list children out of range");
                for (int32_t k = 0; k < n.b; k++)
                    if (!emit(sc.list_children[n.a + k], chain, in_medium, depth + 1)) return false;
                return true;
            case RTK_NODE_BVH: {
                if (n.c < 0 || n.c >= sc.n_bvh_boxes) return bad(RTK_ERR_INVALID, "bvh box index out of range");
                uint32_t at = push(OP_BOX, uint32_t(n.c), 0);
                if (!emit(n.a, chain, in_medium, depth + 1)) return false;
                if (!emit(n.b, chain, in_medium, depth + 1)) return false;
                prog.ops[at].aux = uint32_t(prog.ops.size());
                label_here();
                return true;
            }
            case RTK_NODE_TRANSLATE:
            case RTK_NODE_ROTATE_Y: {
                const bool rot = n.kind == RTK_NODE_ROTATE_Y;
                if (n.a < 0 || n.a >= (rot ? sc.n_rotates : sc.n_translates)) return bad(RTK_ERR_INVALID, "transform index out of range");
                if (int(chain.size()) >= kMaxChain) return bad(RTK_ERR_UNSUPPORTED, "more than 4 nested instance transforms");
                Chain inner = chain;
                inner.push_back(ChainStep{rot, n.a});
                prog.features |= F_XFORM;
                push_chain(chain_id(inner), 1);
                if (!emit(n.b, inner, in_medium, depth + 1)) return false;
                push_chain(cid, 0);
                return true;
            }
            case RTK_NODE_MEDIUM: {
                if (n.a < 0 || n.a >= sc.n_media) return bad(RTK_ERR_INVALID, "medium index out of range");
                if (in_medium) return bad(RTK_ERR_UNSUPPORTED, "constant_medium inside the boundary of another constant_medium");
                prog.features |= F_MEDIA;
                if (n.b >= 0 && n.b < sc.n_nodes && sc.nodes[n.b].kind == RTK_NODE_SPHERE && sc.nodes[n.b].a >= 0 && sc.nodes[n.b].a < sc.n_spheres && !getenv("RTK_NO_MED_SPHERE")) {
                    const rtk_sphere& s = sc.spheres[sc.nodes[n.b].a];
                    if (s.center_dir.x == 0 && s.center_dir.y == 0 && s.center_dir.z == 0) {  // a stationary sphere as the boundary: the fused record
                        const uint32_t at = push(OP_MED_SPHERE, uint32_t(n.a), cid);
                        prog.extra[at] = uint32_t(sc.nodes[n.b].a);
                        prog.n_primitive_ops++;
                        return true;
                    }
                }
                prog.has_media_bracket = true;
                push(OP_MED_BEGIN, uint32_t(n.a), 0);
                if (!emit(n.b, chain, true, depth + 1)) return false;
                uint32_t mid = push(OP_MED_MID, uint32_t(n.a), 0);
                if (!emit(n.b, chain, true, depth + 1)) return false;
                push(OP_MED_END, uint32_t(n.a), cid);
                prog.ops[mid].aux = uint32_t(prog.ops.size());
                label_here();
                prog.n_primitive_ops++;
                return true;
            }
        }
        return bad(RTK_ERR_INVALID, "unknown node kind");
    }
};

bool texture_needs_uv(const rtk_scene_desc& sc, int32_t tex, int depth = 0) {
    if (tex < 0 || tex >= sc.n_textures || depth > 64) return false;
    const rtk_texture& t = sc.textures[tex];
    if (t.kind == RTK_TEX_IMAGE || t.kind == RTK_TEX_CHECKER_TRI) return true;
    if (t.kind == RTK_TEX_CHECKER) return texture_needs_uv(sc, t.even, depth + 1) || texture_needs_uv(sc, t.odd, depth + 1);
    return false;
}

int validate_tables(const rtk_scene_desc& sc) {
    if (sc.abi_version != RTK_ABI_VERSION) return fail(RTK_ERR_INVALID, "scene abi_version %d != %d", sc.abi_version, RTK_ABI_VERSION);
    if (sc.n_nodes <= 0 || !sc.nodes) return fail(RTK_ERR_INVALID, "scene has no nodes (the reference recurses forever on an empty world, bvh.h:38-43)");
    if (sc.n_materials >= (1 << 24)) return fail(RTK_ERR_UNSUPPORTED, "more than 2^24 materials");
    for (int32_t i = 0; i < sc.n_textures; i++) {
        const rtk_texture& t = sc.textures[i];
        switch (t.kind) {
            case RTK_TEX_SOLID: break;
            case RTK_TEX_CHECKER:
            case RTK_TEX_CHECKER_TRI:
                // children are emitted before their parent by every flattener, which also rules out cycles
                if (t.even < 0 || t.even >= i || t.odd < 0 || t.odd >= i) return fail(RTK_ERR_INVALID, "texture %d: child index must precede it", i);
                break;
            case RTK_TEX_IMAGE: {
                if (t.image < 0 || t.image >= sc.n_images) return fail(RTK_ERR_INVALID, "texture %d: image index out of range", i);
                const rtk_image& im = sc.images[t.image];
                if (im.width > 0 && im.height > 0 &&
                    (im.texel_offset < 0 || im.texel_offset + int64_t(im.width) * im.height * 3 > sc.n_texel_bytes))
                    return fail(RTK_ERR_INVALID, "image %d: texels out of range", t.image);
                break;
            }
            case RTK_TEX_NOISE:
                if (t.image < 0 || t.image >= sc.n_perlins) return fail(RTK_ERR_INVALID, "texture %d: perlin index out of range", i);
                break;
            default: return fail(RTK_ERR_INVALID, "texture %d: unknown kind %d", i, t.kind);
        }
    }
    for (int32_t i = 0; i < sc.n_materials; i++) {
        const rtk_material& m = sc.materials[i];
        const bool textured = m.kind == RTK_MAT_LAMBERTIAN || m.kind == RTK_MAT_DIFFUSE_LIGHT || m.kind == RTK_MAT_ISOTROPIC;
        if (m.kind < RTK_MAT_LAMBERTIAN || m.kind > RTK_MAT_SPECULAR) return fail(RTK_ERR_INVALID, "material %d: unknown kind %d", i, m.kind);
        if (textured && (m.texture < 0 || m.texture >= sc.n_textures)) return fail(RTK_ERR_INVALID, "material %d: texture index out of range", i);
    }
    auto mat_ok = [&](int32_t m) { return m >= 0 && m < sc.n_materials; };
    for (int32_t i = 0; i < sc.n_spheres; i++)
        if (!mat_ok(sc.spheres[i].material)) return fail(RTK_ERR_INVALID, "sphere %d: material out of range", i);
    for (int32_t i = 0; i < sc.n_quads; i++)
        if (!mat_ok(sc.quads[i].material)) return fail(RTK_ERR_INVALID, "quad %d: material out of range", i);
    for (int32_t i = 0; i < sc.n_triangles; i++)
        if (!mat_ok(sc.triangles[i].material)) return fail(RTK_ERR_INVALID, "triangle %d: material out of range", i);
    for (int32_t i = 0; i < sc.n_media; i++)
        if (!mat_ok(sc.media[i].material)) return fail(RTK_ERR_INVALID, "medium %d: material out of range", i);
    return RTK_OK;
}

// ---------------------------------------------------------------------------
// Device image of a scene for one arithmetic type
// ---------------------------------------------------------------------------
template <typename real>
struct DeviceScene {
    std::vector<void*> allocations;
    SceneView<real> view{};
    int64_t bytes = 0;

    void release() {
        for (void* p : allocations) (void)hipFree(p);
        allocations.clear();
        view = SceneView<real>{};
        bytes = 0;
    }
    template <typename T>
    int upload(const std::vector<T>& host, const T** out) {
        *out = nullptr;
        if (host.empty()) return RTK_OK;
        void* d = nullptr;
        RTK_HIP(hipMalloc(&d, host.size() * sizeof(T)));
        allocations.push_back(d);
        RTK_HIP(hipMemcpy(d, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
        bytes += int64_t(host.size() * sizeof(T));
        *out = static_cast<const T*>(d);
        return RTK_OK;
    }
};

template <typename real>
void store3(real* dst, const rtk_vec3& v) {
    dst[0] = real(v.x);
    dst[1] = real(v.y);
    dst[2] = real(v.z);
}

// Outward rounding of a box bound to float.
inline float round_down(double x) {
    float f = float(x);
    return double(f) > x ? std::nextafterf(f, -INFINITY) : f;
}
inline float round_up(double x) {
    float f = float(x);
    return double(f) < x ? std::nextafterf(f, INFINITY) : f;
}

// How far an f32 culling box is grown beyond its outward-rounded bounds.  RTK_SIGNED_SLAB: the ray carries the error of
// its own origin term (a bracket of o/d, rtk_trace.hip begin_culling32), so a box answers only for the product b * rcp(d)
// -- float(d) 2^-24, v_rcp_f32 one ulp, the fused multiply-add's final rounding 2^-24 of it: < 2^-22 relative -- i.e. a
// plane displaced by < 2^-22 |b|; grown by 2^-21 of the box's largest coordinate (twice that).  Otherwise: the scene-wide
// margin the caller computed (2^-19 x extent).
static double box_margin(const rtk_aabb& b, double scene_margin) {
#if RTK_SIGNED_SLAB
    (void)scene_margin;
    double big = 0.0;
    for (double v : {b.xmin, b.xmax, b.ymin, b.ymax, b.zmin, b.zmax}) big = std::max(big, std::fabs(v));
    return std::ldexp(big, -21) + 1e-300;
#else
    return scene_margin;
#endif
}

// The MIXED program of rtk_device_layout.h for a sphere-only scene whose boxes carry rtk_scene_optimize's margin:
// f32 culling boxes (rounded outward, then grown by 2^-19 of the largest coordinate in play), f64 spheres.
// Error budget of the f32 slab test t = fma(b, inv32, -(o32 * inv32)) with inv32 = rcp(float(d)) (1 ulp), o32 =
// float(o): |b*inv| 2^-22.4 + |o*inv| 2^-22 + |t| 2^-24, i.e. a plane displaced by less than 2^-21 (|b| + |o|)
// <= 2^-20 x extent -- half the margin.  The kernel sends rays whose origin leaves [-extent, extent]^3 through the
// exact test instead (they cannot come from a surface of the scene).
static void build_mixed_program(const rtk_scene_desc& sc, const Program& prog, double eye_extent, std::vector<MixedHead>& units, std::vector<uint32_t>& rank_of_unit,
                                float& extent_out) {
    auto kind_of = [&](const Op& op) -> uint32_t {
        uint32_t kind = op.kind_payload & 15u;
        if (kind == OP_SPHERE) {
            const rtk_sphere& s = sc.spheres[op.kind_payload >> 4];
            if (s.center_dir.x != 0 || s.center_dir.y != 0 || s.center_dir.z != 0) kind = OP_SPHERE_MOVING;
        }
        return kind;
    };
    std::vector<uint32_t> unit_of_op(prog.ops.size() + 1, 0);
    double extent = eye_extent;
    for (size_t i = 0; i < prog.ops.size(); i++) {
        const Op& op = prog.ops[i];
        unit_of_op[i + 1] = unit_of_op[i] + uint32_t(mixed_units(kind_of(op)));
        if ((op.kind_payload & 15u) == OP_BOX) {
            const rtk_aabb& b = sc.bvh_boxes[op.kind_payload >> 4];
            for (double v : {b.xmin, b.xmax, b.ymin, b.ymax, b.zmin, b.zmax}) extent = std::max(extent, std::fabs(v));
        } else if ((op.kind_payload & 15u) == OP_SPHERE) {
            const rtk_sphere& s = sc.spheres[op.kind_payload >> 4];
            for (double c : {s.center0.x, s.center0.y, s.center0.z, s.center0.x + s.center_dir.x, s.center0.y + s.center_dir.y, s.center0.z + s.center_dir.z})
                extent = std::max(extent, std::fabs(c) + s.radius);
        }
    }
    extent = double(round_up(extent * 1.0000001));
    const double margin = std::ldexp(extent, -19);  // (A/B on C2: a margin 128 times smaller renders 0.7 % faster -- nothing to gain)
    (void)margin;
    units.assign(unit_of_op.back(), MixedHead{});
    std::memset(units.data(), 0, units.size() * sizeof(MixedHead));
    rank_of_unit.assign(unit_of_op.back(), 0u);
    for (size_t i = 0; i < prog.ops.size(); i++) {
        const Op& op = prog.ops[i];
        const uint32_t kind = kind_of(op), payload = op.kind_payload >> 4;
        MixedHead* rec = &units[unit_of_op[i]];
        rec->kind_payload = make_op(kind, payload);
        rec->aux = op.aux;
        rank_of_unit[unit_of_op[i]] = prog.ranks[i];
        if (kind == OP_BOX) {
            const rtk_aabb& b = sc.bvh_boxes[payload];
#if RTK_CH_BOX
            // Centre / half-extent form.  The kernel computes, per axis, tc = fma(c, inv, -oi), near = fma(-h, |inv|, tc),
            // far = fma(h, |inv|, tc) with inv = v_rcp_f32(float(d)) (relative error <= 2^-24 + 2^-23) and oi = float(o) * inv
            // (<= 2^-24 + that + 2^-24).  Against the exact planes (c -/+ h - o) / d of the stored box the computed ones are off
            // by at most [(1.5 + 1) 2^-23 (|c| + h) + (2.5 + 1) 2^-23 |o|] / |d| -- the products' errors plus one final rounding
            // each of tc and of near / far, both bounded by (|c| + |o| + h) / |d| -- i.e. a plane displaced by less than
            // 2^-21 (|c| + h) + 2^-20 |o|.  |o| <= extent on every axis (a ray that starts outside never enters the float loop,
            // rtk_trace.hip begin_culling32), so the half-extent is grown by 2^-21 (|c| + h) + 2^-20 extent, plus the distance
            // the centre moved when it was rounded to float, and rounded up.
            const double lo[3] = {b.xmin, b.ymin, b.zmin}, hi[3] = {b.xmax, b.ymax, b.zmax};
            for (int ax = 0; ax < 3; ax++) {
                const double c = 0.5 * (lo[ax] + hi[ax]);
                const float cf = float(c);
                const double h = std::max(hi[ax] - double(cf), double(cf) - lo[ax]);  // covers [lo, hi] from the float centre
                const double grown = h + std::ldexp(std::fabs(double(cf)) + h, -21) + std::ldexp(extent, -20) + 1e-300;
                rec->set_f(ax, cf);
                rec->set_f(3 + ax, round_up(grown * (1.0 + 1e-7)));
            }
#else
            const double m = box_margin(b, margin);
            rec->set_f(0, round_down(b.xmin - m)); rec->set_f(1, round_up(b.xmax + m));
            rec->set_f(2, round_down(b.ymin - m)); rec->set_f(3, round_up(b.ymax + m));
            rec->set_f(4, round_down(b.zmin - m)); rec->set_f(5, round_up(b.zmax + m));
#endif
            rec->aux = unit_of_op[op.aux] * ((RTK_CH_BOX && RTK_CH_BYTE_PC) ? uint32_t(sizeof(MixedHead)) : 1u);  // RTK_CH_BOX: the kernel's pcs count bytes
        } else if (kind == OP_SPHERE || kind == OP_SPHERE_MOVING) {
            const rtk_sphere& s = sc.spheres[payload];
            rec->set_d(0, s.center0.x); rec->set_d(1, s.center0.y); rec->set_d(2, s.center0.z);
            double* cont = reinterpret_cast<double*>(rec + 1);
            cont[0] = s.radius;
            cont[1] = 1.0 / s.radius;
            if (kind == OP_SPHERE_MOVING) {
                cont[2] = s.center_dir.x;
                cont[3] = s.center_dir.y;
                cont[4] = s.center_dir.z;  // first double of the third unit
            }
            rec->aux = (op.aux & 255u) | (uint32_t(s.material) << 8);
        }
    }
    extent_out = float(extent);
}

// Largest coordinate magnitude the f32 culling boxes of a scene have to cope with: every box bound and primitive
// extent (each in its own space), every instance offset, the eye.
static double scene_extent(const rtk_scene_desc& sc, double eye_extent) {
    double e = eye_extent;
    auto take = [&](double v) { e = std::max(e, std::fabs(v)); };
    for (int32_t i = 0; i < sc.n_bvh_boxes; i++) {
        const rtk_aabb& b = sc.bvh_boxes[i];
        for (double v : {b.xmin, b.xmax, b.ymin, b.ymax, b.zmin, b.zmax}) take(v);
    }
    for (int32_t i = 0; i < sc.n_spheres; i++) {
        const rtk_sphere& s = sc.spheres[i];
        for (double c : {s.center0.x, s.center0.y, s.center0.z, s.center0.x + s.center_dir.x, s.center0.y + s.center_dir.y, s.center0.z + s.center_dir.z})
            e = std::max(e, std::fabs(c) + std::fabs(s.radius));
    }
    for (int32_t i = 0; i < sc.n_quads; i++) {
        const rtk_quad& q = sc.quads[i];
        for (int k = 0; k < 4; k++) {
            const double a = (k & 1) ? 1.0 : 0.0, b = (k & 2) ? 1.0 : 0.0;
            take(q.Q.x + a * q.u.x + b * q.v.x); take(q.Q.y + a * q.u.y + b * q.v.y); take(q.Q.z + a * q.u.z + b * q.v.z);
        }
    }
    for (int32_t i = 0; i < sc.n_triangles; i++) {
        const rtk_triangle& t = sc.triangles[i];
        for (const rtk_vec3* p : {&t.p0, &t.p1, &t.p2}) { take(p->x); take(p->y); take(p->z); }
    }
    double offsets = 0;  // a ray's origin in an instance's space is at most this far beyond its world-space position
    for (int32_t i = 0; i < sc.n_translates; i++)
        offsets = std::max(offsets, std::max(std::fabs(sc.translates[i].offset.x), std::max(std::fabs(sc.translates[i].offset.y), std::fabs(sc.translates[i].offset.z))));
    return (e + offsets * double(kMaxChain)) * (sc.n_rotates > 0 ? 1.5 : 1.0);  // a rotation about y mixes x and z: sqrt(2) < 1.5
}

// The COMPACT program of rtk_device_layout.h for any scene whose boxes carry rtk_scene_optimize's margin: f32 culling
// boxes exactly as in build_mixed_program (rounded outward, grown by 2^-19 of the extent; same error budget -- under an
// instance transform the kernel applies it to the object-space ray and sends a ray whose object-space origin leaves
// [-extent, extent]^3 through the exact test), every primitive in f64, 16-byte units.
// One record of the COMPACT layout for op i (kind = its kind with moving spheres told apart) at `dst`; links (a box's or
// a MED_MID's `aux`) are translated through `unit_of_link`.
// COMPACT programs of the mesh family hold centre / half-extent box records (the rule of rtk_trace.hip kCompactChStatic)
static bool compact_ch_family(const Program& prog) {
    const uint32_t scene = prog.features & ~uint32_t(F_FMA_BOX | F_MATTE);
    return RTK_CH_COMPACT && scene != kFeatLean && (scene & ~kFeatQuadBox) != 0 && (scene & ~kFeatMesh) == 0;
}

static void write_compact_record(const rtk_scene_desc& sc, const Program& prog, size_t i, uint32_t kind, Unit16* dst, double margin,
                                 const std::vector<uint32_t>& unit_of_link, bool ch_boxes = false) {
    const Op& op = prog.ops[i];
    const uint32_t payload = op.kind_payload >> 4;
    MixedHead* head = reinterpret_cast<MixedHead*>(dst);
    double* more = reinterpret_cast<double*>(dst) + 4;  // payload element 3 onwards (byte 32)
    head->kind_payload = make_op(kind, payload);
    head->aux = op.aux;
    auto with_material = [&](int32_t material) { head->aux = (op.aux & 255u) | (uint32_t(material) << 8); };
    switch (kind) {
        case OP_BOX: {
            const rtk_aabb& b = sc.bvh_boxes[payload];
            if (ch_boxes) {
            // centre / half-extent form, as in build_mixed_program, except that the origin's share of the error ((2.5 + 1) 2^-23
            // |o / d|) is not the box's to carry: the ray does (slab_test32_che's slack, 2^-20 max |o / d|), so that a box grows
            // by 2^-21 of its OWN coordinates whatever the scene's extent (the Cornell quads' boxes are 1e-4 thick, at 555)
            const double lo[3] = {b.xmin, b.ymin, b.zmin}, hi[3] = {b.xmax, b.ymax, b.zmax};
            for (int ax = 0; ax < 3; ax++) {
                const double c = 0.5 * (lo[ax] + hi[ax]);
                const float cf = float(c);
                const double h = std::max(hi[ax] - double(cf), double(cf) - lo[ax]);
                const double grown = h + std::ldexp(std::fabs(double(cf)) + h, -21) + 1e-300;
                head->set_f(ax, cf);
                head->set_f(3 + ax, round_up(grown * (1.0 + 1e-7)));
            }
            } else {
            const double m = box_margin(b, margin);
            head->set_f(0, round_down(b.xmin - m)); head->set_f(1, round_up(b.xmax + m));
            head->set_f(2, round_down(b.ymin - m)); head->set_f(3, round_up(b.ymax + m));
            head->set_f(4, round_down(b.zmin - m)); head->set_f(5, round_up(b.zmax + m));
            }
            head->aux = unit_of_link[op.aux];
            break;
        }
        case OP_SPHERE:
        case OP_SPHERE_MOVING: {
            const rtk_sphere& s = sc.spheres[payload];
            head->set_d(0, s.center0.x); head->set_d(1, s.center0.y); head->set_d(2, s.center0.z);
            more[0] = s.radius;
            more[1] = 1.0 / s.radius;  // the factor of `(p - center) / radius` (vec3.h:91-93), once instead of per hit
            if (kind == OP_SPHERE_MOVING) { more[2] = s.center_dir.x; more[3] = s.center_dir.y; more[4] = s.center_dir.z; }
            with_material(s.material);
            break;
        }
        case OP_QUAD: {
            const rtk_quad& q = sc.quads[payload];
            const double vals[16] = {q.normal.x, q.normal.y, q.normal.z, q.D, q.Q.x, q.Q.y, q.Q.z, q.w.x, q.w.y, q.w.z,
                                     q.v.x, q.v.y, q.v.z, q.u.x, q.u.y, q.u.z};
            for (int e = 0; e < 3; e++) head->set_d(e, vals[e]);
            for (int e = 3; e < 16; e++) more[e - 3] = vals[e];
            with_material(q.material);
            break;
        }
        case OP_TRI: {
            const rtk_triangle& t = sc.triangles[payload];
            // v0v1 = p1 - p0, v0v2 = p2 - p0 (triangle.h:67-68) in double
            const double vals[9] = {t.p2.x - t.p0.x, t.p2.y - t.p0.y, t.p2.z - t.p0.z, t.p1.x - t.p0.x, t.p1.y - t.p0.y, t.p1.z - t.p0.z,
                                    t.p0.x, t.p0.y, t.p0.z};
            for (int e = 0; e < 3; e++) head->set_d(e, vals[e]);
            for (int e = 3; e < 9; e++) more[e - 3] = vals[e];
            with_material(t.material);
            break;
        }
        case OP_MED_MID: head->aux = unit_of_link[op.aux]; break;
        case OP_MED_END:
            head->set_d(0, sc.media[payload].neg_inv_density);
            with_material(sc.media[payload].material);
            break;
        case OP_MED_SPHERE: {
            const rtk_sphere& s = sc.spheres[prog.extra[i]];
            head->set_d(0, s.center0.x); head->set_d(1, s.center0.y); head->set_d(2, s.center0.z);
            more[0] = s.radius;
            more[1] = sc.media[payload].neg_inv_density;
            with_material(sc.media[payload].material);
            break;
        }
        default: break;
    }
}

static uint32_t compact_kind_of(const rtk_scene_desc& sc, const Op& op) {
    uint32_t kind = op.kind_payload & 15u;
    if (kind == OP_SPHERE) {
        const rtk_sphere& s = sc.spheres[op.kind_payload >> 4];
        if (s.center_dir.x != 0 || s.center_dir.y != 0 || s.center_dir.z != 0) kind = OP_SPHERE_MOVING;
    }
    return kind;
}

static void build_compact_program(const rtk_scene_desc& sc, const Program& prog, double eye_extent, std::vector<Unit16>& units, std::vector<uint32_t>& rank_of_unit,
                                  float& extent_out) {
    std::vector<uint32_t> unit_of_op(prog.ops.size() + 1, 0);
    for (size_t i = 0; i < prog.ops.size(); i++) unit_of_op[i + 1] = unit_of_op[i] + uint32_t(compact_units(compact_kind_of(sc, prog.ops[i])));
    const double extent = double(round_up(scene_extent(sc, eye_extent) * 1.0000001));
    const double margin = std::ldexp(extent, -19);
    units.assign(unit_of_op.back(), Unit16{{0u, 0u, 0u, 0u}});
    rank_of_unit.assign(unit_of_op.back(), 0u);
    for (size_t i = 0; i < prog.ops.size(); i++) {
        write_compact_record(sc, prog, i, compact_kind_of(sc, prog.ops[i]), &units[unit_of_op[i]], margin, unit_of_op, compact_ch_family(prog));
        rank_of_unit[unit_of_op[i]] = prog.ranks[i];
    }
    extent_out = float(extent);
}

// The same program in two parts, for COMPACT programs that do not fit one CU's LDS (rtk_device_layout.h, SceneView::
// program_hot): quads and triangles -- the bulky records, tested a few times per sample -- go to `cold` in program order;
// `hot` is the program without them, each run of consecutive quads (or triangles) replaced by ONE two-unit record {kind,
// count; aux = first unit of the run in `cold`}.  A run ends where the kind changes, at 255 primitives, and in front of any
// record some box or medium links to (so that every link target is the start of a record of the hot program).
// rank_of_id[unit of a hot record, or hot.size() + unit of a cold one] = the primitive's reference rank.
static void build_hot_cold_program(const rtk_scene_desc& sc, const Program& prog, double eye_extent, std::vector<Unit16>& hot, std::vector<Unit16>& cold,
                                   std::vector<uint32_t>& rank_of_id) {
    const size_t n = prog.ops.size();
    std::vector<uint32_t> kind(n);
    std::vector<char> is_target(n + 1, 0);
    for (size_t i = 0; i < n; i++) {
        kind[i] = compact_kind_of(sc, prog.ops[i]);
        if (kind[i] == OP_BOX || kind[i] == OP_MED_MID) is_target[prog.ops[i].aux] = 1;
    }
    auto is_cold = [&](size_t i) { return kind[i] == OP_QUAD || kind[i] == OP_TRI; };
    std::vector<uint32_t> hot_unit(n + 1, 0), cold_unit(n, 0), run_len(n, 0);
    uint32_t n_hot = 0, n_cold = 0;
    for (size_t i = 0; i < n;) {
        if (!is_cold(i)) {
            hot_unit[i] = n_hot;
            n_hot += uint32_t(compact_units(kind[i]));
            i++;
            continue;
        }
        size_t j = i;
        while (j < n && is_cold(j) && kind[j] == kind[i] && (j == i || !is_target[j]) && j - i < 255) {
            hot_unit[j] = n_hot;  // (only the run's first primitive can be a link target)
            cold_unit[j] = n_cold;
            n_cold += uint32_t(compact_units(kind[j]));
            j++;
        }
        run_len[i] = uint32_t(j - i);
        n_hot += 2;
        i = j;
    }
    hot_unit[n] = n_hot;
    const double extent = double(round_up(scene_extent(sc, eye_extent) * 1.0000001));
    const double margin = std::ldexp(extent, -19);
    hot.assign(n_hot, Unit16{{0u, 0u, 0u, 0u}});
    cold.assign(std::max<uint32_t>(n_cold, 1u), Unit16{{0u, 0u, 0u, 0u}});
    rank_of_id.assign(size_t(n_hot) + cold.size(), 0u);
    for (size_t i = 0; i < n; i++) {
        if (!is_cold(i)) {
            write_compact_record(sc, prog, i, kind[i], &hot[hot_unit[i]], margin, hot_unit);
            rank_of_id[hot_unit[i]] = prog.ranks[i];
            continue;
        }
        if (run_len[i] > 0) {
            MixedHead* run = reinterpret_cast<MixedHead*>(&hot[hot_unit[i]]);
            run->kind_payload = make_op(kind[i], run_len[i]);
            run->aux = cold_unit[i];
        }
        write_compact_record(sc, prog, i, kind[i], &cold[cold_unit[i]], margin, hot_unit);
        rank_of_id[size_t(n_hot) + cold_unit[i]] = prog.ranks[i];
    }
}

template <typename real>
int build_device_scene(const rtk_scene_desc& sc, const Program& prog, DeviceScene<real>& out, bool fast_order = false, bool want_mixed = false, double eye_extent = 0.0) {
    out.release();
    // ---- the fused traversal program: one or more slots per op, skip links
    // translated from op indices to slot indices
    std::vector<uint32_t> slot_of_op(prog.ops.size() + 1, 0);
    auto slot_kind = [&](const Op& op) -> uint32_t {
        uint32_t kind = op.kind_payload & 15u;
        if (kind == OP_SPHERE) {
            const rtk_sphere& s = sc.spheres[op.kind_payload >> 4];
            if (s.center_dir.x != 0 || s.center_dir.y != 0 || s.center_dir.z != 0) kind = OP_SPHERE_MOVING;
        }
        return kind;
    };
    for (size_t i = 0; i < prog.ops.size(); i++) slot_of_op[i + 1] = slot_of_op[i] + uint32_t(slots_of<real>(slot_kind(prog.ops[i])));
    std::vector<Slot<real>> slots(slot_of_op.back());
    std::memset(slots.data(), 0, slots.size() * sizeof(Slot<real>));
    auto pack = [&](Slot<real>* rec, const double* vals, int n) {
        for (int e = 0; e < n; e++) rec[e / Slot<real>::kReals].v[e % Slot<real>::kReals] = real(vals[e]);
    };
    for (size_t i = 0; i < prog.ops.size(); i++) {
        const Op& op = prog.ops[i];
        const uint32_t kind = slot_kind(op), payload = op.kind_payload >> 4;
        Slot<real>* rec = &slots[slot_of_op[i]];
        rec->kind_payload = make_op(kind, payload);
        rec->aux = op.aux;
        auto with_material = [&](int32_t material) { rec->aux = (op.aux & 255u) | (uint32_t(material) << 8); };
        switch (kind) {
            case OP_BOX: {
                const rtk_aabb& b = sc.bvh_boxes[payload];
                const double vals[6] = {b.xmin, b.xmax, b.ymin, b.ymax, b.zmin, b.zmax};
                pack(rec, vals, 6);
                rec->aux = slot_of_op[op.aux];
                break;
            }
            case OP_SPHERE:
            case OP_SPHERE_MOVING: {
                const rtk_sphere& s = sc.spheres[payload];
                const double vals[4] = {s.center0.x, s.center0.y, s.center0.z, s.radius};
                pack(rec, vals, 4);
                rec->v[4] = real(1) / real(s.radius);  // the factor of `(p - center) / radius` (vec3.h:91-93), once instead of per hit
                with_material(s.material);
                if (kind == OP_SPHERE_MOVING) {
                    const double dir[3] = {s.center_dir.x, s.center_dir.y, s.center_dir.z};
                    pack(rec + 1, dir, 3);
                }
                break;
            }
            case OP_QUAD: {
                const rtk_quad& q = sc.quads[payload];
                const double vals[16] = {q.normal.x, q.normal.y, q.normal.z, q.D, q.Q.x, q.Q.y, q.Q.z, q.w.x, q.w.y, q.w.z,
                                         q.v.x, q.v.y, q.v.z, q.u.x, q.u.y, q.u.z};
                pack(rec, vals, 16);
                with_material(q.material);
                break;
            }
            case OP_TRI: {
                const rtk_triangle& t = sc.triangles[payload];
                // v0v1 = p1 - p0, v0v2 = p2 - p0 (triangle.h:67-68) in double, then to `real`
                const double vals[9] = {t.p2.x - t.p0.x, t.p2.y - t.p0.y, t.p2.z - t.p0.z, t.p1.x - t.p0.x, t.p1.y - t.p0.y, t.p1.z - t.p0.z,
                                        t.p0.x, t.p0.y, t.p0.z};
                pack(rec, vals, 9);
                with_material(t.material);
                break;
            }
            case OP_MED_MID: rec->aux = slot_of_op[op.aux]; break;
            case OP_MED_END:
                rec->v[0] = real(sc.media[payload].neg_inv_density);
                with_material(sc.media[payload].material);
                break;
            case OP_MED_SPHERE: {
                const rtk_sphere& s = sc.spheres[prog.extra[i]];
                const double vals[5] = {s.center0.x, s.center0.y, s.center0.z, s.radius, sc.media[payload].neg_inv_density};
                pack(rec, vals, 5);
                with_material(sc.media[payload].material);
                break;
            }
            default: break;
        }
    }
    // Sphere, quad and medium records live entirely in their program slots; only triangles keep a side record
    // (normal + UVs for the deferred hit record).
    std::vector<TriRec<real>> tris(sc.n_triangles);
    for (int32_t i = 0; i < sc.n_triangles; i++) {
        const rtk_triangle& t = sc.triangles[i];
        TriRec<real>& r = tris[i];
        // v0v1 = p1 - p0, v0v2 = p2 - p0 (triangle.h:67-68) in double, then to `real`
        store3(r.p0, t.p0);
        store3(r.e1, rtk_vec3{t.p1.x - t.p0.x, t.p1.y - t.p0.y, t.p1.z - t.p0.z});
        store3(r.e2, rtk_vec3{t.p2.x - t.p0.x, t.p2.y - t.p0.y, t.p2.z - t.p0.z});
        store3(r.n, t.normal);
        for (int k = 0; k < 2; k++) { r.uv0[k] = t.uv0[k]; r.uv1[k] = t.uv1[k]; r.uv2[k] = t.uv2[k]; }
        r.material = t.material;
        r._pad = 0;
    }

    std::vector<MaterialRec<real>> mats(sc.n_materials);
    for (int32_t i = 0; i < sc.n_materials; i++) {
        const rtk_material& m = sc.materials[i];
        MaterialRec<real>& r = mats[i];
        store3(r.albedo, m.albedo);
        r.param = real(m.param);
        if (m.kind == RTK_MAT_DIELECTRIC) {
            // a dielectric has no albedo (material.h:47-65): the slots carry its per-face constants instead, computed
            // in the kernel's arithmetic type exactly as scatter() does per call -- 1/ri (material.h:50) and Schlick's r0^2 for ri = 1/index
            // (front face) and ri = index (material.h:71-72)
            const real index = real(m.param), inv = real(1) / index;
            real r0f = (real(1) - inv) / (real(1) + inv), r0b = (real(1) - index) / (real(1) + index);
            r0f = r0f * r0f;
            r0b = r0b * r0b;
            r.albedo[0] = inv;
            r.albedo[1] = r0f;
            r.albedo[2] = r0b;
        }
        r.kind = m.kind;
        r.tex = -1;
        r.needs_uv = 0;
        r._pad = 0;
        const bool textured = m.kind == RTK_MAT_LAMBERTIAN || m.kind == RTK_MAT_DIFFUSE_LIGHT || m.kind == RTK_MAT_ISOTROPIC;
        if (textured) {
            const rtk_texture& t = sc.textures[m.texture];
            if (t.kind == RTK_TEX_SOLID) {
                store3(r.albedo, t.color);  // solid_color::value is the constant (texture.h:26): fold it in
            } else {
                r.tex = m.texture;
                r.needs_uv = texture_needs_uv(sc, m.texture) ? 1 : 0;
            }
        }
    }
    std::vector<TextureRec<real>> texs(sc.n_textures);
    for (int32_t i = 0; i < sc.n_textures; i++) {
        const rtk_texture& t = sc.textures[i];
        TextureRec<real>& r = texs[i];
        store3(r.color, t.color);
        r.param = real(t.param);
        r.kind = t.kind; r.even = t.even; r.odd = t.odd; r.image = t.image;
    }
    std::vector<ImageRec> images(sc.n_images);
    for (int32_t i = 0; i < sc.n_images; i++) images[i] = ImageRec{sc.images[i].width, sc.images[i].height, sc.images[i].texel_offset};
    std::vector<uint8_t> texels(sc.texels, sc.texels + sc.n_texel_bytes);
    std::vector<PerlinRec<real>> perlins(sc.n_perlins);
    for (int32_t i = 0; i < sc.n_perlins; i++) {
        for (int k = 0; k < 256; k++) {
            for (int c = 0; c < 3; c++) perlins[i].randvec[k][c] = real(sc.perlins[i].randvec[k][c]);
            perlins[i].perm_x[k] = sc.perlins[i].perm_x[k];
            perlins[i].perm_y[k] = sc.perlins[i].perm_y[k];
            perlins[i].perm_z[k] = sc.perlins[i].perm_z[k];
        }
    }
    std::vector<ChainRec<real>> chains(prog.chains.size());
    for (size_t i = 0; i < prog.chains.size(); i++) {
        ChainRec<real>& r = chains[i];
        std::memset(&r, 0, sizeof r);
        r.count = int32_t(prog.chains[i].size());
        for (int k = 0; k < r.count; k++) {
            const ChainStep& st = prog.chains[i][k];
            r.is_rotate[k] = st.rotate ? 1 : 0;
            if (st.rotate) {
                r.a[k] = real(sc.rotates[st.index].sin_theta);
                r.b[k] = real(sc.rotates[st.index].cos_theta);
            } else {
                r.a[k] = real(sc.translates[st.index].offset.x);
                r.b[k] = real(sc.translates[st.index].offset.y);
                r.c[k] = real(sc.translates[st.index].offset.z);
            }
        }
    }
    std::vector<LightRec<real>> lights(sc.n_lights);
    for (int32_t i = 0; i < sc.n_lights; i++) {
        store3(lights[i].position, sc.lights[i].position);
        store3(lights[i].intensity, sc.lights[i].intensity);
        lights[i].size = real(sc.lights[i].size);
    }
    int rc;
    if ((rc = out.upload(slots, &out.view.program)) != RTK_OK) return rc;
    if ((rc = out.upload(tris, &out.view.tris)) != RTK_OK) return rc;
    if ((rc = out.upload(mats, &out.view.materials)) != RTK_OK) return rc;
    if ((rc = out.upload(texs, &out.view.textures)) != RTK_OK) return rc;
    if ((rc = out.upload(images, &out.view.images)) != RTK_OK) return rc;
    if ((rc = out.upload(texels, &out.view.texels)) != RTK_OK) return rc;
    if ((rc = out.upload(perlins, &out.view.perlins)) != RTK_OK) return rc;
    if ((rc = out.upload(chains, &out.view.chains)) != RTK_OK) return rc;
    if ((rc = out.upload(lights, &out.view.lights)) != RTK_OK) return rc;
    out.view.n_slots = int32_t(slots.size());
    out.view.n_lights = sc.n_lights;
    out.view.n_materials = sc.n_materials;
    out.view.n_perlins = sc.n_perlins;
    out.view.n_chains = int32_t(chains.size());
    out.view.program_mixed = nullptr;
    out.view.n_units = 0;
    out.view.extent = 0.0f;
    // Programs that cannot be staged in LDS whole: the box slots alone, with the tables that find them (F_LDS_BOXES).
    out.view.box_cache = nullptr;
    out.view.kind_words = nullptr;
    out.view.box_rank = nullptr;
    out.view.n_cached_boxes = out.view.n_kind_words = out.view.n_rank_words = 0;
    if (slots.size() * sizeof(Slot<real>) + mats.size() * sizeof(MaterialRec<real>) > size_t(kLdsBytesPerCU)) {
        std::vector<BoxRec<real>> boxes;
        std::vector<uint32_t> kind_words((slots.size() + 7) / 8, 0u);
        std::vector<uint2> rank((slots.size() + 31) / 32, uint2{0u, 0u});
        for (size_t pc = 0; pc < slots.size();) {
            const uint32_t kind = slots[pc].kind_payload & 15u;
            kind_words[pc >> 3] |= kind << ((pc & 7) * 4);
            if (kind == OP_BOX) {
                rank[pc >> 5].x |= 1u << (pc & 31);
                BoxRec<real> b{};
                for (int k = 0; k < 6; k++) b.v[k] = slots[pc].v[k];
                b.aux = slots[pc].aux;
                boxes.push_back(b);
            }
            pc += size_t(slots_of<real>(kind));
        }
        uint32_t before = 0;
        for (auto& r : rank) {
            r.y = before;
            before += uint32_t(__builtin_popcount(r.x));
        }
        const size_t bytes = boxes.size() * sizeof(BoxRec<real>) + ((kind_words.size() * 4 + 7) & ~size_t(7)) + rank.size() * 8;
        if (getenv("RTK_DEBUG")) fprintf(stderr, "[rtk] %zu-byte reals: %zu slots, %zu box slots, boxes + tables %zu B\n", sizeof(real), slots.size(), boxes.size(), bytes);
        if (!boxes.empty() && bytes + 64 <= size_t(kLdsBytesPerCU)) {
            if ((rc = out.upload(boxes, &out.view.box_cache)) != RTK_OK) return rc;
            if ((rc = out.upload(kind_words, &out.view.kind_words)) != RTK_OK) return rc;
            if ((rc = out.upload(rank, &out.view.box_rank)) != RTK_OK) return rc;
            out.view.n_cached_boxes = int32_t(boxes.size());
            out.view.n_kind_words = int32_t(kind_words.size());
            out.view.n_rank_words = int32_t(rank.size());
        }
    }
    out.view.program_compact = nullptr;
    out.view.n_units16 = 0;
    out.view.compact_ch = 0;
    out.view.program_hot = nullptr;
    out.view.program_cold = nullptr;
    out.view.n_hot_units = out.view.n_cold_units = 0;
    out.view.tie_rank_hot = nullptr;
    out.view.tie_rank = nullptr;
    out.view.tie_rank_slot = nullptr;
    if (fast_order) {  // reference ranks of the primitive records, for exact ties (see SceneView::tie_rank)
        std::vector<uint32_t> rank_of_slot(slots.size(), 0u);
        bool any = false;
        for (size_t i = 0; i < prog.ops.size(); i++) {
            rank_of_slot[slot_of_op[i]] = prog.ranks[i];
            any = any || prog.ranks[i] != 0;
        }
        if (any && (rc = out.upload(rank_of_slot, &out.view.tie_rank_slot)) != RTK_OK) return rc;
    }
    if constexpr (sizeof(real) == 8) {
        if (fast_order && want_mixed) {
            std::vector<MixedHead> units;
            std::vector<uint32_t> ranks;
            build_mixed_program(sc, prog, eye_extent, units, ranks, out.view.extent);
            if ((rc = out.upload(units, &out.view.program_mixed)) != RTK_OK) return rc;
            if ((rc = out.upload(ranks, &out.view.tie_rank)) != RTK_OK) return rc;
            out.view.n_units = int32_t(units.size());
        } else if (fast_order) {
            std::vector<Unit16> units;
            std::vector<uint32_t> ranks;
            build_compact_program(sc, prog, eye_extent, units, ranks, out.view.extent);
            if ((rc = out.upload(units, &out.view.program_compact)) != RTK_OK) return rc;
            if ((rc = out.upload(ranks, &out.view.tie_rank)) != RTK_OK) return rc;
            out.view.n_units16 = int32_t(units.size());
            out.view.compact_ch = compact_ch_family(prog) ? 1 : 0;
            // the same program in its hot/cold form: for programs too large for LDS -- and, through variant bit 23, for tests
            // of that form on scenes small enough for the oracle (only the full-feature kernel family has it)
            if ((prog.features & ~kFeatQuadBox) != 0 && (prog.features & ~kFeatMesh) != 0) {
                std::vector<Unit16> hot, cold;
                std::vector<uint32_t> ranks_by_id;
                build_hot_cold_program(sc, prog, eye_extent, hot, cold, ranks_by_id);
                if (getenv("RTK_DEBUG")) fprintf(stderr, "[rtk] COMPACT program %zu B; hot part %zu B, cold part %zu B\n", units.size() * 16, hot.size() * 16, cold.size() * 16);
                if (hot.size() * sizeof(Unit16) + mats.size() * sizeof(MaterialRec<real>) + 64 <= size_t(kLdsBytesPerCU) && hot.size() < (size_t(1) << 24)) {
                    if ((rc = out.upload(hot, &out.view.program_hot)) != RTK_OK) return rc;
                    if ((rc = out.upload(cold, &out.view.program_cold)) != RTK_OK) return rc;
                    if ((rc = out.upload(ranks_by_id, &out.view.tie_rank_hot)) != RTK_OK) return rc;
                    out.view.n_hot_units = int32_t(hot.size());
                    out.view.n_cold_units = int32_t(cold.size());
                }
            }
        }
    }
    return RTK_OK;
}

template <typename real>
CameraRec<real> to_device_camera(const rtk_camera& c) {
    CameraRec<real> r;
    store3(r.background, c.background);
    store3(r.center, c.center);
    store3(r.pixel00, c.pixel00_loc);
    store3(r.du, c.pixel_delta_u);
    store3(r.dv, c.pixel_delta_v);
    store3(r.disk_u, c.defocus_disk_u);
    store3(r.disk_v, c.defocus_disk_v);
    r.defocus_angle = real(c.defocus_angle);
    r.samples_scale = real(c.pixel_samples_scale);
    r.width = c.image_width;
    r.height = c.image_height;
    r.spp = c.samples_per_pixel;
    r.max_depth = c.max_depth;
    return r;
}

}  // namespace

struct rtk_ctx {
    int device = 0;
    bool has_scene = false;
    uint32_t features = 0;
    int32_t n_ops = 0;
    int32_t n_materials = 0, n_textures = 0;  // of the uploaded description (index checks of the known-answer entry points)
    DeviceScene<double> scene64;
    DeviceScene<float> scene32;
    // Words the persistent waves pull tile indices from; one per launch in a
    // small ring so that back-to-back launches on a stream never share one.
    unsigned int* tile_counters = nullptr;
    unsigned int next_counter = 0;
    // Camera records in device memory (the kernel reads them with scalar loads),
    // same ring discipline; the host copies stay alive until the async copy ran.
    unsigned char* d_cameras = nullptr;
    // Partial-sum workspace [work item][3][64] reals, grown on demand and reused by
    // successive launches (they are ordered on the caller's stream).
    void* d_partial = nullptr;
    size_t partial_bytes = 0;
    // Longest-processing-time-first tile order, learned from the previous frame of the same shape: the render
    // kernel accumulates a per-tile cost, rtk_tile_order_kernel turns it into the hand-out order of the next launch.
    unsigned int* d_tile_cost = nullptr;
    int32_t* d_tile_order = nullptr;
    int order_capacity = 0;
    bool order_valid = false;
    int order_shape[5] = {0, 0, 0, 0, 0};  // width, height, rank, n_ranks, tiles: what the stored order was measured on
    std::vector<CameraRec<double>> h_cameras64;
    std::vector<CameraRec<float>> h_cameras32;
    // The camera of the previous launch per arithmetic type: a frame loop renders the same camera again and again, and
    // re-sending 256 bytes through a pageable-memory copy costs a host/stream round trip per frame for nothing.
    unsigned int next_camera = 0;
    bool cam_valid[2] = {false, false};
    rtk_camera cam_last[2];
    unsigned int cam_slot[2] = {0, 0};
    // Progress reporting (rtk_set_progress_callback): the work-item counter of the last launch and how many items it
    // hands out; a blocking render polls the counter with a 4-byte device-to-host copy on a stream of its own.
    rtk_progress_fn progress_fn = nullptr;
    void* progress_user = nullptr;
    int progress_interval_ms = 100;
    // a frame is up to kMaxPasses launches (one per range of sample chunks): the work-item counter of each and what it hands out
    const unsigned int* last_tile_counter[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int64_t last_pass_items[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int last_n_passes = 0;
    int64_t last_n_items = 0;               // of the whole frame
    hipStream_t progress_stream = nullptr;
    unsigned int* progress_word = nullptr;  // pinned host memory, one word per pass
    bool progress_pending = false;
};
// Sample chunks per launch: the partial-sum workspace holds up to this many planes [local tile][3][64] at 1920x1080 f64 (+ one
// for the running sum of a frame that needs several launches): 21 + 1 planes = 1.09 GB, where the 63 chunks of a 1000-spp
// frame used to take 3.14 GB per context.  Every launch has a fixed cost -- staging the program, the drain of its last long
// paths -- measured on the full-size frames (same box, same image): one launch / passes of 16 chunks: C5 999.6 / 1014.6 ms,
// C3 247.7 / 254.5 ms; 21 chunks per pass makes a 1000-spp frame three launches instead of four.  Frames of up to 168 spp
// (C2: 13 chunks) are one launch as before.
constexpr int kMaxPlanesPerPass = 21;
// The plane count is a BUDGET, not a constant: 22 planes of a 1920x1080 f64 frame = 1.095 GB per context.  A smaller
// frame -- 800x800, or one rank's share of the tiles on several GPUs -- gets as many planes as that budget holds (up to all
// 64 chunks: the Cornell box at 1000 spp is one launch again, 247.7 instead of 251.8 ms), a larger one at least 8.
constexpr size_t kWorkspaceBudgetBytes = size_t(22) * 32400 * 192 * 8;
constexpr int kMinPlanesPerPass = 8;
constexpr int kMaxPasses = (rtk::kMaxChunks + kMinPlanesPerPass - 1) / kMinPlanesPerPass;
static int planes_per_pass_for(size_t plane_bytes, int variant) {
    if (variant & (1 << 24)) return rtk::kMaxChunks;  // variant bit 24: one pass whatever the chunk count (tests: same image)
    if (plane_bytes == 0) return rtk::kMaxChunks;
    const size_t fit = kWorkspaceBudgetBytes / plane_bytes;  // planes the budget holds, one of them the running sum
    const int planes = fit > size_t(rtk::kMaxChunks) ? rtk::kMaxChunks : (fit < size_t(kMinPlanesPerPass + 1) ? kMinPlanesPerPass : int(fit) - 1);
    return planes;
}
// Samples per chunk: 8, or the smallest size that keeps a pixel's chunks within kMaxChunks -- a function of spp only.
static int chunk_size_for(int spp, int variant) {
    const int ab = (variant >> 3) & 3;  // tools/: chunk size A/B (0 = default 8, 1 = 4, 2 = 2, 3 = 16)
    int size = ab == 0 ? 8 : (ab == 1 ? 4 : (ab == 2 ? 2 : 16));
    if (variant & 2) size = spp;  // variant bit 1: one lane per pixel for all samples (tests)
    while ((spp + size - 1) / size > rtk::kMaxChunks) size++;
    return size;
}
static_assert(kMaxPasses <= 8, "rtk_ctx keeps eight pass counters");
constexpr unsigned int kCounterRing = 256;
constexpr size_t kCameraStride = 256;
static_assert(sizeof(CameraRec<double>) <= kCameraStride, "camera stride");

namespace rtk {

int ctx_device(const rtk_ctx* ctx) { return ctx->device; }

// Block until every stream[i] (a stream of ctxs[i]'s device) has drained.  When ctxs[0] has a progress callback, the
// work-item counters of the launches in flight are sampled meanwhile -- a 4-byte device-to-host copy per context on a
// stream of its own (the copy engine, not a CU; the persistent kernel is not touched) -- and the callback gets the sums.
// A sample that has not landed by the next tick is simply skipped, so a copy stuck behind the kernel can only mean
// fewer reports, never a stall.
int wait_with_progress(rtk_ctx* const* ctxs, const hipStream_t* streams, int n) {
    rtk_ctx* reporter = n > 0 ? ctxs[0] : nullptr;
    if (!reporter || !reporter->progress_fn) {
        for (int i = 0; i < n; i++) {
            hipError_t e = hipSetDevice(ctxs[i]->device);
            if (e == hipSuccess) e = hipStreamSynchronize(streams[i]);
            if (e != hipSuccess) return fail(RTK_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
        }
        return RTK_OK;
    }
    int64_t total = 0;
    std::vector<int64_t> seen(size_t(n), 0);
    for (int i = 0; i < n; i++) {
        rtk_ctx* c = ctxs[i];
        total += c->last_n_items;
        if (!c->progress_stream) {
            (void)hipSetDevice(c->device);
            if (hipStreamCreateWithFlags(&c->progress_stream, hipStreamNonBlocking) != hipSuccess) c->progress_stream = nullptr;
            if (hipHostMalloc(reinterpret_cast<void**>(&c->progress_word), 8 * sizeof(unsigned int), hipHostMallocDefault) != hipSuccess) c->progress_word = nullptr;
        }
        c->progress_pending = false;
    }
    const auto tick = std::chrono::milliseconds(reporter->progress_interval_ms);
    auto next_report = std::chrono::steady_clock::now();
    for (;;) {
        bool all_done = true;
        for (int i = 0; i < n; i++) {
            (void)hipSetDevice(ctxs[i]->device);
            const hipError_t q = hipStreamQuery(streams[i]);
            if (q == hipErrorNotReady) all_done = false;
            else if (q != hipSuccess) return fail(RTK_ERR_HIP, "hipStreamQuery failed: %s", hipGetErrorString(q));
        }
        if (all_done) break;
        if (std::chrono::steady_clock::now() >= next_report) {
            int64_t done = 0;
            for (int i = 0; i < n; i++) {
                rtk_ctx* c = ctxs[i];
                if (c->progress_stream && c->progress_word && c->last_n_passes > 0) {
                    (void)hipSetDevice(c->device);
                    if (c->progress_pending && hipStreamQuery(c->progress_stream) == hipSuccess) {
                        int64_t sum = 0;  // a pass that has not started yet still shows the zero its counter was reset to at enqueue time ... or an older launch's count: clamp, and count a pass only once every earlier one is complete
                        bool earlier_done = true;
                        for (int p = 0; p < c->last_n_passes; p++) {
                            const int64_t v = earlier_done ? std::min<int64_t>(int64_t(c->progress_word[p]), c->last_pass_items[p]) : 0;
                            sum += v;
                            earlier_done = earlier_done && v == c->last_pass_items[p];
                        }
                        seen[size_t(i)] = std::max(seen[size_t(i)], sum);
                        c->progress_pending = false;
                    }
                    if (!c->progress_pending) {
                        bool ok = true;
                        for (int p = 0; p < c->last_n_passes && ok; p++)
                            ok = hipMemcpyAsync(c->progress_word + p, c->last_tile_counter[p], sizeof(unsigned int), hipMemcpyDeviceToHost, c->progress_stream) == hipSuccess;
                        c->progress_pending = ok;
                    }
                }
                done += seen[size_t(i)];
            }
            reporter->progress_fn(done, total, reporter->progress_user);
            next_report = std::chrono::steady_clock::now() + tick;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    for (int i = 0; i < n; i++)
        if (ctxs[i]->progress_stream) {
            (void)hipSetDevice(ctxs[i]->device);
            (void)hipStreamSynchronize(ctxs[i]->progress_stream);
            ctxs[i]->progress_pending = false;
        }
    reporter->progress_fn(total, total, reporter->progress_user);
    return RTK_OK;
}

}  // namespace rtk

namespace {
// Device buffers of a known-answer call: released on every path out.
struct KatBuffers {
    std::vector<void*> ptrs;
    hipError_t err = hipSuccess;
    template <typename T>
    T* in(const T* host, size_t count) {
        T* d = out<T>(count);
        if (d && err == hipSuccess) err = hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice);
        return d;
    }
    template <typename T>
    T* out(size_t count) {
        void* d = nullptr;
        if (err == hipSuccess) err = hipMalloc(&d, count * sizeof(T) > 0 ? count * sizeof(T) : 16);
        if (d) ptrs.push_back(d);
        return static_cast<T*>(d);
    }
    template <typename T>
    void back(T* host, const T* dev, size_t count) {
        if (err == hipSuccess) err = hipMemcpy(host, dev, count * sizeof(T), hipMemcpyDeviceToHost);
    }
    ~KatBuffers() {
        for (void* p : ptrs) (void)hipFree(p);
    }
};
}  // namespace

extern "C" {

int rtk_abi_version(void) { return RTK_ABI_VERSION; }

const char* rtk_last_error(void) { return g_error.c_str(); }

int rtk_init(int device, rtk_ctx** out_ctx) {
    if (!out_ctx) return fail(RTK_ERR_INVALID, "rtk_init: out_ctx is null");
    *out_ctx = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(RTK_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(RTK_ERR_INVALID, "rtk_init: device %d out of range (0..%d)", device, count - 1);
    hipDeviceProp_t prop;
    RTK_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RTK_ERR_NO_DEVICE, "device %d is %s; the kernels are built for gfx950 (MI355X) only", device, prop.gcnArchName);
    RTK_HIP(hipSetDevice(device));
    auto* ctx = new rtk_ctx;
    ctx->device = device;
    hipError_t me = hipMalloc(reinterpret_cast<void**>(&ctx->tile_counters), kCounterRing * sizeof(unsigned int));
    if (me == hipSuccess) me = hipMalloc(reinterpret_cast<void**>(&ctx->d_cameras), kCounterRing * kCameraStride);
    if (me != hipSuccess) {
        if (ctx->tile_counters) (void)hipFree(ctx->tile_counters);
        delete ctx;
        return fail(RTK_ERR_HIP, "hipMalloc(launch state) failed: %s", hipGetErrorString(me));
    }
    ctx->h_cameras64.resize(kCounterRing);
    ctx->h_cameras32.resize(kCounterRing);
    *out_ctx = ctx;
    return RTK_OK;
}

int rtk_destroy(rtk_ctx* ctx) {
    if (!ctx) return RTK_OK;
    (void)hipSetDevice(ctx->device);
    ctx->scene64.release();
    ctx->scene32.release();
    if (ctx->tile_counters) (void)hipFree(ctx->tile_counters);
    if (ctx->d_cameras) (void)hipFree(ctx->d_cameras);
    if (ctx->d_partial) (void)hipFree(ctx->d_partial);
    if (ctx->d_tile_cost) (void)hipFree(ctx->d_tile_cost);
    if (ctx->d_tile_order) (void)hipFree(ctx->d_tile_order);
    if (ctx->progress_stream) (void)hipStreamDestroy(ctx->progress_stream);
    if (ctx->progress_word) (void)hipHostFree(ctx->progress_word);
    delete ctx;
    return RTK_OK;
}

int rtk_set_progress_callback(rtk_ctx* ctx, rtk_progress_fn fn, void* user, int interval_ms) {
    if (!ctx) return fail(RTK_ERR_INVALID, "rtk_set_progress_callback: null context");
    ctx->progress_fn = fn;
    ctx->progress_user = user;
    ctx->progress_interval_ms = interval_ms > 0 ? interval_ms : 100;
    return RTK_OK;
}

// How a description's hierarchy was made, which decides the slab tests the kernels may use on it.
enum UploadOrder {
    kOrderReference,  // the caller's own boxes: aabb::hit as written (aabb.h:61-85)
    kOrderFast,       // rtk_scene_optimize output, every box grown by the pass's margin: fused / f32 culling tests
};
static int upload_scene(rtk_ctx* ctx, const rtk_scene_desc* scene, UploadOrder order, double eye_extent = 0.0);

int rtk_scene_upload(rtk_ctx* ctx, const rtk_scene_desc* scene) {
    if (!ctx || !scene) return fail(RTK_ERR_INVALID, "rtk_scene_upload: null argument");
    return upload_scene(ctx, scene, kOrderReference);
}

int rtk_scene_upload_fast(rtk_ctx* ctx, const rtk_scene_desc* scene, const rtk_optimize_opts* opts, rtk_optimize_info* info) {
    if (!ctx || !scene) return fail(RTK_ERR_INVALID, "rtk_scene_upload_fast: null argument");
    rtk_scene_desc* fast = nullptr;
    int rc = rtk_scene_optimize(scene, opts, &fast, info);
    if (rc != RTK_OK) return fail(rc, "rtk_scene_upload_fast: rtk_scene_optimize rejected the scene description");
    // every box of `fast` was grown by rtk_scene_optimize's margin: the kernels may use the fused slab test
    double eye_extent = 0.0;
    if (opts && opts->has_eye) eye_extent = std::max(std::fabs(opts->eye.x), std::max(std::fabs(opts->eye.y), std::fabs(opts->eye.z)));
    rc = upload_scene(ctx, fast, kOrderFast, eye_extent);
    rtk_scene_optimized_free(fast);
    return rc;
}

int rtk_scene_upload_optimized(rtk_ctx* ctx, const rtk_scene_desc* optimized, const rtk_optimize_opts* opts) {
    if (!ctx || !optimized) return fail(RTK_ERR_INVALID, "rtk_scene_upload_optimized: null argument");
    double eye_extent = 0.0;
    if (opts && opts->has_eye) eye_extent = std::max(std::fabs(opts->eye.x), std::max(std::fabs(opts->eye.y), std::fabs(opts->eye.z)));
    return upload_scene(ctx, optimized, kOrderFast, eye_extent);
}

// Everything that can be decided without a device: table validation and the compilation of the traversal program.
static int compile_scene(const rtk_scene_desc* scene, Program& prog) {
    int rc = validate_tables(*scene);
    if (rc != RTK_OK) return rc;
    Compiler comp{*scene, prog};
    comp.chain_id(Chain{});  // chain 0 = world space
    comp.emit(scene->root, Chain{}, false, 0);
    if (prog.error_code != RTK_OK) return fail(prog.error_code, "rtk_scene_upload: %s", prog.error.c_str());
    if (prog.n_primitive_ops == 0) return fail(RTK_ERR_INVALID, "rtk_scene_upload: no primitive reachable from the root");
    // A switch back to world space that is the program's last record serves nobody: what follows is OP_END, the shade step works
    // from the world ray and the hit record's own chain, and begin_segment sets the object ray anew.  It is a scheduler step of
    // its own for every ray that entered the last instance (the Cornell box: a fifth of its chain steps), so it is dropped and
    // the skip links that pointed behind it are moved up.  (RTK_KEEP_LAST_CHAIN=1 keeps it: tools/, A/B.)
    if (prog.ops.size() > 1 && (prog.ops.back().kind_payload & 15u) == OP_CHAIN && (prog.ops.back().kind_payload >> 4) == 0u && prog.ops.back().aux == 0u &&
        !getenv("RTK_KEEP_LAST_CHAIN")) {
        const uint32_t behind = uint32_t(prog.ops.size());
        prog.ops.pop_back();
        prog.ranks.pop_back();
        prog.extra.pop_back();
        for (Op& op : prog.ops) {
            const uint32_t kind = op.kind_payload & 15u;
            if ((kind == OP_BOX || kind == OP_MED_MID) && op.aux == behind) op.aux = behind - 1;
        }
    }
    prog.ops.push_back(Op{make_op(OP_END, 0), 0});
    prog.ranks.push_back(0);
    prog.extra.push_back(0);
    return RTK_OK;
}

int rtk_scene_validate(const rtk_scene_desc* scene, int32_t* n_program_ops) {
    if (!scene) return fail(RTK_ERR_INVALID, "rtk_scene_validate: null argument");
    try {
        Program prog;
        const int rc = compile_scene(scene, prog);
        if (rc == RTK_OK && n_program_ops) *n_program_ops = int32_t(prog.ops.size());
        return rc;
    } catch (const std::bad_alloc&) {
        return fail(RTK_ERR_UNSUPPORTED, "rtk_scene_validate: out of memory while compiling the traversal program");
    }
}

static int upload_scene(rtk_ctx* ctx, const rtk_scene_desc* scene, UploadOrder order, double eye_extent) {
    Program prog;
    int rc = compile_scene(scene, prog);
    if (rc != RTK_OK) return rc;
    // (Media scenes included: rtk_scene_optimize keeps a medium's POSITION in the reference's order; the boxes around it may
    // be as conservative as any other -- a medium the reference would have skipped returns false before it draws.)
    const bool fast_order = order == kOrderFast;
    uint32_t hierarchy_flags = fast_order ? uint32_t(F_FMA_BOX) : 0u;
    for (int32_t i = 0; i < scene->n_materials; i++) {
        const rtk_material& m = scene->materials[i];
        if (m.kind == RTK_MAT_DIFFUSE_LIGHT || m.kind == RTK_MAT_ISOTROPIC || m.kind == RTK_MAT_SPECULAR) prog.features |= F_EXOTIC_MAT;
        const bool textured = m.kind == RTK_MAT_LAMBERTIAN || m.kind == RTK_MAT_DIFFUSE_LIGHT || m.kind == RTK_MAT_ISOTROPIC;
        if (textured && scene->textures[m.texture].kind != RTK_TEX_SOLID) prog.features |= F_TEXTURE;
    }
    if (scene->n_lights > 0) prog.features |= F_LIGHTS;
    bool matte = true;  // only lambertians and lights: the quad/box subset kernel drops its glossy branches (F_MATTE)
    for (int32_t i = 0; i < scene->n_materials; i++)
        if (scene->materials[i].kind != RTK_MAT_LAMBERTIAN && scene->materials[i].kind != RTK_MAT_DIFFUSE_LIGHT) matte = false;
    if (matte) hierarchy_flags |= F_MATTE;
#ifndef RTK_NO_SPHERE_MEDIA_ONLY   // (A/B builds)
    if ((prog.features & F_MEDIA) != 0 && !prog.has_media_bracket) hierarchy_flags |= F_SPHERE_MEDIA_ONLY;
#endif

    RTK_HIP(hipSetDevice(ctx->device));
    ctx->has_scene = false;
    ctx->order_valid = false;  // a new scene: tile costs measured on the old one mean nothing
    // sphere-only scenes in the fast order additionally get the MIXED program (f32 culling boxes) for the f64 kernels
    // ... every other scene of the fast order the COMPACT program (f32 culling boxes, f64 primitives, 16-byte units)
    const bool want_mixed = fast_order && prog.features == kFeatLean && prog.chains.size() <= 1;
    if ((rc = build_device_scene<double>(*scene, prog, ctx->scene64, fast_order, want_mixed, eye_extent)) != RTK_OK) return rc;
    if ((rc = build_device_scene<float>(*scene, prog, ctx->scene32, fast_order)) != RTK_OK) return rc;
    ctx->features = prog.features | hierarchy_flags;
    ctx->n_ops = int32_t(prog.ops.size());
    ctx->n_materials = scene->n_materials;
    ctx->n_textures = scene->n_textures;
    ctx->has_scene = true;
    return RTK_OK;
}

int64_t rtk_tiles_per_rank(int image_width, int image_height, int n_ranks) {
    if (image_width <= 0 || image_height <= 0 || n_ranks <= 0) return 0;
    const int64_t tiles = int64_t((image_width + RTK_TILE_W - 1) / RTK_TILE_W) * ((image_height + RTK_TILE_H - 1) / RTK_TILE_H);
    return (tiles + n_ranks - 1) / n_ranks;
}

int rtk_render_device(rtk_ctx* ctx, const rtk_camera* cam, const rtk_render_opts* opts, void* d_linear, uint8_t* d_rgb8, rtk_work_counters* d_counters) {
    if (!ctx || !cam || !opts) return fail(RTK_ERR_INVALID, "rtk_render_device: null argument");
    if (!ctx->has_scene) return fail(RTK_ERR_NO_SCENE, "rtk_render_device: no scene uploaded");
    if (cam->image_width <= 0 || cam->image_height <= 0 || cam->samples_per_pixel <= 0 || cam->samples_per_pixel > 32767 || cam->max_depth < 0)
        return fail(RTK_ERR_INVALID, "rtk_render_device: bad camera dimensions (samples_per_pixel must be 1..32767)");
    if (opts->n_ranks < 1 || opts->rank < 0 || opts->rank >= opts->n_ranks) return fail(RTK_ERR_INVALID, "rtk_render_device: bad rank %d of %d", opts->rank, opts->n_ranks);
    const bool compact_out = opts->n_ranks > 1 || (opts->variant & (1 << 22)) != 0;  // variant bit 22: the tile-buffer layout for a single rank too
    if (compact_out && d_rgb8) return fail(RTK_ERR_INVALID, "rtk_render_device: d_rgb8 must be null when n_ranks > 1 (use rtk_tiles_unpermute)");
    if (opts->count_work && !d_counters) return fail(RTK_ERR_INVALID, "rtk_render_device: count_work needs d_counters");
    if (opts->real_mode != RTK_REAL_F64 && opts->real_mode != RTK_REAL_F32) return fail(RTK_ERR_INVALID, "rtk_render_device: unknown real_mode %d", opts->real_mode);
    RTK_HIP(hipSetDevice(ctx->device));
    TileMap tm;
    tm.tiles_x = (cam->image_width + RTK_TILE_W - 1) / RTK_TILE_W;
    tm.tiles_y = (cam->image_height + RTK_TILE_H - 1) / RTK_TILE_H;
    tm.rank = opts->rank;
    tm.n_ranks = opts->n_ranks;
    tm.n_tiles_local = int32_t(rtk_tiles_per_rank(cam->image_width, cam->image_height, opts->n_ranks));
    tm.compact = compact_out ? 1 : 0;
    tm.order_in_lds = 0;
    tm.order_lds_offset = 0;
    tm.mats_lds_offset = 0;
    // Split every pixel's samples into chunks of 8 (at most 64 chunks) so that no lane is stuck with a whole
    // heavy pixel: a glass pixel's samples cost ~0.2 ms each, and the largest (pixel, chunk) bounds the end of
    // the frame whatever the GPU count.  Measured on C2 (v16 kernel time, N = 1 / one of 8 shards): 4-sample chunks
    // 22.8 / 3.21 ms, 8-sample 21.8 / 3.21 ms, 16-sample 21.7 / 3.60 ms (with the v9 kernel, whose refills were
    // dearer, 4 was the optimum).  A function of spp only.
    {
        const int spp = cam->samples_per_pixel;
        int n = 0;
        tm.chunk_start[0] = 0;
        const int size = chunk_size_for(spp, opts->variant);
        for (int s0 = size; s0 < spp; s0 += size) tm.chunk_start[++n] = int16_t(s0);
        tm.chunk_start[++n] = int16_t(spp);
        tm.n_chunks = n;
        for (int k = n + 1; k <= kMaxChunks; k++) tm.chunk_start[k] = int16_t(spp);
    }
    const size_t elem = opts->real_mode == RTK_REAL_F64 ? sizeof(double) : sizeof(float);
    // Up to kMaxPlanesPerPass chunks per launch; a frame with more is rendered in consecutive passes, the resolve kernel
    // carrying the running sum in one extra plane -- the same additions in the same order as one pass over all chunks.
    const int n_chunks_total = tm.n_chunks;
    const size_t plane = size_t(tm.n_tiles_local) * 192 * elem;
    const int planes_per_pass = planes_per_pass_for(plane, opts->variant);
    const int n_passes = (n_chunks_total + planes_per_pass - 1) / planes_per_pass;
    const size_t need = plane * size_t(n_passes > 1 ? planes_per_pass + 1 : n_chunks_total);
    if (need > ctx->partial_bytes) {
        if (ctx->d_partial) {
            RTK_HIP(hipDeviceSynchronize());  // earlier launches may still read the old workspace
            RTK_HIP(hipFree(ctx->d_partial));
            ctx->d_partial = nullptr;
            ctx->partial_bytes = 0;
        }
        RTK_HIP(hipMalloc(&ctx->d_partial, need));
        ctx->partial_bytes = need;
    }
    hipStream_t stream = static_cast<hipStream_t>(opts->stream);
    auto* counters = reinterpret_cast<unsigned long long*>(d_counters);
    // tile order / cost buffers for this shape
    const int shape[5] = {cam->image_width, cam->image_height, opts->rank, opts->n_ranks, tm.n_tiles_local};
    if (tm.n_tiles_local > ctx->order_capacity) {
        if (ctx->d_tile_cost) {
            RTK_HIP(hipDeviceSynchronize());
            RTK_HIP(hipFree(ctx->d_tile_cost));
            RTK_HIP(hipFree(ctx->d_tile_order));
            ctx->d_tile_cost = nullptr;
            ctx->d_tile_order = nullptr;
        }
        RTK_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_tile_cost), size_t(tm.n_tiles_local) * sizeof(unsigned int)));
        RTK_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_tile_order), size_t(tm.n_tiles_local) * sizeof(int32_t)));
        ctx->order_capacity = tm.n_tiles_local;
        ctx->order_valid = false;
    }
    if (std::memcmp(shape, ctx->order_shape, sizeof shape) != 0) ctx->order_valid = false;
    const bool learn = (opts->variant & 4) == 0;  // variant bit 2: fixed row-major tile order (A/B, tests)
    if (!ctx->order_valid && learn) RTK_HIP(hipMemsetAsync(ctx->d_tile_cost, 0, size_t(tm.n_tiles_local) * sizeof(unsigned int), stream));
    const int32_t* tile_order = (ctx->order_valid && learn) ? ctx->d_tile_order : nullptr;
    unsigned int* tile_cost = learn ? ctx->d_tile_cost : nullptr;
    ctx->last_n_passes = n_passes;
    ctx->last_n_items = int64_t(tm.n_tiles_local) * n_chunks_total;
    const int cam_mode = opts->real_mode == RTK_REAL_F64 ? 0 : 1;
    const bool cam_cached = ctx->cam_valid[cam_mode] && std::memcmp(&ctx->cam_last[cam_mode], cam, sizeof(rtk_camera)) == 0;
    const unsigned int cslot = cam_cached ? ctx->cam_slot[cam_mode] : (ctx->next_camera++ % kCounterRing);
    if (!cam_cached) {
        // a slot about to be overwritten may still be the other mode's cached camera
        for (int m = 0; m < 2; m++)
            if (ctx->cam_valid[m] && ctx->cam_slot[m] == cslot) ctx->cam_valid[m] = false;
        ctx->cam_last[cam_mode] = *cam;
        ctx->cam_slot[cam_mode] = cslot;
        ctx->cam_valid[cam_mode] = true;
    }
    unsigned char* d_cam = ctx->d_cameras + cslot * kCameraStride;
    const bool allow_lds = (opts->variant & 1) == 0;  // variant bit 0: keep the program in global memory (A/B)
    const uint32_t diag = uint32_t(opts->variant) & 0xBFFF00u;  // bits 8..21, 23: scheduler policy / program layout A/B used by tools/ and tests only
    hipError_t e = hipSuccess;
    if (!cam_cached) {
        if (opts->real_mode == RTK_REAL_F64) {
            ctx->h_cameras64[cslot] = to_device_camera<double>(*cam);
            RTK_HIP(hipMemcpyAsync(d_cam, &ctx->h_cameras64[cslot], sizeof(CameraRec<double>), hipMemcpyHostToDevice, stream));
        } else {
            ctx->h_cameras32[cslot] = to_device_camera<float>(*cam);
            RTK_HIP(hipMemcpyAsync(d_cam, &ctx->h_cameras32[cslot], sizeof(CameraRec<float>), hipMemcpyHostToDevice, stream));
        }
    }
    void* const acc = n_passes > 1 ? static_cast<char*>(ctx->d_partial) + plane * size_t(planes_per_pass) : nullptr;
    unsigned int* pass_counter[kMaxPasses];
    for (int pass = 0; pass < n_passes; pass++) {  // every pass's work-item counter reads 0 from now on (progress reports add them up)
        pass_counter[pass] = ctx->tile_counters + (ctx->next_counter++ % kCounterRing);
        if (n_passes > 1) RTK_HIP(hipMemsetAsync(pass_counter[pass], 0, sizeof(unsigned int), stream));
    }
    for (int pass = 0; pass < n_passes && e == hipSuccess; pass++) {
        const int c0 = pass * planes_per_pass, c1 = std::min(n_chunks_total, c0 + planes_per_pass);
        TileMap tp = tm;  // this pass's chunks, numbered from 0: planes, work items and chunk boundaries are the pass's own
        tp.n_chunks = c1 - c0;
        for (int k = 0; k <= kMaxChunks; k++) tp.chunk_start[k] = tm.chunk_start[std::min(c0 + k, n_chunks_total)];
        unsigned int* tile_counter = pass_counter[pass];
        ctx->last_tile_counter[pass] = tile_counter;
        ctx->last_pass_items[pass] = int64_t(tp.n_tiles_local) * tp.n_chunks;
        unsigned int* pass_cost = pass == 0 ? tile_cost : nullptr;  // a tile's cost is measured on every pixel's first chunk
        const bool first = pass == 0, last = pass + 1 == n_passes;
        if (opts->real_mode == RTK_REAL_F64) {
            e = launch_render<double>(ctx->scene64.view, reinterpret_cast<const CameraRec<double>*>(d_cam), tp, opts->seed, ctx->features,
                                      opts->count_work != 0, allow_lds, diag, ctx->d_partial, counters, tile_counter, tile_order, pass_cost, stream);
            if (e == hipSuccess)
                e = launch_resolve<double>(ctx->d_partial, tp, cam->image_width, cam->image_height, cam->pixel_samples_scale, d_linear, d_rgb8, acc, first, last, stream);
        } else {
            e = launch_render<float>(ctx->scene32.view, reinterpret_cast<const CameraRec<float>*>(d_cam), tp, opts->seed, ctx->features,
                                     opts->count_work != 0, allow_lds, diag, ctx->d_partial, counters, tile_counter, tile_order, pass_cost, stream);
            if (e == hipSuccess)
                e = launch_resolve<float>(ctx->d_partial, tp, cam->image_width, cam->image_height, cam->pixel_samples_scale, d_linear, d_rgb8, acc, first, last, stream);
        }
    }
    if (e == hipSuccess && learn) {  // this frame's costs become the next frame's hand-out order
        e = launch_tile_order(ctx->d_tile_cost, tm.n_tiles_local, ctx->d_tile_order, stream);
        ctx->order_valid = e == hipSuccess;
        std::memcpy(ctx->order_shape, shape, sizeof shape);
    }
    if (e != hipSuccess) return fail(RTK_ERR_HIP, "render kernel launch failed: %s", hipGetErrorString(e));
    return RTK_OK;
}

int rtk_tiles_unpermute(rtk_ctx* ctx, int image_width, int image_height, int n_ranks, int real_mode, const void* d_gathered, void* d_linear, uint8_t* d_rgb8,
                        void* stream) {
    if (!ctx || !d_gathered || image_width <= 0 || image_height <= 0 || n_ranks < 1) return fail(RTK_ERR_INVALID, "rtk_tiles_unpermute: bad argument");
    RTK_HIP(hipSetDevice(ctx->device));
    const long long tpr = rtk_tiles_per_rank(image_width, image_height, n_ranks);
    hipError_t e = real_mode == RTK_REAL_F64
                       ? launch_unpermute<double>(d_gathered, image_width, image_height, n_ranks, tpr, d_linear, d_rgb8, static_cast<hipStream_t>(stream))
                       : launch_unpermute<float>(d_gathered, image_width, image_height, n_ranks, tpr, d_linear, d_rgb8, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(RTK_ERR_HIP, "unpermute kernel launch failed: %s", hipGetErrorString(e));
    return RTK_OK;
}

int rtk_render_host(rtk_ctx* ctx, const rtk_camera* cam, const rtk_render_opts* opts, double* h_linear, uint8_t* h_rgb8, rtk_work_counters* counters) {
    if (!ctx || !cam || !opts) return fail(RTK_ERR_INVALID, "rtk_render_host: null argument");
    if (opts->n_ranks != 1) return fail(RTK_ERR_INVALID, "rtk_render_host renders whole images (n_ranks must be 1)");
    RTK_HIP(hipSetDevice(ctx->device));
    const size_t n = size_t(cam->image_width) * cam->image_height * 3;
    const size_t elem = opts->real_mode == RTK_REAL_F64 ? sizeof(double) : sizeof(float);
    void* d_linear = nullptr;
    uint8_t* d_rgb8 = nullptr;
    rtk_work_counters* d_cnt = nullptr;
    int rc = RTK_OK;
    auto cleanup = [&]() {
        if (d_linear) (void)hipFree(d_linear);
        if (d_rgb8) (void)hipFree(d_rgb8);
        if (d_cnt) (void)hipFree(d_cnt);
    };
#define RTK_HIP_CLEAN(call)                                                                       \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            cleanup();                                                                            \
            return fail(RTK_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));               \
        }                                                                                         \
    } while (0)
    RTK_HIP_CLEAN(hipMalloc(&d_linear, n * elem));
    RTK_HIP_CLEAN(hipMalloc(reinterpret_cast<void**>(&d_rgb8), n));
    rtk_render_opts o = *opts;
    o.count_work = counters ? 1 : 0;
    if (counters) {
        RTK_HIP_CLEAN(hipMalloc(reinterpret_cast<void**>(&d_cnt), sizeof(rtk_work_counters)));
        RTK_HIP_CLEAN(hipMemsetAsync(d_cnt, 0, sizeof(rtk_work_counters), static_cast<hipStream_t>(o.stream)));
    }
    rc = rtk_render_device(ctx, cam, &o, d_linear, d_rgb8, d_cnt);
    if (rc != RTK_OK) {
        cleanup();
        return rc;
    }
    {
        hipStream_t st = static_cast<hipStream_t>(o.stream);
        rc = rtk::wait_with_progress(&ctx, &st, 1);
        if (rc != RTK_OK) {
            cleanup();
            return rc;
        }
    }
    if (h_linear) {
        if (opts->real_mode == RTK_REAL_F64) {
            RTK_HIP_CLEAN(hipMemcpy(h_linear, d_linear, n * sizeof(double), hipMemcpyDeviceToHost));
        } else {
            std::vector<float> tmp(n);
            RTK_HIP_CLEAN(hipMemcpy(tmp.data(), d_linear, n * sizeof(float), hipMemcpyDeviceToHost));
            for (size_t k = 0; k < n; k++) h_linear[k] = double(tmp[k]);
        }
    }
    if (h_rgb8) RTK_HIP_CLEAN(hipMemcpy(h_rgb8, d_rgb8, n, hipMemcpyDeviceToHost));
    if (counters) RTK_HIP_CLEAN(hipMemcpy(counters, d_cnt, sizeof(rtk_work_counters), hipMemcpyDeviceToHost));
    cleanup();
    return RTK_OK;
}

int rtk_debug_closest_hit(rtk_ctx* ctx, int real_mode, int n, const double* h_rays, const uint32_t* h_keys, double* h_out, uint64_t* h_draws) {
    if (!ctx || n < 0 || !h_rays || !h_keys || !h_out || !h_draws) return fail(RTK_ERR_INVALID, "rtk_debug_closest_hit: bad argument");
    if (!ctx->has_scene) return fail(RTK_ERR_NO_SCENE, "rtk_debug_closest_hit: no scene uploaded");
    if (n == 0) return RTK_OK;
    RTK_HIP(hipSetDevice(ctx->device));
    double *d_rays = nullptr, *d_out = nullptr;
    uint32_t* d_keys = nullptr;
    unsigned long long* d_draws = nullptr;
    auto cleanup = [&]() {
        if (d_rays) (void)hipFree(d_rays);
        if (d_out) (void)hipFree(d_out);
        if (d_keys) (void)hipFree(d_keys);
        if (d_draws) (void)hipFree(d_draws);
    };
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_rays), size_t(n) * 9 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_out), size_t(n) * 12 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_keys), size_t(n) * 3 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_draws), size_t(n) * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemcpy(d_rays, h_rays, size_t(n) * 9 * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_keys, h_keys, size_t(n) * 3 * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = real_mode == RTK_REAL_F64 ? launch_debug_hit<double>(ctx->scene64.view, n, d_rays, d_keys, d_out, d_draws, nullptr)
                                      : launch_debug_hit<float>(ctx->scene32.view, n, d_rays, d_keys, d_out, d_draws, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(h_out, d_out, size_t(n) * 12 * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(h_draws, d_draws, size_t(n) * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fail(RTK_ERR_HIP, "rtk_debug_closest_hit: %s", hipGetErrorString(e));
    return RTK_OK;
}

int rtk_debug_scatter(rtk_ctx* ctx, int real_mode, int n, const int32_t* h_materials, const double* h_rays, const double* h_records, const uint32_t* h_keys,
                      double* h_out, uint64_t* h_draws) {
    if (!ctx || n < 0 || !h_materials || !h_rays || !h_records || !h_keys || !h_out || !h_draws) return fail(RTK_ERR_INVALID, "rtk_debug_scatter: bad argument");
    if (!ctx->has_scene) return fail(RTK_ERR_NO_SCENE, "rtk_debug_scatter: no scene uploaded");
    if (real_mode != RTK_REAL_F64 && real_mode != RTK_REAL_F32) return fail(RTK_ERR_INVALID, "rtk_debug_scatter: unknown real_mode %d", real_mode);
    for (int k = 0; k < n; k++)
        if (h_materials[k] < 0 || h_materials[k] >= ctx->n_materials) return fail(RTK_ERR_INVALID, "rtk_debug_scatter: material %d of case %d out of range", h_materials[k], k);
    if (n == 0) return RTK_OK;
    RTK_HIP(hipSetDevice(ctx->device));
    KatBuffers b;
    const int32_t* d_mat = b.in(h_materials, size_t(n));
    const double* d_ray = b.in(h_rays, size_t(n) * 7);
    const double* d_rec = b.in(h_records, size_t(n) * 11);
    const uint32_t* d_keys = b.in(h_keys, size_t(n) * 3);
    double* d_out = b.out<double>(size_t(n) * 14);
    auto* d_draws = b.out<unsigned long long>(size_t(n));
    if (b.err == hipSuccess)
        b.err = real_mode == RTK_REAL_F64 ? launch_debug_scatter<double>(ctx->scene64.view, n, d_mat, d_ray, d_rec, d_keys, d_out, d_draws, nullptr)
                                          : launch_debug_scatter<float>(ctx->scene32.view, n, d_mat, d_ray, d_rec, d_keys, d_out, d_draws, nullptr);
    if (b.err == hipSuccess) b.err = hipDeviceSynchronize();
    b.back(h_out, d_out, size_t(n) * 14);
    b.back(reinterpret_cast<unsigned long long*>(h_draws), d_draws, size_t(n));
    if (b.err != hipSuccess) return fail(RTK_ERR_HIP, "rtk_debug_scatter: %s", hipGetErrorString(b.err));
    return RTK_OK;
}

int rtk_debug_texture(rtk_ctx* ctx, int real_mode, int n, const int32_t* h_textures, const double* h_uvp, double* h_out, uint64_t* h_work) {
    if (!ctx || n < 0 || !h_textures || !h_uvp || !h_out || !h_work) return fail(RTK_ERR_INVALID, "rtk_debug_texture: bad argument");
    if (!ctx->has_scene) return fail(RTK_ERR_NO_SCENE, "rtk_debug_texture: no scene uploaded");
    if (real_mode != RTK_REAL_F64 && real_mode != RTK_REAL_F32) return fail(RTK_ERR_INVALID, "rtk_debug_texture: unknown real_mode %d", real_mode);
    for (int k = 0; k < n; k++)
        if (h_textures[k] < 0 || h_textures[k] >= ctx->n_textures) return fail(RTK_ERR_INVALID, "rtk_debug_texture: texture %d of case %d out of range", h_textures[k], k);
    if (n == 0) return RTK_OK;
    RTK_HIP(hipSetDevice(ctx->device));
    KatBuffers b;
    const int32_t* d_tex = b.in(h_textures, size_t(n));
    const double* d_uvp = b.in(h_uvp, size_t(n) * 5);
    double* d_out = b.out<double>(size_t(n) * 3);
    auto* d_work = b.out<unsigned long long>(size_t(n) * 2);
    if (b.err == hipSuccess)
        b.err = real_mode == RTK_REAL_F64 ? launch_debug_texture<double>(ctx->scene64.view, n, d_tex, d_uvp, d_out, d_work, nullptr)
                                          : launch_debug_texture<float>(ctx->scene32.view, n, d_tex, d_uvp, d_out, d_work, nullptr);
    if (b.err == hipSuccess) b.err = hipDeviceSynchronize();
    b.back(h_out, d_out, size_t(n) * 3);
    b.back(reinterpret_cast<unsigned long long*>(h_work), d_work, size_t(n) * 2);
    if (b.err != hipSuccess) return fail(RTK_ERR_HIP, "rtk_debug_texture: %s", hipGetErrorString(b.err));
    return RTK_OK;
}

int rtk_debug_get_ray(rtk_ctx* ctx, int real_mode, const rtk_camera* cam, uint32_t seed, int n, const int32_t* h_pixel_sample, double* h_out, uint64_t* h_draws) {
    if (!ctx || !cam || n < 0 || !h_pixel_sample || !h_out || !h_draws) return fail(RTK_ERR_INVALID, "rtk_debug_get_ray: bad argument");
    if (real_mode != RTK_REAL_F64 && real_mode != RTK_REAL_F32) return fail(RTK_ERR_INVALID, "rtk_debug_get_ray: unknown real_mode %d", real_mode);
    if (cam->image_width <= 0 || cam->image_height <= 0) return fail(RTK_ERR_INVALID, "rtk_debug_get_ray: bad camera dimensions");
    if (n == 0) return RTK_OK;
    RTK_HIP(hipSetDevice(ctx->device));
    KatBuffers b;
    const int32_t* d_ijs = b.in(h_pixel_sample, size_t(n) * 3);
    double* d_out = b.out<double>(size_t(n) * 7);
    auto* d_draws = b.out<unsigned long long>(size_t(n));
    if (b.err == hipSuccess)
        b.err = real_mode == RTK_REAL_F64 ? launch_debug_get_ray<double>(to_device_camera<double>(*cam), seed, n, d_ijs, d_out, d_draws, nullptr)
                                          : launch_debug_get_ray<float>(to_device_camera<float>(*cam), seed, n, d_ijs, d_out, d_draws, nullptr);
    if (b.err == hipSuccess) b.err = hipDeviceSynchronize();
    b.back(h_out, d_out, size_t(n) * 7);
    b.back(reinterpret_cast<unsigned long long*>(h_draws), d_draws, size_t(n));
    if (b.err != hipSuccess) return fail(RTK_ERR_HIP, "rtk_debug_get_ray: %s", hipGetErrorString(b.err));
    return RTK_OK;
}

int rtk_frame_launches(const rtk_camera* cam, const rtk_render_opts* opts) {
    if (!cam || !opts || cam->samples_per_pixel <= 0 || cam->samples_per_pixel > 32767 || cam->image_width <= 0 || cam->image_height <= 0 || opts->n_ranks < 1)
        return fail(RTK_ERR_INVALID, "rtk_frame_launches: bad camera or options");
    const int size = chunk_size_for(cam->samples_per_pixel, opts->variant);
    const int chunks = (cam->samples_per_pixel + size - 1) / size;
    const size_t elem = opts->real_mode == RTK_REAL_F64 ? sizeof(double) : sizeof(float);
    const size_t plane = size_t(rtk_tiles_per_rank(cam->image_width, cam->image_height, opts->n_ranks)) * 192 * elem;
    const int per_pass = planes_per_pass_for(plane, opts->variant);
    return (chunks + per_pass - 1) / per_pass;
}

int rtk_scene_info(rtk_ctx* ctx, int32_t* n_program_ops, int64_t* bytes_f64, int64_t* bytes_f32) {
    if (!ctx || !ctx->has_scene) return fail(RTK_ERR_NO_SCENE, "rtk_scene_info: no scene uploaded");
    if (n_program_ops) *n_program_ops = ctx->scene64.view.n_slots;
    if (bytes_f64) *bytes_f64 = ctx->scene64.bytes;
    if (bytes_f32) *bytes_f32 = ctx->scene32.bytes;
    return RTK_OK;
}

const char* rtk_kernel_name(rtk_ctx* ctx, int real_mode, int variant) {
    if (!ctx || !ctx->has_scene) return "";
    const bool allow_lds = (variant & 1) == 0;
    const uint32_t diag = uint32_t(variant) & 0xBFFF00u;
    return real_mode == RTK_REAL_F64 ? render_kernel_name<double>(ctx->scene64.view, ctx->features, false, allow_lds, diag)
                                     : render_kernel_name<float>(ctx->scene32.view, ctx->features, false, allow_lds, diag);
}

}  // extern "C"

namespace rtk {

// What the hot part of `scene`'s COMPACT program takes in LDS, materials included (build_hot_cold_program; exact: the
// program is compiled).  For rtk_scene_optimize's choice of the primitive cost (rtk_optimize.cpp); 0 when the
// description does not compile.
size_t hot_program_lds_bytes(const rtk_scene_desc* scene) {
    if (!scene) return 0;
    try {
        Program prog;
        if (compile_scene(scene, prog) != RTK_OK) return 0;
        std::vector<Unit16> hot, cold;
        std::vector<uint32_t> ranks;
        build_hot_cold_program(*scene, prog, 0.0, hot, cold, ranks);
        return hot.size() * sizeof(Unit16) + size_t(scene->n_materials) * sizeof(MaterialRec<double>);
    } catch (const std::bad_alloc&) {
        return 0;
    }
}

}  // namespace rtk
