// rtk_device_layout.h -- how an uploaded scene lies in HBM, shared by the host
// side of librtk_hip.so (rtk_api.cpp builds and uploads it) and the kernels
// (rtk_trace.hip reads it).  Everything is templated on `real` (double for the
// parity mode, float for the throughput mode).
//
// The scene graph of rtk_scene_desc is compiled into a linear TRAVERSAL PROGRAM:
// one fused record ("slot": 8-byte header + the reals the visit needs, 64 B in f64,
// 32 B in f32) per visit, laid out in exactly the order the reference's
// recursion visits things (bvh.h:64-72: box test, then left subtree, then right
// subtree, both always), with a skip link on every box op.  Executing the
// program from op 0 to OP_END with "on a failed box test jump to op.aux" performs
// the same sequence of aabb::hit / primitive hit calls as the reference, with no
// traversal stack at all:
//
//   hittable_list          -> its children's programs, concatenated
//   bvh_node               -> OP_BOX(box, skip) . left . right          skip -> after right
//   sphere/quad/triangle   -> OP_SPHERE / OP_QUAD / OP_TRI (payload = record index, aux = chain)
//   translate / rotate_y   -> OP_CHAIN(child chain) . child . OP_CHAIN(parent chain)
//   constant_medium        -> OP_MED_BEGIN . boundary . OP_MED_MID(skip) . boundary . OP_MED_END
//
// A "chain" is the sequence of instance transforms from world space down to a
// primitive; the kernel keeps only the world-space ray and re-derives the
// object-space ray from it whenever the chain changes (bit-identical to the
// reference's nested ray construction, hittable.h:46-58,101-139, because the same
// operations are applied in the same order).  A shared subtree (DAG) is simply
// emitted once per use.
#ifndef RTK_DEVICE_LAYOUT_H
#define RTK_DEVICE_LAYOUT_H

#include <stdint.h>
#include <hip/hip_vector_types.h>

// f32 culling boxes (MIXED / COMPACT programs): 1 = the sign-selected slab test with a per-ray bracket of o/d, boxes grown by
// 2^-21 of their OWN coordinates; 0 = the min/max form, boxes grown by 2^-19 of the largest coordinate in the scene.  Shared
// by the kernels (rtk_trace.hip) and the program builders (rtk_api.cpp); A/B builds pass -DRTK_SIGNED_SLAB=0.
#ifndef RTK_SIGNED_SLAB
#define RTK_SIGNED_SLAB 1
#endif

// MIXED program (sphere-only scenes): 1 = a box record holds centre and half-extent, f[0..2] = c, f[3..5] = h, and the slab test
// is near = (c/d - o/d) - h/|d|, far = (c/d - o/d) + h/|d| -- nine fused multiply-adds (|1/d| is a free source modifier)
// instead of six selects and six multiply-adds; the half-extent carries the whole float error budget (rtk_api.cpp).
#ifndef RTK_CH_BOX
#define RTK_CH_BOX 1
#endif

// ... (RTK_CH_SCALED) with the ray parameter scaled per ray so that the current interval's end maps to 1: the two interval
// clamps of the slab test become the free [0, 1] output clamp of v_max3 / v_min3 (rtk_trace.hip slab_test32_chs)
#ifndef RTK_CH_SCALED
#define RTK_CH_SCALED 1
#endif
// ... and (RTK_CH_BYTE_PC) that kernel's program counters and box links count bytes instead of 32-byte units
#ifndef RTK_CH_BYTE_PC
#define RTK_CH_BYTE_PC 1
#endif

// COMPACT program: 1 = centre / half-extent records for the programs of the mesh family (triangles + spheres) as well; the
// origin's share of the float error stays with the RAY, as one slack term for all three axes (rtk_trace.hip
// slab_test32_che), so a box is grown by 2^-21 of its own coordinates only -- nine fused multiply-adds, a subtraction and a
// compare against -2 x slack.  The other families keep the sign-selected test (measured, see rtk_trace.hip).
#ifndef RTK_CH_COMPACT
#define RTK_CH_COMPACT 1
#endif

namespace rtk {

enum OpKind : uint32_t {
    OP_END = 0,
    OP_BOX = 1,        // 1 slot : v = xmin,xmax,ymin,ymax,zmin,zmax; aux = pc to continue at when the slab test fails
    // primitive records: aux = chain id (bits 0..7) | material index (bits 8..31)
    OP_SPHERE = 2,     // 1 slot : v = cx,cy,cz,radius,1/radius (static sphere); payload = sphere index
    OP_QUAD = 3,       // 3 slots: n(3),D,Q(3),w(3),v(3),u(3) packed over the slots' v[]; payload = quad index
    OP_TRI = 4,        // 2 slots: e2(3),e1(3),p0(3); payload = triangle index
    OP_CHAIN = 5,      // 1 slot : payload = chain id to make current, aux = number of transform entries it stands for
    OP_MED_BEGIN = 6,  // 1 slot : payload = medium index
    OP_MED_MID = 7,    // 1 slot : payload = medium index, aux = pc after the matching OP_MED_END
    OP_MED_END = 8,    // 1 slot : v[0] = neg_inv_density; payload = medium index, aux = chain | material << 8
    OP_SPHERE_MOVING = 9,  // 2 slots: as OP_SPHERE, then v = dx,dy,dz (center2 - center1)
    // constant_medium::hit whose boundary is one stationary sphere (both media of the book-2 scene: the fog inside the glass
    // ball and the global fog of radius 5000, main.cpp:305-309), as ONE record instead of the five-step bracket
    // OP_MED_BEGIN . OP_SPHERE . OP_MED_MID . OP_SPHERE . OP_MED_END: the same two sphere::hit calls with the same
    // intervals, the same clamping, the same single random_double() (constant_medium.h:20-53) -- four scheduler rounds
    // fewer per medium test, and no query state to park meanwhile.
    OP_MED_SPHERE = 10,    // 1 slot : v = cx,cy,cz,radius,neg_inv_density; payload = medium index, aux = chain | material << 8
    OP_DEAD = 15           // never in a program: the kernel's marker for a lane that owns no pixel
};

struct Op {
    uint32_t kind_payload;  // kind in bits 0..3, payload in bits 4..31
    uint32_t aux;
};
inline constexpr uint32_t make_op(uint32_t kind, uint32_t payload) { return kind | (payload << 4); }

// The fused program record.  pc counts slots; a box visit reads exactly one.
template <typename real>
struct alignas(16) Slot {
    static constexpr int kReals = sizeof(real) == 8 ? 7 : 6;
    real v[kReals];              // first: a box's six bounds are then whole 16-byte LDS reads (ds_read_b128)
    uint32_t kind_payload, aux;  // header last: f64 at byte 56, f32 at byte 24
};
static_assert(sizeof(Slot<double>) == 64 && sizeof(Slot<float>) == 32, "slot size");

// A box slot without its header word (F_LDS_BOXES keeps only these in LDS): 56 bytes in f64, so that the 2 570 boxes of
// the book-2 scene fit one CU's LDS beside the kind and rank tables (64-byte slots would not).
template <typename real>
struct alignas(8) BoxRec {
    real v[6];
    uint32_t aux, unused;
};
static_assert(sizeof(BoxRec<double>) == 56 && sizeof(BoxRec<float>) == 32, "box record size");

template <typename real>
inline constexpr int slots_of(uint32_t kind) {
    return kind == OP_QUAD ? 3 : ((kind == OP_TRI || kind == OP_SPHERE_MOVING) ? 2 : 1);
}

// The MIXED traversal program (F_F32_BOX): records are sequences of 32-byte units, pc counts units.  The first unit of
// every record carries the header at bytes 24..31, so the kind of the record at pc is always one 8-byte read away:
//   OP_BOX            1 unit : f[0..5] = xmin,xmax,ymin,ymax,zmin,zmax as floats, rounded OUTWARD and grown (see
//                              rtk_api.cpp) -- or, RTK_CH_BOX, centre(3) and half-extent(3), the half-extent rounded up
//                              and grown; aux = pc to continue at when the slab test fails
//   OP_SPHERE         2 units: d[0..2] = centre (f64); unit 1 = radius, 1/radius (f64)
//   OP_SPHERE_MOVING  3 units: as OP_SPHERE; unit 1 also holds dx, dy; unit 2 holds dz (centre2 - centre1)
//   OP_END            1 unit
// A culling box only decides which primitives get tested; the tests themselves, and so the hit, stay exact f64.
struct alignas(16) MixedHead {
    // a box: six floats xmin, xmax, ymin, ymax, zmin, zmax; any other record: its first three payload doubles.  Plain words
    // with bit-cast accessors rather than a union: a union read under both types keeps a stack copy of the record alive in
    // the kernels (six dead scratch stores around every box loop in the ISA of the lean MIXED kernel).
    uint32_t w[6];
    uint32_t kind_payload, aux;
    constexpr float f(int k) const { return __builtin_bit_cast(float, w[k]); }
    constexpr double d(int k) const { return __builtin_bit_cast(double, (unsigned long long)(w[2 * k]) | ((unsigned long long)(w[2 * k + 1]) << 32)); }
    void set_f(int k, float x) { w[k] = __builtin_bit_cast(uint32_t, x); }
    void set_d(int k, double x) {
        const unsigned long long v = __builtin_bit_cast(unsigned long long, x);
        w[2 * k] = uint32_t(v);
        w[2 * k + 1] = uint32_t(v >> 32);
    }
};
static_assert(sizeof(MixedHead) == 32, "mixed unit size");
inline constexpr int mixed_units(uint32_t kind) { return kind == OP_SPHERE ? 2 : (kind == OP_SPHERE_MOVING ? 3 : 1); }

// The COMPACT traversal program: the MIXED idea for EVERY scene of the fast order (f64 kernels) -- f32 culling boxes, exact
// f64 primitives -- with records packed in 16-byte units (pc counts units) so that a triangle costs 80 bytes instead of
// 128 and the C4 mesh program (1 334 boxes + 1 280 triangles: 151 KB) fits one CU's LDS whole.  Every record starts
// with a MixedHead (24 payload bytes, then {kind_payload, aux} at bytes 24..31); further payload follows as doubles:
// element e of a record's payload is head.d[e] for e < 3 and the double at byte 32 + 8 (e - 3) otherwise.
//   OP_BOX            2 units: f[0..5] as in the MIXED program; aux = pc to continue at when the slab test fails
//   OP_SPHERE         3 units: centre(3), radius, 1/radius
//   OP_SPHERE_MOVING  5 units: centre(3), radius, 1/radius, centre2 - centre1 (3)
//   OP_QUAD           9 units: n(3), D, Q(3), w(3), v(3), u(3)
//   OP_TRI            5 units: e2(3), e1(3), p0(3)
//   OP_MED_SPHERE     3 units: centre(3), radius, neg_inv_density
//   OP_CHAIN, OP_MED_BEGIN, OP_MED_MID, OP_MED_END (element 0 = neg_inv_density), OP_END: 2 units
struct alignas(16) Unit16 {
    uint32_t w[4];
};
inline constexpr int compact_units(uint32_t kind) {
    return (kind == OP_SPHERE || kind == OP_MED_SPHERE) ? 3 : (kind == OP_SPHERE_MOVING ? 5 : (kind == OP_QUAD ? 9 : (kind == OP_TRI ? 5 : 2)));
}

constexpr uint32_t kNoHit = 0xFFFFFFFFu;
constexpr int kMaxChain = 4;
constexpr int kMaxChunks = 64;
constexpr int kLdsBytesPerCU = 160 * 1024;  // gfx950

// Scene feature bits: which op kinds / shading paths a scene needs.  The host
// picks the leanest kernel instantiation that covers the scene's mask.
enum Feature : uint32_t {
    F_QUAD = 1u << 0,
    F_TRI = 1u << 1,
    F_XFORM = 1u << 2,
    F_MEDIA = 1u << 3,
    F_TEXTURE = 1u << 4,  // any non-solid texture (checker, checker_tri, image, noise)
    F_LIGHTS = 1u << 5,
    F_EXOTIC_MAT = 1u << 6,  // isotropic / specular / diffuse_light
    // Not a scene feature but a property of the uploaded hierarchy: its boxes carry rtk_scene_optimize's margin, so the
    // slab test may use the fused form (rtk_scene_upload_fast).  Orthogonal to the bits above.
    F_FMA_BOX = 1u << 7,
    // f64 kernels only, sphere-only scenes: the MIXED program below -- conservative f32 culling boxes in 32-byte
    // records, exact f64 primitive tests (rtk_scene_upload_fast picks it when the scene qualifies).
    F_F32_BOX = 1u << 8,
    // A restriction, not a feature: every material is a lambertian or a light (the Cornell box).  The quad/box subset
    // kernel then carries no metal / dielectric / isotropic / specular code at all.
    F_MATTE = 1u << 9,
    // Not a scene feature either: the program is too large for LDS but its box records are not -- the kernel stages the
    // boxes alone (SceneView::box_cache), the per-slot kind nibbles and a rank table in LDS; primitives stay in HBM/L2.
    F_LDS_BOXES = 1u << 10,
    // A restriction like F_MATTE: every constant_medium of the scene is bounded by one stationary sphere (OP_MED_SPHERE records
    // only: book 2).  The full-feature kernel then carries neither the OP_MED_BEGIN / MID / END steps nor the parked outer
    // query they need (sv_tmin, sv_best_t, rec1_t, sv_best_pc): 168 VGPRs + 96 B of scratch instead of + 160 B at three waves.
    F_SPHERE_MEDIA_ONLY = 1u << 11
};
constexpr uint32_t kFeatLean = 0;                       // spheres + lambertian/metal/dielectric with solid colours
constexpr uint32_t kFeatAll = 0x7F;
// Two common subsets get their own instantiation (fewer live registers than the full kernel):
constexpr uint32_t kFeatQuadBox = F_QUAD | F_XFORM | F_EXOTIC_MAT;  // quads, box() instances, area lights (the Cornell box)
constexpr uint32_t kFeatMesh = F_TRI | F_EXOTIC_MAT;                // triangle meshes + spheres + emissive spheres

template <typename real>
struct alignas(16) TriRec {  // triangle.h:124-143; e1 = p1-p0, e2 = p2-p0 are what hit() recomputes per call
    real p0[3], e1[3], e2[3], n[3];
    float uv0[2], uv1[2], uv2[2];
    int32_t material, _pad;
};

template <typename real>
struct alignas(16) MaterialRec {  // material.h
    real albedo[3], param;
    int32_t kind, tex;
    int32_t needs_uv, _pad;       // texture (transitively) reads rec.u/rec.v
};

template <typename real>
struct alignas(16) TextureRec {  // texture.h
    real color[3], param;
    int32_t kind, even, odd, image;
};

struct ImageRec {  // rtw_stb_image.h:84-90
    int32_t width, height;
    int64_t texel_offset;
};

template <typename real>
struct PerlinRec {  // perlin.h:52-57
    real randvec[256][3];
    int32_t perm_x[256], perm_y[256], perm_z[256];
};

template <typename real>
struct ChainRec {  // instance transforms from world space to the primitive, outermost first
    int32_t count;
    int32_t is_rotate[kMaxChain];
    real a[kMaxChain], b[kMaxChain], c[kMaxChain];  // translate: offset xyz; rotate_y: a = sin, b = cos
};

template <typename real>
struct LightRec {  // point_light.h:24-27
    real position[3], intensity[3], size;
};

template <typename real>
struct CameraRec {  // Camera.txt:122-131
    real background[3], center[3], pixel00[3], du[3], dv[3], disk_u[3], disk_v[3];
    real defocus_angle, samples_scale;
    int32_t width, height, spp, max_depth;
};

template <typename real>
struct SceneView {  // device pointers, passed to the kernel by value
    const Slot<real>* program;  // the traversal program (copied to LDS by each workgroup when it fits)
    const TriRec<real>* tris;
    const MaterialRec<real>* materials;
    const TextureRec<real>* textures;
    const ImageRec* images;
    const uint8_t* texels;
    const PerlinRec<real>* perlins;
    const ChainRec<real>* chains;
    const LightRec<real>* lights;
    int32_t n_slots, n_lights, n_materials;
    // MIXED program (F_F32_BOX kernels; null otherwise) and the coordinate bound its box margin was sized for: a ray
    // whose origin lies outside [-extent, extent]^3 takes the exact f64 slab test instead of the f32 one
    const MixedHead* program_mixed;
    int32_t n_units;
    float extent;
    // COMPACT program (F_F32_BOX kernels of the other families; null when the scene has a MIXED program or was uploaded in
    // the reference order) -- same `extent` rule, applied to the object-space origin under instance transforms
    const Unit16* program_compact;
    int32_t n_units16;
    int32_t compact_ch;  // 1 = program_compact's box records are centre / half-extent (RTK_CH_COMPACT: the mesh family's programs)
    // ... and, for COMPACT programs too large for LDS, the same program in two parts (F_LDS_BOXES | F_F32_BOX kernels): the
    // hot part -- staged in LDS -- holds every record except quads and triangles, with one two-unit record {kind, count;
    // aux = first unit in program_cold} per run of them; program_cold holds those primitives' usual records.  A hit on a
    // cold primitive is named n_hot_units + its unit; tie_rank_hot is indexed the same way.  Null when not built.
    const Unit16* program_hot;
    const Unit16* program_cold;
    int32_t n_hot_units, n_cold_units;
    const uint32_t* tie_rank_hot;
    // Fast-order uploads: rank of the primitive record at pc in the REFERENCE's visiting order (rtk_node.c), indexed by the
    // pc of the f32-box program (MIXED units or COMPACT units) resp. of the slot program; 0 = unknown.  Only read when two
    // primitives are hit at exactly the same distance, to resolve the tie the way the reference's left-to-right
    // traversal does (strict `surrounds` for spheres, sphere.h:44-48; inclusive `contains` for quads and triangles,
    // quad.h:39, triangle.h:91).  Null in reference-order uploads: there the visiting order itself is the reference's.
    const uint32_t* tie_rank;
    const uint32_t* tie_rank_slot;
    // F_LDS_BOXES (programs larger than LDS; null when the program fits or the boxes do not): the OP_BOX slots in program
    // order; kind_words[pc >> 3] holds the kind of slot pc in nibble (pc & 7); box_rank[pc >> 5] = {bit per slot that
    // starts a box, number of boxes before slot 32 * (pc >> 5)} -- cache index of the box at pc = y + popc(x & below(pc)).
    const BoxRec<real>* box_cache;
    const uint32_t* kind_words;
    const uint2* box_rank;
    int32_t n_cached_boxes, n_kind_words, n_rank_words;
    int32_t n_perlins, n_chains;
};

struct TileMap {  // which tiles this launch renders and where the pixels go
    int32_t tiles_x, tiles_y, n_tiles_local;
    int32_t rank, n_ranks;
    int32_t compact;  // resolve output: 1 = this rank's [local_tile][3][64] buffer, 0 = the row-major image
    // Work items pulled by the persistent waves from an atomic counter: item = local_tile * n_chunks + chunk,
    // chunk c = samples [chunk_start[c], chunk_start[c+1]) of each of the tile's 64 pixels (all chunks of a tile
    // are adjacent items, so a heavy tile is spread over many waves at once).  The boundaries depend on spp only,
    // never on the number of ranks, so a pixel's partial sums -- and the image -- are the same on any GPU count.
    // The render kernel writes one partial sum per (chunk, tile, pixel) into partial[chunk][local_tile][3][64];
    // the resolve kernel adds a pixel's chunks in index order.
    int32_t n_chunks;
    int16_t chunk_start[kMaxChunks + 1];
    // set by the launcher: the tile hand-out order is staged in LDS at this byte offset of the dynamic segment
    int32_t order_in_lds, order_lds_offset;
    // F_LDS_BOXES kernels: byte offset of the material table staged behind the box tables (0 = materials stay in memory)
    int32_t mats_lds_offset;
    // ... and of the Perlin tables (9 KB each in f64: 256 gradient vectors + three permutations) when they fit as well
    // (0 = they stay in memory): perlin::turb reads 7 x (6 permutation entries + 8 gradients) through a dependent index
    int32_t perlin_lds_offset;
    // every kernel that stages anything: byte offset of the instance-transform chains (0 = they stay in memory).  apply_chain
    // walks a record through data-dependent branches -- count, is_rotate[k], then that step's constants: up to six
    // DEPENDENT reads per chain switch, each an L2 round trip from memory
    int32_t chains_lds_offset;
};

// Indices into the uint64 work-counter block (same order as rtk_work_counters).
enum CounterSlot {
    C_SAMPLES, C_SEGMENTS, C_BOX, C_SPHERE, C_QUAD, C_TRI, C_XFORM, C_MEDIUM, C_SURFACE, C_NOISE, C_TEXEL, C_RNG, C_COUNT
};

}  // namespace rtk

#endif  // RTK_DEVICE_LAYOUT_H
