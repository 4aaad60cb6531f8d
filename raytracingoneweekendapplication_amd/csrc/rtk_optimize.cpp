// rtk_optimize.cpp -- the "fast order" scene optimiser behind rtk_scene_optimize
// (include/rtk.h).  Host code only: no device is touched, so it also runs where
// there is no GPU (the CPU tests execute the oracle on its output).
//
// the reference's bvh_node (bvh.h:13-45) splits the object
// list at the median of the longest axis and bvh_node::hit (bvh.h:64-72) always
// visits left then right.  On the book-1 scene that costs ~40 aabb::hit calls per
// ray.  This pass re-groups the SAME primitives -- nothing about a primitive, a
// material or an instance transform changes -- into a surface-area-heuristic
// hierarchy, and fixes the visiting order of every node's two children so that
// the child nearer to the eye comes first (a closest-hit found early shrinks the
// interval that every later slab test sees, exactly as `hit_left ? rec.t :
// ray_t.max` does in bvh.h:69).  The result is again an rtk_scene_desc made of
// RTK_NODE_BVH / RTK_NODE_LIST nodes over the same primitive tables, so the same
// linear traversal program, the same kernels and the same oracle run on it.
//
// What is preserved: the closest hit of every ray, hence every pixel bit for bit
// (the RNG stream of a sample does not depend on the visiting order unless a
// medium draws inside hit(), constant_medium.h:40 -- and media keep their place in
// the reference's order, see ordered_group; only with opts.free_media_order are they
// re-grouped too, parity then is statistical and rtk_optimize_info.exact says so: 0 = statistical, 1 = identical in every
// measurement but not provable (triangle scenes, below), 2 = proven bit-identical).
// What changes: the work counters (fewer box tests).  Triangles: triangle::hit scales
// its hit distance by a float reciprocal of a float determinant (triangle.h:72,77),
// so an accepted hit can lie ~1e-7 of the travelled distance outside the triangle's
// exact box; whether the reference's own boxes let such a hit through depends on
// its visiting order, which no other hierarchy can reproduce.  Triangle boxes are
// grown so that no accepted hit is ever culled here (the fast order finds the
// closest of ALL hits triangle::hit accepts; rtk_optimize_info.has_triangles
// flags such scenes): on the C4 mesh scene 1 sample in 5.3e8 took a different
// path before the boxes were grown.  Ties: two different primitives
// hit at exactly the same t are resolved by visiting order in the reference
// (strict `surrounds` for spheres, sphere.h:44-48; inclusive `contains` for
// quads/triangles, quad.h:39, triangle.h:91); every primitive node of the output
// carries its rank in the reference's order (rtk_node.c) and the kernels and the
// oracle resolve a tie by those ranks, i.e. the way the reference does.
//
// Boxes: every primitive box is computed the way the reference computes it
// (sphere.h:17,24-26; quad.h:21-25; triangle.h:59; hittable.h:43,75-98;
// constant_medium.h:55) and then grown by a relative margin of 2^-40 of the scene
// extent, so a slab test (aabb.h:61-85) in a re-grouped node can never cull a hit
// the reference's own boxes would have let through because of rounding.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <unordered_set>
#include <utility>
#include <vector>

#include "rtk.h"

namespace {

struct Box {
    double lo[3], hi[3];
    static Box empty() {
        const double inf = std::numeric_limits<double>::infinity();
        return Box{{inf, inf, inf}, {-inf, -inf, -inf}};
    }
    void grow(const Box& o) {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], o.lo[a]);
            hi[a] = std::max(hi[a], o.hi[a]);
        }
    }
    void grow(double x, double y, double z) {
        const double p[3] = {x, y, z};
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], p[a]);
            hi[a] = std::max(hi[a], p[a]);
        }
    }
    bool valid() const { return lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]; }
    double area() const {
        if (!valid()) return 0.0;
        const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return 2.0 * (dx * dy + dy * dz + dz * dx);
    }
    double centre(int a) const { return 0.5 * (lo[a] + hi[a]); }
    // squared distance from p to the box (0 inside)
    double distance2(const double p[3]) const {
        double d2 = 0;
        for (int a = 0; a < 3; a++) {
            const double d = p[a] < lo[a] ? lo[a] - p[a] : (p[a] > hi[a] ? p[a] - hi[a] : 0.0);
            d2 += d * d;
        }
        return d2;
    }
};

struct Item {
    int32_t node;  // index into the OUTPUT node table
    Box box;
    double cost;   // relative cost of testing it once
};

struct Holder {
    rtk_scene_desc desc;
    std::vector<rtk_node> nodes;
    std::vector<int32_t> children;
    std::vector<rtk_aabb> boxes;
};

struct Optimizer {
    const rtk_scene_desc& in;
    const rtk_optimize_opts& opts;
    Holder& out;
    std::vector<int32_t> memo;       // input node -> output node (a shared subtree stays shared)
    std::vector<Box> memo_box;       // box of that output node, in the parent's space
    std::vector<double> memo_cost;
    std::vector<char> on_stack;
    std::vector<char> memo_media;    // the subtree of that input node contains a constant_medium
    std::vector<int32_t> ref_rank;   // input primitive node -> 1 + rank of its first visit in the reference order (0: never visited)
    std::vector<char> holds_medium;  // input node -> 1: a constant_medium is reachable from it, 0: none (scan_media)
    bool keep_media_order = true;    // media keep their place in the reference's visiting order (see ordered_group)
    int32_t n_ordered = 0;
    bool failed = false;
    bool has_media = false;
    double margin = 0.0, tri_margin = 0.0;
    int split_depth = 0;
    bool has_triangles = false;
    int64_t n_bvh_in = 0;

    Optimizer(const rtk_scene_desc& i, const rtk_optimize_opts& o, Holder& h)
        : in(i), opts(o), out(h), memo(size_t(i.n_nodes), -1), memo_box(size_t(i.n_nodes)), memo_cost(size_t(i.n_nodes), 1.0), on_stack(size_t(i.n_nodes), 0), memo_media(size_t(i.n_nodes), 0) {}

    static constexpr double kBoxCost = 1.0;    // one slab test
    static constexpr double kChainCost = 3.0;  // entering and leaving an instance transform (two chain switches of the kernel)
    double prim_cost(int32_t kind) const { return opts.prim_cost_scale * base_prim_cost(kind); }
    static double base_prim_cost(int32_t kind) {  // measured ratios of the kernel's step costs, roughly
        switch (kind) {
            case RTK_NODE_SPHERE: return 2.1;   // A/B on the MI355X (tools/fast_order_probe.py, scale sweeps on C2 / C3 / C4):
            case RTK_NODE_QUAD: return 1.75;    // a sphere step (square root + two quotients) weighs about two slab tests,
            case RTK_NODE_TRIANGLE: return 1.75;  // quad and triangle steps somewhat less
            default: return 4.0;
        }
    }

    int32_t fail_node() {
        failed = true;
        return -1;
    }
    int32_t push_node(int32_t kind, int32_t a, int32_t b, int32_t c) {
        out.nodes.push_back(rtk_node{kind, a, b, c});
        return int32_t(out.nodes.size()) - 1;
    }

    // ---- boxes of the primitives, as the reference's constructors compute them
    static void pad_axis(double& lo, double& hi) {  // aabb.h:98-105 pad_to_minimums
        const double delta = 0.0001;
        if (hi - lo < delta) {
            const double padding = delta / 2;
            lo -= padding;
            hi += padding;
        }
    }
    bool primitive_box(const rtk_node& n, Box& b) {
        b = Box::empty();
        if (n.kind == RTK_NODE_SPHERE) {
            if (n.a < 0 || n.a >= in.n_spheres) return false;
            const rtk_sphere& s = in.spheres[n.a];
            const double r = s.radius;
            b.grow(s.center0.x - r, s.center0.y - r, s.center0.z - r);
            b.grow(s.center0.x + r, s.center0.y + r, s.center0.z + r);
            const double ex = s.center0.x + s.center_dir.x, ey = s.center0.y + s.center_dir.y, ez = s.center0.z + s.center_dir.z;
            b.grow(ex - r, ey - r, ez - r);
            b.grow(ex + r, ey + r, ez + r);
        } else if (n.kind == RTK_NODE_QUAD) {
            if (n.a < 0 || n.a >= in.n_quads) return false;
            const rtk_quad& q = in.quads[n.a];
            b.grow(q.Q.x, q.Q.y, q.Q.z);
            b.grow(q.Q.x + q.u.x, q.Q.y + q.u.y, q.Q.z + q.u.z);
            b.grow(q.Q.x + q.v.x, q.Q.y + q.v.y, q.Q.z + q.v.z);
            b.grow(q.Q.x + q.u.x + q.v.x, q.Q.y + q.u.y + q.v.y, q.Q.z + q.u.z + q.v.z);
        } else {
            if (n.a < 0 || n.a >= in.n_triangles) return false;
            const rtk_triangle& t = in.triangles[n.a];
            b.grow(t.p0.x, t.p0.y, t.p0.z);
            b.grow(t.p1.x, t.p1.y, t.p1.z);
            b.grow(t.p2.x, t.p2.y, t.p2.z);
        }
        for (int a = 0; a < 3; a++) pad_axis(b.lo[a], b.hi[a]);
        if (n.kind == RTK_NODE_TRIANGLE) {
            // triangle::hit scales u, v and t by a FLOAT 1/det of a float det (triangle.h:72,77): the accepted hit
            // distance is off by up to 2^-23 relative, so the hit point can lie outside the triangle's exact box
            // by that fraction of the distance the ray has travelled.  Grow by that, or a box would cull a hit
            // triangle::hit itself accepts.
            b = grown(b, tri_margin);
        }
        return true;
    }

    // Items of a group: everything reachable through nested LIST / BVH nodes without crossing an instance
    // transform or a medium.  Those become single items (optimised recursively, in their own space).
    bool gather(int32_t node, std::vector<Item>& items, std::unordered_set<int32_t>& present, int depth) {
        if (failed || node < 0 || node >= in.n_nodes || depth > 4096) return !(failed = true);
        const rtk_node& n = in.nodes[node];
        if (n.kind == RTK_NODE_LIST) {
            if (n.a < 0 || n.b < 0 || int64_t(n.a) + n.b > in.n_list_children) return !(failed = true);
            for (int32_t k = 0; k < n.b; k++)
                if (!gather(in.list_children[n.a + k], items, present, depth + 1)) return false;
            return true;
        }
        if (n.kind == RTK_NODE_BVH) {
            n_bvh_in++;
            // a span of one stores the object twice (bvh.h:30-32); the duplicate is sorted out per item below
            return gather(n.a, items, present, depth + 1) && gather(n.b, items, present, depth + 1);
        }
        Box b;
        double cost;
        const int32_t o = convert(node, b, cost, depth + 1);
        if (o < 0) return false;
        // The same object may be listed twice (a span of one: bvh.h:30-32).  Testing a surface twice finds nothing
        // new, so one test is kept; a constant_medium draws a fresh random number per test (constant_medium.h:40),
        // so its second test is a second, independent chance to scatter and stays.
        if (!present.insert(o).second && !memo_media[node]) return true;
        items.push_back(Item{o, b, cost});
        return true;
    }

    // Output node for a non-group input node (primitive, transform, medium); groups go through build_group.
    int32_t convert(int32_t node, Box& box, double& cost, int depth) {
        if (failed || node < 0 || node >= in.n_nodes || depth > 4096) {
            failed = true;
            return -1;
        }
        if (memo[node] >= 0) {
            box = memo_box[node];
            cost = memo_cost[node];
            if (memo_media[node]) has_media = true;
            return memo[node];
        }
        if (on_stack[node]) {  // a cycle
            failed = true;
            return -1;
        }
        on_stack[node] = 1;
        const rtk_node& n = in.nodes[node];
        int32_t o = -1;
        switch (n.kind) {
            case RTK_NODE_SPHERE:
            case RTK_NODE_QUAD:
            case RTK_NODE_TRIANGLE:
                if (n.kind == RTK_NODE_TRIANGLE) has_triangles = true;
                if (!primitive_box(n, box)) {
                    failed = true;
                    break;
                }
                cost = prim_cost(n.kind);
                o = push_node(n.kind, n.a, n.b, ref_rank[size_t(node)]);  // c = 1 + rank in the reference's visiting order (tie-break)
                break;
            case RTK_NODE_LIST:
            case RTK_NODE_BVH: {
                if (keep_media_order && holds_medium[size_t(node)]) {
                    o = ordered_group(node, box, cost, depth);
                    memo_media[node] = 1;
                    has_media = true;
                    break;
                }
                const bool media_before = has_media;
                has_media = false;
                o = build_group(node, box, cost, depth);
                memo_media[node] = has_media ? 1 : 0;  // some item of the group (recursively) is or contains a medium
                has_media = has_media || media_before;
                break;
            }
            case RTK_NODE_TRANSLATE: {
                if (n.a < 0 || n.a >= in.n_translates) {
                    failed = true;
                    break;
                }
                Box cb;
                double cc;
                const int32_t child = convert(n.b, cb, cc, depth + 1);
                if (child < 0) break;
                const rtk_vec3& off = in.translates[n.a].offset;  // hittable.h:43: bbox = object->bounding_box() + offset
                box = cb;
                box.lo[0] += off.x; box.hi[0] += off.x;
                box.lo[1] += off.y; box.hi[1] += off.y;
                box.lo[2] += off.z; box.hi[2] += off.z;
                cost = cc + kChainCost;
                o = push_node(RTK_NODE_TRANSLATE, n.a, child, 0);
                memo_media[node] = memo_media[n.b];
                break;
            }
            case RTK_NODE_ROTATE_Y: {
                if (n.a < 0 || n.a >= in.n_rotates) {
                    failed = true;
                    break;
                }
                Box cb;
                double cc;
                const int32_t child = convert(n.b, cb, cc, depth + 1);
                if (child < 0) break;
                const double s = in.rotates[n.a].sin_theta, c = in.rotates[n.a].cos_theta;
                box = Box::empty();  // hittable.h:75-98: the box of the eight rotated corners
                for (int i = 0; i < 2; i++)
                    for (int j = 0; j < 2; j++)
                        for (int k = 0; k < 2; k++) {
                            const double x = i ? cb.hi[0] : cb.lo[0], y = j ? cb.hi[1] : cb.lo[1], z = k ? cb.hi[2] : cb.lo[2];
                            box.grow(c * x + s * z, y, -s * x + c * z);
                        }
                cost = cc + kChainCost;
                o = push_node(RTK_NODE_ROTATE_Y, n.a, child, 0);
                memo_media[node] = memo_media[n.b];
                break;
            }
            case RTK_NODE_MEDIUM: {
                if (n.a < 0 || n.a >= in.n_media) {
                    failed = true;
                    break;
                }
                has_media = true;
                Box cb;
                double cc;
                const int32_t child = convert(n.b, cb, cc, depth + 1);
                if (child < 0) break;
                box = cb;  // constant_medium.h:55
                cost = 2.0 * cc + 3.0;
                o = push_node(RTK_NODE_MEDIUM, n.a, child, 0);
                memo_media[node] = 1;
                break;
            }
            default: failed = true; break;
        }
        on_stack[node] = 0;
        if (o < 0) {
            failed = true;
            return -1;
        }
        // Geometry the surface-area heuristic cannot price (a NaN, an infinity or a coordinate beyond 1e150 -- areas
        // overflow -- from a damaged description) is refused rather than sorted: RTK_ERR_INVALID.
        for (int a = 0; a < 3; a++) {
            const bool empty_axis = !(box.lo[a] <= box.hi[a]);
            const bool nan_axis = box.lo[a] != box.lo[a] || box.hi[a] != box.hi[a];
            if (nan_axis || (!empty_axis && (std::fabs(box.lo[a]) > 1e150 || std::fabs(box.hi[a]) > 1e150)) || !(cost == cost) || cost > 1e300) {
                failed = true;
                return -1;
            }
        }
        memo[node] = o;
        memo_box[node] = box;
        memo_cost[node] = cost;
        return o;
    }

    static Box grown(const Box& b, double m) {
        Box r = b;
        for (int a = 0; a < 3; a++) {
            r.lo[a] -= m;
            r.hi[a] += m;
        }
        return r;
    }

    int32_t emit_list(const std::vector<Item>& items, size_t begin, size_t end) {
        const int32_t first = int32_t(out.children.size());
        for (size_t k = begin; k < end; k++) out.children.push_back(items[k].node);
        return push_node(RTK_NODE_LIST, first, int32_t(end - begin), 0);
    }

    struct Built {
        int32_t node;
        double cost;   // expected cost given that the enclosing box was entered
        double first;  // ... of which this much is paid whatever the ray does next: the first test of every top-level member (one slab
                       // test for a boxed node, the whole test for a bare primitive) -- what a ray that misses the node still costs
    };
    // A boxed pair whose box a ray is likely to enter whenever it entered the enclosing one is not worth its slab test: a box() of
    // six quads, the walls of a room or a leaf of two neighbours span (nearly) the same box at every level of a binary hierarchy.
    // represent() prices the pair WITHOUT its box as well -- members handed to the enclosing node as a plain list -- and takes
    // the cheaper form: with the box C + P x inner, without it P x inner + (1 - P) x (first tests of the members); for two
    // boxed members that is "drop the box when P > 1/2".  RTK_OPT_FLATTEN=0 keeps every box (tools/: A/B).
    bool flatten = [] {
        const char* e = std::getenv("RTK_OPT_FLATTEN");
        return !(e && e[0] == '0');
    }();

    // One side of a split: the cheaper of "test the items one after the other" and "a box, then recurse".
    Built represent(std::vector<Item>& items, size_t begin, size_t end, const Box& parent) {
        if (end - begin == 1) {
            // An expensive item (an instance, a medium) whose own box is much smaller than the box it sits in gets a
            // slab test of its own: a bvh_node whose second child is an empty list.
            const Item& it = items[begin];
            const double pa = parent.area();
            const double boxed = kBoxCost + (pa > 0 ? std::min(1.0, it.box.area() / pa) : 1.0) * it.cost;
            if (boxed < it.cost) {
                const Box padded = grown(it.box, margin);
                out.boxes.push_back(rtk_aabb{padded.lo[0], padded.hi[0], padded.lo[1], padded.hi[1], padded.lo[2], padded.hi[2]});
                const int32_t nothing = push_node(RTK_NODE_LIST, int32_t(out.children.size()), 0, 0);
                return Built{push_node(RTK_NODE_BVH, it.node, nothing, int32_t(out.boxes.size()) - 1), boxed, kBoxCost};
            }
            return Built{it.node, it.cost, it.cost};
        }
        double linear = 0;
        Box b = Box::empty();
        for (size_t k = begin; k < end; k++) {
            linear += items[k].cost;
            b.grow(items[k].box);
        }
        const size_t nodes_mark = out.nodes.size(), child_mark = out.children.size(), box_mark = out.boxes.size();
        Built l{-1, 0, 0}, r{-1, 0, 0};
        Built inner = split(items, begin, end, b, &l, &r);
        const double pa = parent.area();
        const double pb = pa > 0 ? std::min(1.0, b.area() / pa) : 1.0;
        const double boxed = kBoxCost + pb * inner.cost;
        const double bare = flatten ? pb * inner.cost + (1.0 - pb) * (l.first + r.first) : std::numeric_limits<double>::infinity();
        if (int(end - begin) <= opts.max_leaf && linear <= boxed && linear <= bare) {
            out.nodes.resize(nodes_mark);  // discard the subtree just built (primitive nodes were created earlier)
            out.children.resize(child_mark);
            out.boxes.resize(box_mark);
            // nearest first inside a leaf as well
            order_items(items, begin, end);
            return Built{emit_list(items, begin, end), linear, linear};
        }
        if (bare < boxed && inner.node == int32_t(out.nodes.size()) - 1 && out.nodes.back().kind == RTK_NODE_BVH &&
            out.nodes.back().c == int32_t(out.boxes.size()) - 1) {
            // the pair's own node and box were pushed last: drop them, hand its two members (in their visiting order) to the caller
            const int32_t first_member = out.nodes.back().a, second_member = out.nodes.back().b;
            out.nodes.pop_back();
            out.boxes.pop_back();
            const int32_t at = int32_t(out.children.size());
            out.children.push_back(first_member);
            out.children.push_back(second_member);
            return Built{push_node(RTK_NODE_LIST, at, 2, 0), bare, l.first + r.first};
        }
        return Built{inner.node, boxed, kBoxCost};
    }

    void order_items(std::vector<Item>& items, size_t begin, size_t end) {
        if (!opts.has_eye) return;
        const double eye[3] = {opts.eye.x, opts.eye.y, opts.eye.z};
        std::stable_sort(items.begin() + long(begin), items.begin() + long(end),
                         [&](const Item& a, const Item& b) { return key(a.box, eye) < key(b.box, eye); });
    }
    static double key(const Box& b, const double eye[3]) {
        const double c[3] = {b.centre(0) - eye[0], b.centre(1) - eye[1], b.centre(2) - eye[2]};
        return b.distance2(eye) + 1e-6 * (c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    }

    // A BVH node over items[begin, end) (>= 2 items) whose box is `b`: full-sweep SAH on the three axes.
    Built split(std::vector<Item>& items, size_t begin, size_t end, const Box& b, Built* left_out = nullptr, Built* right_out = nullptr) {
        const size_t n = end - begin;
        int best_axis = -1;
        size_t best_k = 0;
        double best = std::numeric_limits<double>::infinity();
        std::vector<double> right_area(n), right_cost(n);
        for (int axis = 0; axis < 3; axis++) {
            std::sort(items.begin() + long(begin), items.begin() + long(end), [&](const Item& x, const Item& y) {
                const double cx = x.box.centre(axis), cy = y.box.centre(axis);
                return cx != cy ? cx < cy : x.node < y.node;
            });
            Box acc = Box::empty();
            double cost = 0;
            for (size_t k = n; k-- > 1;) {
                acc.grow(items[begin + k].box);
                cost += items[begin + k].cost;
                right_area[k] = acc.area();
                right_cost[k] = cost;
            }
            acc = Box::empty();
            cost = 0;
            for (size_t k = 1; k < n; k++) {  // left = [0, k), right = [k, n)
                acc.grow(items[begin + k - 1].box);
                cost += items[begin + k - 1].cost;
                const double c = acc.area() * cost + right_area[k] * right_cost[k];
                if (c < best) {
                    best = c;
                    best_axis = axis;
                    best_k = k;
                }
            }
        }
        // Degenerate inputs (thousands of coincident boxes) can make the heuristic peel off one item per level; past a
        // depth no sane hierarchy needs, split at the median instead so that recursion depth and work stay bounded.
        if (split_depth > 48 && (best_k == 1 || best_k + 1 == n)) best_k = n / 2;
        if (best_axis < 0) {  // no finite cost at all (cannot happen for the finite boxes convert() lets through): median of the last sort
            best_axis = 2;
            best_k = n / 2;
        }
        if (best_axis != 2)
            std::sort(items.begin() + long(begin), items.begin() + long(end), [&](const Item& x, const Item& y) {
                const double cx = x.box.centre(best_axis), cy = y.box.centre(best_axis);
                return cx != cy ? cx < cy : x.node < y.node;
            });
        const size_t mid = begin + best_k;
        split_depth++;
        Built l = represent(items, begin, mid, b);
        Built r = represent(items, mid, end, b);
        split_depth--;
        // visiting order of the two children (bvh.h:68-69 visits `left` first): nearer to the eye first; without
        // an eye, the side that is more likely to be hit
        bool swap = false;
        Box lb = Box::empty(), rb = Box::empty();
        for (size_t k = begin; k < mid; k++) lb.grow(items[k].box);
        for (size_t k = mid; k < end; k++) rb.grow(items[k].box);
        if (opts.has_eye) {
            const double eye[3] = {opts.eye.x, opts.eye.y, opts.eye.z};
            swap = key(rb, eye) < key(lb, eye);
        } else {
            swap = rb.area() > lb.area();
        }
        if (swap) std::swap(l, r);
        const Box padded = grown(b, margin);
        out.boxes.push_back(rtk_aabb{padded.lo[0], padded.hi[0], padded.lo[1], padded.hi[1], padded.lo[2], padded.hi[2]});
        const int32_t node = push_node(RTK_NODE_BVH, l.node, r.node, int32_t(out.boxes.size()) - 1);
        if (left_out) *left_out = l;
        if (right_out) *right_out = r;
        return Built{node, l.cost + r.cost, l.first + r.first};
    }

    int32_t build_group(int32_t node, Box& box, double& cost, int depth) {
        std::vector<Item> items;
        std::unordered_set<int32_t> present;
        if (!gather(node, items, present, depth)) return -1;
        return build_items(items, box, cost);
    }

    // A group that holds a constant_medium, directly or further down.  constant_medium::hit draws a random number
    // (constant_medium.h:40), so for the image to stay bit-identical a medium must be called with the very interval the
    // reference calls it with.  That interval is (0.001, closest hit so far), and "so far" means: among everything that
    // precedes the medium in the reference's visiting order -- hittable_list children in order, a bvh_node's left then
    // right (hittable_list.h:29-33, bvh.h:68-69).  The bvh_nodes above the medium do not matter beyond that order: when one
    // of their slab tests fails, the reference skips the medium, but a medium called in that situation returns false
    // BEFORE it draws -- its boundary lies inside that box, so either the ray misses the boundary (constant_medium.h:23,26)
    // or enters it beyond the closest hit so far / leaves it before 0.001, and `rec1.t >= rec2.t` ends the call
    // (constant_medium.h:31-34).  So the group is flattened into the reference's SEQUENCE: runs of medium-free objects,
    // each re-grouped freely (its closest hit within a given interval does not depend on the visiting order), separated by
    // the ordered items -- the media, and the instances that hold one -- in their reference order.  Every medium then
    // meets the same interval as in the reference, draws the same numbers, and the image stays bit-identical (checked
    // with boxes above the media grown by up to 1e6: same images, same draws; only the count of medium calls changes).
    struct Sequence {
        struct Entry {
            int32_t node;
            Box box;
            double cost;
            bool ordered;
        };
        std::vector<Entry> entries;
        std::vector<Item> run;
        std::unordered_set<int32_t> present;
    };
    bool close_run(Sequence& q) {
        if (q.run.empty()) return true;
        Box rb;
        double rc = 0;
        const int32_t g = build_items(q.run, rb, rc);
        if (g < 0) return false;
        q.entries.push_back({g, rb, rc, false});
        q.run.clear();
        q.present.clear();  // (an object listed again after a medium is tested again, as in the reference)
        return true;
    }
    bool walk_ordered(int32_t node, Sequence& q, int depth) {
        if (failed || node < 0 || node >= in.n_nodes || depth > 4096) return !(failed = true);
        if (!holds_medium[size_t(node)]) return gather(node, q.run, q.present, depth);
        const rtk_node n = in.nodes[node];
        if (n.kind == RTK_NODE_LIST) {
            if (n.a < 0 || n.b < 0 || int64_t(n.a) + n.b > in.n_list_children) return !(failed = true);
            for (int32_t k = 0; k < n.b; k++)
                if (!walk_ordered(in.list_children[n.a + k], q, depth + 1)) return false;
            return true;
        }
        if (n.kind == RTK_NODE_BVH) {  // a span of one lists its object twice (bvh.h:30-32): walked twice, as the reference visits it
            n_bvh_in++;
            return walk_ordered(n.a, q, depth + 1) && walk_ordered(n.b, q, depth + 1);
        }
        // a medium, or an instance transform with a medium inside: an ordered item
        if (!close_run(q)) return false;
        Box b;
        double c = 0;
        const int32_t o = convert(node, b, c, depth + 1);
        if (o < 0) return false;
        q.entries.push_back({o, b, c, true});
        n_ordered++;
        return true;
    }
    int32_t ordered_group(int32_t node, Box& box, double& cost, int depth) {
        Sequence q;
        if (!walk_ordered(node, q, depth) || !close_run(q)) return -1;
        box = Box::empty();
        for (const auto& e : q.entries) box.grow(e.box);
        cost = 0;
        std::vector<int32_t> members;
        for (const auto& e : q.entries) {
            int32_t member = e.node;
            double c = e.cost;
            // an ordered item much smaller than the group gets a slab test of its own (a box class step instead of the
            // item's far dearer one for every ray that passes it by), like any expensive single item (represent)
            const double pa = box.area();
            const double boxed = kBoxCost + (pa > 0 ? std::min(1.0, e.box.area() / pa) : 1.0) * e.cost;
            if (e.ordered && e.box.valid() && boxed < e.cost) {
                const Box padded = grown(e.box, margin);
                out.boxes.push_back(rtk_aabb{padded.lo[0], padded.hi[0], padded.lo[1], padded.hi[1], padded.lo[2], padded.hi[2]});
                const int32_t nothing = push_node(RTK_NODE_LIST, int32_t(out.children.size()), 0, 0);
                member = push_node(RTK_NODE_BVH, e.node, nothing, int32_t(out.boxes.size()) - 1);
                c = boxed;
            }
            members.push_back(member);
            cost += c;
        }
        const int32_t first = int32_t(out.children.size());
        out.children.insert(out.children.end(), members.begin(), members.end());
        return push_node(RTK_NODE_LIST, first, int32_t(members.size()), 0);
    }

    // Is a constant_medium reachable from each node?  (Iterative post-order; a cycle fails the pass.)
    bool scan_media(int32_t root) {
        holds_medium.assign(size_t(in.n_nodes), 0);
        std::vector<char> state(size_t(in.n_nodes), 0);  // 0 new, 1 open, 2 done
        std::vector<std::pair<int32_t, int32_t>> stack{{root, 0}};  // node, next child
        auto child_of = [&](const rtk_node& n, int32_t k) -> int32_t {  // -1: no such child, -2: malformed
            switch (n.kind) {
                case RTK_NODE_LIST:
                    if (n.a < 0 || n.b < 0 || int64_t(n.a) + n.b > in.n_list_children) return -2;
                    return k < n.b ? in.list_children[n.a + k] : -1;
                case RTK_NODE_BVH: return k == 0 ? n.a : (k == 1 ? n.b : -1);
                case RTK_NODE_TRANSLATE: case RTK_NODE_ROTATE_Y: case RTK_NODE_MEDIUM: return k == 0 ? n.b : -1;
                default: return -1;
            }
        };
        while (!stack.empty()) {
            auto& [node, next] = stack.back();
            if (node < 0 || node >= in.n_nodes || stack.size() > 4100) return false;
            const rtk_node& n = in.nodes[node];
            if (next == 0) {
                if (state[size_t(node)] == 2) { stack.pop_back(); continue; }
                if (state[size_t(node)] == 1) return false;
                state[size_t(node)] = 1;
                if (n.kind == RTK_NODE_MEDIUM) holds_medium[size_t(node)] = 1;
            }
            const int32_t c = child_of(n, next);
            if (c == -2) return false;
            if (c == -1) {
                state[size_t(node)] = 2;
                const int32_t done = node;
                stack.pop_back();
                if (!stack.empty() && holds_medium[size_t(done)]) holds_medium[size_t(stack.back().first)] = 1;
                continue;
            }
            next++;
            if (c < 0 || c >= in.n_nodes) return false;
            if (state[size_t(c)] == 2) {
                if (holds_medium[size_t(c)]) holds_medium[size_t(node)] = 1;
                continue;
            }
            stack.push_back({c, 0});
        }
        return true;
    }

    // The cheaper of a plain list and a SAH hierarchy over `items` (>= 0 of them), as one output node.
    int32_t build_items(std::vector<Item>& items, Box& box, double& cost) {
        if (items.empty()) {  // an empty list hits nothing (hittable_list.h:22-35)
            box = Box::empty();
            cost = 0;
            return push_node(RTK_NODE_LIST, int32_t(out.children.size()), 0, 0);
        }
        box = Box::empty();
        double linear = 0;
        for (const Item& it : items) {
            box.grow(it.box);
            linear += it.cost;
        }
        if (items.size() == 1) {
            cost = items[0].cost;
            return items[0].node;
        }
        const size_t nodes_mark = out.nodes.size(), child_mark = out.children.size(), box_mark = out.boxes.size();
        Built inner = split(items, 0, items.size(), box);
        if (int(items.size()) <= opts.max_leaf && linear <= kBoxCost + inner.cost) {
            out.nodes.resize(nodes_mark);
            out.children.resize(child_mark);
            out.boxes.resize(box_mark);
            order_items(items, 0, items.size());
            cost = linear;
            return emit_list(items, 0, items.size());
        }
        cost = kBoxCost + inner.cost;
        return inner.node;
    }

    // Scene extent for the rounding margin: boxes of all primitives in their own spaces plus the transforms'
    // offsets are bounded by the root box and the object-space boxes; use the largest coordinate seen.
    // The reference's visiting order of the INPUT graph: hittable_list children in order (hittable_list.h:29-33), a
    // bvh_node's left then right (bvh.h:68-69), a transform's child, a medium's boundary.  Iterative (graphs from
    // files may be deep); a node reached again keeps the rank of its first visit.
    void compute_ranks(int32_t root) {
        ref_rank.assign(size_t(in.n_nodes), 0);
        std::vector<char> seen(size_t(in.n_nodes), 0);
        std::vector<int32_t> stack{root};
        int32_t next = 1;
        size_t guard = 0;
        while (!stack.empty() && guard++ < (size_t(1) << 26)) {
            const int32_t node = stack.back();
            stack.pop_back();
            if (node < 0 || node >= in.n_nodes || seen[size_t(node)]) continue;
            seen[size_t(node)] = 1;
            const rtk_node& n = in.nodes[node];
            switch (n.kind) {
                case RTK_NODE_SPHERE: case RTK_NODE_QUAD: case RTK_NODE_TRIANGLE: ref_rank[size_t(node)] = next++; break;
                case RTK_NODE_LIST:
                    if (n.a >= 0 && n.b >= 0 && int64_t(n.a) + n.b <= in.n_list_children)
                        for (int32_t k = n.b; k-- > 0;) stack.push_back(in.list_children[n.a + k]);  // reversed: popped in list order
                    break;
                case RTK_NODE_BVH: stack.push_back(n.b); stack.push_back(n.a); break;
                case RTK_NODE_TRANSLATE: case RTK_NODE_ROTATE_Y: case RTK_NODE_MEDIUM: stack.push_back(n.b); break;
                default: break;
            }
        }
    }

    void compute_margin() {
        double extent = 0;
        for (int32_t i = 0; i < in.n_nodes; i++) {
            const rtk_node& n = in.nodes[i];
            if (n.kind == RTK_NODE_SPHERE || n.kind == RTK_NODE_QUAD || n.kind == RTK_NODE_TRIANGLE) {
                Box b;
                if (!primitive_box(n, b)) continue;
                for (int a = 0; a < 3; a++) extent = std::max(extent, std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])));
            }
        }
        for (int32_t i = 0; i < in.n_translates; i++)
            extent += std::fabs(in.translates[i].offset.x) + std::fabs(in.translates[i].offset.y) + std::fabs(in.translates[i].offset.z);
        if (opts.has_eye) extent = std::max(extent, std::max(std::fabs(opts.eye.x), std::max(std::fabs(opts.eye.y), std::fabs(opts.eye.z))));
        margin = std::ldexp(extent > 0 ? extent : 1.0, -40);
        tri_margin = std::ldexp(extent > 0 ? extent : 1.0, -20);  // 2^-23 relative x scene diameter (<= 2 sqrt(3) extent), doubled
    }
};

}  // namespace

namespace rtk {
size_t hot_program_lds_bytes(const rtk_scene_desc* scene);  // rtk_api.cpp
}

extern "C" {

static int optimize_once(const rtk_scene_desc* scene, const rtk_optimize_opts* opts_in, rtk_scene_desc** out_scene, rtk_optimize_info* info) {
    if (!scene || !out_scene) return RTK_ERR_INVALID;
    *out_scene = nullptr;
    if (scene->abi_version != RTK_ABI_VERSION || scene->n_nodes <= 0 || !scene->nodes || scene->root < 0 || scene->root >= scene->n_nodes) return RTK_ERR_INVALID;
    rtk_optimize_opts opts;
    std::memset(&opts, 0, sizeof opts);
    if (opts_in) opts = *opts_in;
    if (opts.max_leaf <= 0) opts.max_leaf = 4;
    if (!(opts.prim_cost_scale > 0)) opts.prim_cost_scale = 1.0;
    Holder* h = new (std::nothrow) Holder;
    if (!h) return RTK_ERR_INVALID;
    Optimizer op(*scene, opts, *h);
    op.keep_media_order = opts.free_media_order == 0;
    if (!op.scan_media(scene->root)) {
        delete h;
        return RTK_ERR_INVALID;
    }
    op.compute_ranks(scene->root);
    op.compute_margin();
    Box box;
    double cost = 0;
    int32_t root = op.convert(scene->root, box, cost, 0);
    if (root < 0 || op.failed) {
        delete h;
        return RTK_ERR_INVALID;
    }
    // camera::render is handed a hittable_list (main.cpp:442); keep the root a list so that a single primitive is valid too
    if (h->nodes[size_t(root)].kind != RTK_NODE_LIST && h->nodes[size_t(root)].kind != RTK_NODE_BVH) {
        const int32_t first = int32_t(h->children.size());
        h->children.push_back(root);
        root = op.push_node(RTK_NODE_LIST, first, 1, 0);
    }
    // The root box holds the whole scene.  Every secondary ray starts on a surface, i.e. inside it, and so does every
    // primary ray when the eye lies inside it: a slab test from the inside cannot fail (the interval starts at 0.001,
    // Camera.txt:211), so the root's test is dropped -- its two children are visited unconditionally, in order.
    if (h->nodes[size_t(root)].kind == RTK_NODE_BVH && opts.has_eye) {
        const rtk_node rn = h->nodes[size_t(root)];
        const rtk_aabb& rb = h->boxes[size_t(rn.c)];
        const bool inside = opts.eye.x > rb.xmin && opts.eye.x < rb.xmax && opts.eye.y > rb.ymin && opts.eye.y < rb.ymax && opts.eye.z > rb.zmin &&
                            opts.eye.z < rb.zmax;
        // ... and from an eye outside it, only primary rays can miss the root box -- one segment in (paths' length) -- and few of them
        // when the box fills the view: the test is dropped as well while the eye is nearer to the box than the box is wide (the
        // Cornell box seen through its open side: one slab test less for every segment, a tenth of that scene's).  A ray
        // that does miss then pays for the root's members' own boxes instead -- cost, never correctness.
        const double lo[3] = {rb.xmin, rb.ymin, rb.zmin}, hi[3] = {rb.xmax, rb.ymax, rb.zmax}, e[3] = {opts.eye.x, opts.eye.y, opts.eye.z};
        double dist2 = 0, diag2 = 0;
        for (int a = 0; a < 3; a++) {
            const double d = e[a] < lo[a] ? lo[a] - e[a] : (e[a] > hi[a] ? e[a] - hi[a] : 0.0);
            dist2 += d * d;
            diag2 += (hi[a] - lo[a]) * (hi[a] - lo[a]);
        }
        const bool near_by = op.flatten && dist2 < diag2;
        if (inside || near_by) {
            const int32_t first = int32_t(h->children.size());
            h->children.push_back(rn.a);
            h->children.push_back(rn.b);
            root = op.push_node(RTK_NODE_LIST, first, 2, 0);
        }
    }
    h->desc = *scene;  // primitive, material, texture, image, perlin and light tables are borrowed from the input
    h->desc.root = root;
    h->desc.n_nodes = int32_t(h->nodes.size());
    h->desc.nodes = h->nodes.data();
    h->desc.n_list_children = int32_t(h->children.size());
    h->desc.list_children = h->children.data();
    h->desc.n_bvh_boxes = int32_t(h->boxes.size());
    h->desc.bvh_boxes = h->boxes.data();
    if (info) {
        // Closest hits are preserved, exact ties are resolved by the reference's ranks (rtk_node.c) and a medium meets the
        // interval it meets in the reference (ordered_group), so the image is the reference order's bit for bit unless the
        // caller asked for the free order of media: exact = 2, PROVEN.  Triangles are the one caveat (header of this file,
        // rtk.h): triangle::hit scales t by a float reciprocal (triangle.h:72,77), so the REFERENCE's own exact boxes can
        // cull a hit its triangle::hit would accept, depending on its visiting order -- no other hierarchy can reproduce
        // that, and nothing here can rule it out for a given scene.  Identical in every measurement (C4: 0 of 5.3e8
        // samples), but that is a measurement: exact = 1, EMPIRICAL.  Callers that promise "the image never depends on
        // the order" (camera::auto_order) take the fast order on 2 only; bench.py takes it on 1 as well and verifies the
        // claim in the same run (both orders rendered, digests compared).
        info->exact = (op.has_media && !op.keep_media_order) ? 0 : (op.has_triangles ? 1 : 2);
        info->has_media = op.has_media ? 1 : 0;
        info->n_ordered_items = op.n_ordered;
        info->has_triangles = op.has_triangles ? 1 : 0;
        info->n_bvh_nodes_in = int32_t(op.n_bvh_in);
        info->n_bvh_nodes_out = int32_t(h->boxes.size());
        info->expected_cost = cost;
        info->box_margin = op.margin;
    }
    *out_scene = &h->desc;  // first member: rtk_scene_optimized_free casts the handle back to its holder
    return RTK_OK;
}

void rtk_scene_optimized_free(rtk_scene_desc* scene) { delete reinterpret_cast<Holder*>(scene); }

// Bytes of the COMPACT traversal program (csrc/rtk_device_layout.h: 16-byte units; box 32, sphere 48, moving sphere 80,
// triangle 80, quad 144, every other record 32) the upload would compile from `d`, materials included -- the quantity
// that decides whether the f64 kernels can keep the whole program in one CU's LDS (160 KB).  Mirrors the compiler's walk
// (rtk_api.cpp): a shared subtree counts once per use; a sphere-bounded medium is one 48-byte record.
// `mixed_layout`: the scene is one the upload gives the MIXED program instead (sphere-only scenes: 32-byte units -- a
// stationary sphere takes 64 bytes there, a moving one 96); sizing such a scene with the COMPACT figures accepted
// hierarchies of ~1 700-2 000 spheres whose MIXED program no longer fits LDS.
static size_t compact_program_bytes(const rtk_scene_desc& d, bool mixed_layout = false) {
    size_t bytes = 32 /* OP_END */ + size_t(d.n_materials) * 48;
    std::vector<std::pair<int32_t, int>> stack{{d.root, 0}};
    size_t guard = 0;
    while (!stack.empty() && guard++ < (size_t(1) << 24) && bytes < (size_t(1) << 30)) {
        const auto [node, depth] = stack.back();
        stack.pop_back();
        if (node < 0 || node >= d.n_nodes || depth > 4096) continue;
        const rtk_node& n = d.nodes[node];
        switch (n.kind) {
            case RTK_NODE_SPHERE: {
                const bool moving = n.a >= 0 && n.a < d.n_spheres && (d.spheres[n.a].center_dir.x != 0 || d.spheres[n.a].center_dir.y != 0 || d.spheres[n.a].center_dir.z != 0);
                bytes += mixed_layout ? (moving ? 96 : 64) : (moving ? 80 : 48);
                break;
            }
            case RTK_NODE_QUAD: bytes += 144; break;
            case RTK_NODE_TRIANGLE: bytes += 80; break;
            case RTK_NODE_LIST:
                if (n.a >= 0 && n.b >= 0 && int64_t(n.a) + n.b <= d.n_list_children)
                    for (int32_t k = 0; k < n.b; k++) stack.push_back({d.list_children[n.a + k], depth + 1});
                break;
            case RTK_NODE_BVH: bytes += 32; stack.push_back({n.a, depth + 1}); stack.push_back({n.b, depth + 1}); break;
            case RTK_NODE_TRANSLATE: case RTK_NODE_ROTATE_Y: bytes += 64; stack.push_back({n.b, depth + 1}); break;
            case RTK_NODE_MEDIUM: {
                const bool sphere_bound = n.b >= 0 && n.b < d.n_nodes && d.nodes[n.b].kind == RTK_NODE_SPHERE;
                if (sphere_bound) bytes += 48;
                else { bytes += 96; stack.push_back({n.b, depth + 1}); stack.push_back({n.b, depth + 1}); }
                break;
            }
            default: break;
        }
    }
    return bytes;
}

// Slots (64 B in f64) and box records of the slot program the upload compiles from `d` -- the same walk as above.  The f64
// kernels keep a slot program whole in LDS when it fits, otherwise its box records (56 B each) plus a nibble per slot and
// a rank word per 32 slots (rtk_api.cpp, F_LDS_BOXES).
static void slot_program_counts(const rtk_scene_desc& d, size_t& n_slots, size_t& n_boxes) {
    n_slots = 1;
    n_boxes = 0;
    std::vector<std::pair<int32_t, int>> stack{{d.root, 0}};
    size_t guard = 0;
    while (!stack.empty() && guard++ < (size_t(1) << 24) && n_slots < (size_t(1) << 26)) {
        const auto [node, depth] = stack.back();
        stack.pop_back();
        if (node < 0 || node >= d.n_nodes || depth > 4096) continue;
        const rtk_node& n = d.nodes[node];
        switch (n.kind) {
            case RTK_NODE_SPHERE: {
                const bool moving = n.a >= 0 && n.a < d.n_spheres && (d.spheres[n.a].center_dir.x != 0 || d.spheres[n.a].center_dir.y != 0 || d.spheres[n.a].center_dir.z != 0);
                n_slots += moving ? 2 : 1;
                break;
            }
            case RTK_NODE_QUAD: n_slots += 3; break;
            case RTK_NODE_TRIANGLE: n_slots += 2; break;
            case RTK_NODE_LIST:
                if (n.a >= 0 && n.b >= 0 && int64_t(n.a) + n.b <= d.n_list_children)
                    for (int32_t k = 0; k < n.b; k++) stack.push_back({d.list_children[n.a + k], depth + 1});
                break;
            case RTK_NODE_BVH: n_slots++; n_boxes++; stack.push_back({n.a, depth + 1}); stack.push_back({n.b, depth + 1}); break;
            case RTK_NODE_TRANSLATE: case RTK_NODE_ROTATE_Y: n_slots += 2; stack.push_back({n.b, depth + 1}); break;
            case RTK_NODE_MEDIUM: {
                const bool sphere_bound = n.b >= 0 && n.b < d.n_nodes && d.nodes[n.b].kind == RTK_NODE_SPHERE;
                if (sphere_bound) n_slots++;
                else { n_slots += 3; stack.push_back({n.b, depth + 1}); stack.push_back({n.b, depth + 1}); }
                break;
            }
            default: break;
        }
    }
}

// Will the upload compile the MIXED program for this scene (rtk_api.cpp upload_scene: want_mixed)?  Spheres only, no instance
// transforms, media or point lights, lambertian / metal / dielectric materials with solid colours.
static bool gets_mixed_program(const rtk_scene_desc& d) {
    if (d.n_quads > 0 || d.n_triangles > 0 || d.n_media > 0 || d.n_translates > 0 || d.n_rotates > 0 || d.n_lights > 0) return false;
    for (int32_t i = 0; i < d.n_materials; i++) {
        const rtk_material& m = d.materials[i];
        if (m.kind != RTK_MAT_LAMBERTIAN && m.kind != RTK_MAT_METAL && m.kind != RTK_MAT_DIELECTRIC) return false;
        if (m.kind == RTK_MAT_LAMBERTIAN && (m.texture < 0 || m.texture >= d.n_textures || d.textures[m.texture].kind != RTK_TEX_SOLID)) return false;
    }
    return true;
}

// With opts->prim_cost_scale left at 0 ("automatic") the price of a primitive test relative to a slab test is chosen by
// where the f64 kernels will find the traversal program (rtk_api.cpp / rtk_trace.hip):
//  * a scene of spheres and triangles is re-grouped with primitive tests priced 1.5x dearer -- more, tighter boxes and fewer
//    primitive tests, which is what pays once everything is read from LDS (C4: 66.0 -> 63.5 ms, C2: 21.31 -> 21.23) -- as
//    long as its COMPACT program still fits one CU's LDS: a program that has to leave LDS loses far more than the better
//    hierarchy gains (C4 at 1.6x: 175 ms), so the scale steps down (1.4, 1.2, 1.0) until it fits.  Scenes with quads start
//    at 2.0 (the Cornell box: 27.4 ms at 1.0 and 1.5, 26.75 at 1.75-2.0, 27.0 at 2.5-3, 27.3 from 4 on), then 1.0;
//  * a program too large for that (C5: 2 400 quads of 144 bytes) keeps its HOT part there -- boxes, spheres, everything
//    but the quads and triangles, which stay in memory (SceneView::program_hot) -- and primitive tests are priced DOWN
//    step by step (fewer, larger leaves: fewer boxes) until that part fits; failing that, until at least the box
//    records of the slot program do (the boxes-in-LDS kernels).
// An explicit scale is honoured as given.
int rtk_scene_optimize(const rtk_scene_desc* scene, const rtk_optimize_opts* opts_in, rtk_scene_desc** out_scene, rtk_optimize_info* info) {
    if (!scene || !out_scene) return RTK_ERR_INVALID;
    rtk_optimize_opts o;
    std::memset(&o, 0, sizeof o);
    if (opts_in) o = *opts_in;
    if (o.prim_cost_scale > 0) return optimize_once(scene, &o, out_scene, info);
    const size_t budget = size_t(160) * 1024 - 2048;  // the kernel's own LDS words
    static const double kScalesNoQuads[] = {1.5, 1.4, 1.2, 1.0, 0.85, 0.7, 0.6, 0.5, 0.4};
    static const double kScalesQuads[] = {2.0, 1.0, 0.85, 0.7, 0.6, 0.5, 0.4};
    const double* kScales = scene->n_quads == 0 ? kScalesNoQuads : kScalesQuads;
    const int n_scales = scene->n_quads == 0 ? int(sizeof kScalesNoQuads / sizeof kScalesNoQuads[0]) : int(sizeof kScalesQuads / sizeof kScalesQuads[0]);
    const bool full_feature = scene->n_media > 0 || scene->n_translates > 0 || scene->n_rotates > 0;  // only those kernels have the hot/cold form
    const bool mixed_layout = gets_mixed_program(*scene);
    rtk_scene_desc* fallback = nullptr;  // the first hierarchy whose slot program at least keeps its boxes in LDS
    rtk_optimize_info fallback_info;
    for (int k = 0; k < n_scales; k++) {
        o.prim_cost_scale = kScales[k];
        rtk_optimize_info local;
        rtk_scene_desc* candidate = nullptr;
        const int rc = optimize_once(scene, &o, &candidate, &local);
        if (rc != RTK_OK) {
            if (fallback) rtk_scene_optimized_free(fallback);
            return rc;
        }
        bool accept = compact_program_bytes(*candidate, mixed_layout) <= budget;  // sized in the layout the upload will use
        if (!accept && kScales[k] <= 1.0 && full_feature) {
            // the hot part (rtk_api.cpp compiles the program: exact), and the Perlin tables behind it: book-2's noise sphere
            // costs 4 % of the frame when perlin::turb gathers from memory instead
            const size_t hot = rtk::hot_program_lds_bytes(candidate);
            accept = hot > 0 && hot + size_t(scene->n_perlins) * 9216 <= budget;
        }
        if (accept) {
            if (fallback) rtk_scene_optimized_free(fallback);
            *out_scene = candidate;
            if (info) *info = local;
            return RTK_OK;
        }
        bool keep = false;
        if (!fallback && kScales[k] <= 1.0) {
            size_t n_slots = 0, n_boxes = 0;
            slot_program_counts(*candidate, n_slots, n_boxes);
            const size_t whole = n_slots * 64 + size_t(scene->n_materials) * 48;
            const size_t boxes_only = n_boxes * 56 + n_slots / 2 + n_slots / 4 + 64 + 2048;  // + a small material table
            keep = whole <= budget || boxes_only <= budget || k + 1 == n_scales;
        }
        if (keep || (!fallback && k + 1 == n_scales)) {
            fallback = candidate;
            fallback_info = local;
        } else {
            rtk_scene_optimized_free(candidate);
        }
        if (fallback && !full_feature) break;  // no hot/cold form to look for further down
    }
    if (!fallback) return RTK_ERR_INVALID;  // not reached: the last scale is always kept
    *out_scene = fallback;
    if (info) *info = fallback_info;
    return RTK_OK;
}

}  // extern "C"
