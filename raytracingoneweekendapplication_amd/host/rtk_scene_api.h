// rtk_scene_api.h -- the scene-graph half of the drop-in API.
//
// Class names, constructors and public members follow the reference so that
// main.cpp's scene code (main.cpp:128-442) is source compatible:
//   texture.h   solid_color, checker_texture, checker_texture_triangle, image_texture, noise_texture
//   material.h  lambertian, metal, dielectric, diffuse_light, emissive_light, isotropic, specular
//   hittable.h  hittable, translate, rotate_y        hittable_list.h  hittable_list
//   bvh.h       bvh_node       sphere.h sphere       quad.h quad, box()
//   triangle.h  triangle, triangle_quad()            constant_medium.h constant_medium
//   perlin.h    perlin         rtw_stb_image.h rtw_image     point_light.h point_light
//
// What is different by design: the objects are *descriptions*.  None of them
// can intersect a ray or scatter on the host -- hittable::hit / material::scatter
// of the reference exist only as device code (csrc/rtk_trace.hip).  Each object
// knows how to append itself to an rtk::scene_builder, which produces the flat,
// index-linked rtk_scene_desc of include/rtk.h.
#ifndef RTK_SCENE_API_H
#define RTK_SCENE_API_H

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <new>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "glm_min.h"  // glm::vec2 for triangle UVs (triangle.h:137-139); the full GLM is not a dependency
#include "rtk.h"
#include "rtk_jpeg.h"
#include "rtk_png.h"
#include "rtk_math.h"

namespace rtk {

inline rtk_vec3 to_abi(const vec3& v) { return rtk_vec3{v.x(), v.y(), v.z()}; }

// Collects the tables of rtk_scene_desc.  Objects are memoised by address so
// that a shared_ptr used twice becomes one node referenced twice.
class scene_builder {
public:
    std::vector<rtk_node> nodes;
    std::vector<int32_t> list_children;
    std::vector<rtk_sphere> spheres;
    std::vector<rtk_quad> quads;
    std::vector<rtk_triangle> triangles;
    std::vector<rtk_aabb> bvh_boxes;
    std::vector<rtk_translate> translates;
    std::vector<rtk_rotate_y> rotates;
    std::vector<rtk_medium> media;
    std::vector<rtk_material> materials;
    std::vector<rtk_texture> textures;
    std::vector<rtk_image> images;
    std::vector<uint8_t> texels;
    std::vector<rtk_perlin> perlins;
    std::vector<rtk_point_light> lights;
    int32_t root = -1;

    template <class Emit>
    int32_t memo(std::unordered_map<const void*, int32_t>& table, const void* key, Emit emit) {
        auto it = table.find(key);
        if (it != table.end()) return it->second;
        int32_t id = emit();
        table[key] = id;
        return id;
    }
    std::unordered_map<const void*, int32_t> node_ids, material_ids, texture_ids, image_ids;

    int32_t add_node(int32_t kind, int32_t a, int32_t b = 0, int32_t c = 0) {
        nodes.push_back(rtk_node{kind, a, b, c});
        return int32_t(nodes.size()) - 1;
    }

    rtk_scene_desc desc() const {
        rtk_scene_desc d;
        std::memset(&d, 0, sizeof d);
        d.abi_version = RTK_ABI_VERSION;
        d.root = root;
        d.n_nodes = int32_t(nodes.size());
        d.n_list_children = int32_t(list_children.size());
        d.n_spheres = int32_t(spheres.size());
        d.n_quads = int32_t(quads.size());
        d.n_triangles = int32_t(triangles.size());
        d.n_bvh_boxes = int32_t(bvh_boxes.size());
        d.n_translates = int32_t(translates.size());
        d.n_rotates = int32_t(rotates.size());
        d.n_media = int32_t(media.size());
        d.n_materials = int32_t(materials.size());
        d.n_textures = int32_t(textures.size());
        d.n_images = int32_t(images.size());
        d.n_perlins = int32_t(perlins.size());
        d.n_lights = int32_t(lights.size());
        d.n_texel_bytes = int64_t(texels.size());
        d.nodes = nodes.data();
        d.list_children = list_children.data();
        d.spheres = spheres.data();
        d.quads = quads.data();
        d.triangles = triangles.data();
        d.bvh_boxes = bvh_boxes.data();
        d.translates = translates.data();
        d.rotates = rotates.data();
        d.media = media.data();
        d.materials = materials.data();
        d.textures = textures.data();
        d.images = images.data();
        d.texels = texels.data();
        d.perlins = perlins.data();
        d.lights = lights.data();
        return d;
    }
};

}  // namespace rtk

// ===========================================================================
// Image data behind image_texture (reference: rtw_stb_image.h).  This loader
// reads baseline JPEG (rtk_jpeg.h: the reference's earthmap.jpg / male_texture.jpg,
// decoded to the very bytes stb_image yields -- SURVEY 8(f) row 4), PNG (rtk_png.h),
// binary PPM (P6) and raw RGB8 buffers.  The reference pipeline is
// stbi_loadf -> float -> float_to_byte (rtw_stb_image.h:53-66,99-121), i.e. the
// texels the sample loop sees are int(256 * (b/255)^2.2); from_file applies the
// same mapping so that a PPM gives the bytes the reference would hold.
// ===========================================================================
class rtw_image {
public:
    rtw_image() {}
    rtw_image(const char* image_filename) {
        std::string name(image_filename);
        const char* dir = getenv("RTW_IMAGES");
        if (dir && load(std::string(dir) + "/" + name)) return;
        if (load(name)) return;
        std::string prefix = "images/";
        for (int up = 0; up < 7; up++) {
            if (load(prefix + name)) return;
            prefix = "../" + prefix;
        }
        std::cerr << "ERROR: Could not load image file '" << image_filename << "'.\n";
    }
    // Bytes exactly as the sample loop should see them (no gamma mapping).
    static shared_ptr<rtw_image> from_rgb8(int w, int h, const uint8_t* rgb) {
        auto im = make_shared<rtw_image>();
        im->w = w;
        im->h = h;
        im->bytes.assign(rgb, rgb + size_t(w) * h * 3);
        return im;
    }
    // false when the file cannot be read or decoded -- including when decoding it would need more memory than the
    // machine has: an image that fails to load renders cyan (texture.h:92), it never takes the program down.
    bool load(const std::string& path) {
        try {
            return load_or_throw(path);
        } catch (const std::bad_alloc&) {
            bytes.clear();
            return false;
        } catch (const std::length_error&) {
            bytes.clear();
            return false;
        }
    }
    bool load_or_throw(const std::string& path) {
        std::ifstream f(path, std::ios::binary);
        if (!f.good()) return false;
        std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        std::vector<uint8_t> raw;
        int iw = 0, ih = 0;
        if (file.size() > 3 && file[0] == 0xFF && file[1] == 0xD8) {
            rtk::jpeg_decoder jpeg;  // baseline JPEG, decoded to the bytes stb_image produces (rtk_jpeg.h)
            if (!jpeg.decode(file.data(), file.size(), iw, ih, raw)) return false;
        } else if (file.size() > 8 && file[0] == 0x89 && file[1] == 'P' && file[2] == 'N' && file[3] == 'G') {
            rtk::png_decoder png;  // every PNG colour type / depth / interlace, reduced to RGB8 the way stbi_load(..., 3) does (rtk_png.h)
            if (!png.decode(file.data(), file.size(), iw, ih, raw)) return false;
        } else if (!parse_ppm(file, iw, ih, raw)) {
            return false;
        }
        w = iw;
        h = ih;
        bytes.resize(raw.size());
        // stbi_loadf turns 8-bit data into "linear" floats, pow(b/255, 2.2) in float (stb_image.h:1858-1873), and
        // rtw_image::float_to_byte (rtw_stb_image.h:99-105) truncates 256*v: that is what the sample loop sees
        uint8_t map[256];
        for (int b = 0; b < 256; b++) {
            float lin = float(std::pow(b / 255.0f, 2.2f) * 1.0f);
            map[b] = lin <= 0.0 ? 0 : (1.0 <= lin ? 255 : static_cast<unsigned char>(256.0 * lin));
        }
        for (size_t i = 0; i < raw.size(); i++) bytes[i] = map[raw[i]];
        return true;
    }
    // Binary PPM (P6, maxval 255).
    static bool parse_ppm(const std::vector<uint8_t>& file, int& iw, int& ih, std::vector<uint8_t>& raw) {
        if (file.size() < 2 || file[0] != 'P' || file[1] != '6') return false;
        size_t pos = 2;
        int vals[3], got = 0;
        while (got < 3 && pos < file.size()) {
            const int c = file[pos];
            if (c == '#') {
                while (pos < file.size() && file[pos] != '\n') pos++;
                continue;
            }
            if (isspace(c)) {
                pos++;
                continue;
            }
            if (c < '0' || c > '9') return false;
            long long v = 0;
            while (pos < file.size() && file[pos] >= '0' && file[pos] <= '9') {
                v = v * 10 + (file[pos++] - '0');
                if (v > (1 << 24)) return false;  // no dimension (or maxval) is that large
            }
            vals[got++] = int(v);
        }
        if (got < 3 || vals[2] != 255 || vals[0] <= 0 || vals[1] <= 0) return false;
        pos++;  // the single whitespace byte after maxval
        const size_t n = size_t(vals[0]) * vals[1] * 3;
        if (file.size() - pos < n) return false;
        raw.assign(file.begin() + long(pos), file.begin() + long(pos + n));
        iw = vals[0];
        ih = vals[1];
        return true;
    }
    int width() const { return bytes.empty() ? 0 : w; }
    int height() const { return bytes.empty() ? 0 : h; }
    const unsigned char* pixel_data(int x, int y) const {
        static unsigned char magenta[] = {255, 0, 255};
        if (bytes.empty()) return magenta;
        x = x < 0 ? 0 : (x < w ? x : w - 1);
        y = y < 0 ? 0 : (y < h ? y : h - 1);
        return bytes.data() + (size_t(y) * w + x) * 3;
    }
    const std::vector<uint8_t>& data() const { return bytes; }

private:
    int w = 0, h = 0;
    std::vector<uint8_t> bytes;
};

// ===========================================================================
// Perlin tables (reference: perlin.h:6-13,59-71).  Generated on the host from
// the scene RNG in the reference's draw order; the tables are device inputs.
// ===========================================================================
class perlin {
public:
    static const int point_count = 256;
    perlin() {
        for (int i = 0; i < point_count; i++) randVec[i] = unit_vector(vec3::random(-1, 1));
        generate_perm(perm_x);
        generate_perm(perm_y);
        generate_perm(perm_z);
    }
    void export_tables(rtk_perlin& out) const {
        for (int i = 0; i < point_count; i++) {
            out.randvec[i][0] = randVec[i].x();
            out.randvec[i][1] = randVec[i].y();
            out.randvec[i][2] = randVec[i].z();
            out.perm_x[i] = perm_x[i];
            out.perm_y[i] = perm_y[i];
            out.perm_z[i] = perm_z[i];
        }
    }

private:
    vec3 randVec[point_count];
    int perm_x[point_count], perm_y[point_count], perm_z[point_count];
    // perlin.h:64-71 swaps with random_int(0, 1), not (0, i): kept.
    static void generate_perm(int* p) {
        for (int i = 0; i < point_count; i++) p[i] = i;
        for (int i = point_count - 1; i > 0; i--) {
            int target = random_int(0, 1);
            std::swap(p[i], p[target]);
        }
    }
};

// ===========================================================================
// Textures
// ===========================================================================
class texture {
public:
    virtual ~texture() = default;
    virtual int32_t rtk_emit(rtk::scene_builder& sb) const = 0;
    int32_t rtk_id(rtk::scene_builder& sb) const {
        return sb.memo(sb.texture_ids, this, [&] { return rtk_emit(sb); });
    }

protected:
    static int32_t push(rtk::scene_builder& sb, int32_t kind, int32_t even, int32_t odd, int32_t image, const color& c, double param) {
        sb.textures.push_back(rtk_texture{kind, even, odd, image, rtk::to_abi(c), param});
        return int32_t(sb.textures.size()) - 1;
    }
};

class solid_color : public texture {
public:
    solid_color(const color& albedo) : albedo(albedo) {}
    solid_color(double r, double g, double b) : albedo(r, g, b) {}
    int32_t rtk_emit(rtk::scene_builder& sb) const override { return push(sb, RTK_TEX_SOLID, -1, -1, -1, albedo, 0); }

private:
    color albedo;
};

class checker_texture : public texture {
public:
    checker_texture(double scale, shared_ptr<texture> even, shared_ptr<texture> odd) : inv_scale(1.0 / scale), even(even), odd(odd) {}
    checker_texture(double scale, const color& c1, const color& c2)
        : checker_texture(scale, make_shared<solid_color>(c1), make_shared<solid_color>(c2)) {}
    int32_t rtk_emit(rtk::scene_builder& sb) const override {
        int32_t e = even->rtk_id(sb), o = odd->rtk_id(sb);
        return push(sb, RTK_TEX_CHECKER, e, o, -1, color(), inv_scale);
    }

private:
    double inv_scale;
    shared_ptr<texture> even, odd;
};

class checker_texture_triangle : public texture {
public:
    checker_texture_triangle(double scale, shared_ptr<texture> even, shared_ptr<texture> odd)
        : inv_scale(1.0 / std::max(0.01, scale)), even(even), odd(odd) {}
    checker_texture_triangle(double scale, const color& c1, const color& c2)
        : checker_texture_triangle(scale, make_shared<solid_color>(c1), make_shared<solid_color>(c2)) {}
    int32_t rtk_emit(rtk::scene_builder& sb) const override {
        int32_t e = even->rtk_id(sb), o = odd->rtk_id(sb);
        return push(sb, RTK_TEX_CHECKER_TRI, e, o, -1, color(), inv_scale);
    }

private:
    double inv_scale;
    shared_ptr<texture> even, odd;
};

class image_texture : public texture {
public:
    image_texture(const char* filename) : image(make_shared<rtw_image>(filename)) {}
    // Extension: wrap texels that are already in memory (synthetic textures).
    image_texture(shared_ptr<rtw_image> im) : image(im) {}
    int32_t rtk_emit(rtk::scene_builder& sb) const override {
        int32_t im = sb.memo(sb.image_ids, image.get(), [&] {
            rtk_image rec{image->width(), image->height(), int64_t(sb.texels.size())};
            sb.texels.insert(sb.texels.end(), image->data().begin(), image->data().end());
            sb.images.push_back(rec);
            return int32_t(sb.images.size()) - 1;
        });
        return push(sb, RTK_TEX_IMAGE, -1, -1, im, color(), 0);
    }

private:
    shared_ptr<rtw_image> image;
};

class noise_texture : public texture {
public:
    noise_texture(double scale) : scale(scale) {}
    int32_t rtk_emit(rtk::scene_builder& sb) const override {
        sb.perlins.emplace_back();
        noise.export_tables(sb.perlins.back());
        return push(sb, RTK_TEX_NOISE, -1, -1, int32_t(sb.perlins.size()) - 1, color(), scale);
    }

private:
    perlin noise;  // constructed (and its RNG draws taken) before `scale`, as texture.h:118-119
    double scale;
};

// ===========================================================================
// Materials
// ===========================================================================
class material {
public:
    virtual ~material() = default;
    virtual int32_t rtk_emit(rtk::scene_builder& sb) const = 0;
    int32_t rtk_id(rtk::scene_builder& sb) const {
        return sb.memo(sb.material_ids, this, [&] { return rtk_emit(sb); });
    }

protected:
    static int32_t push(rtk::scene_builder& sb, int32_t kind, int32_t tex, const color& albedo, double param) {
        sb.materials.push_back(rtk_material{kind, tex, rtk::to_abi(albedo), param});
        return int32_t(sb.materials.size()) - 1;
    }
};

class lambertian : public material {
public:
    lambertian(const color& albedo) : tex(make_shared<solid_color>(albedo)) {}
    lambertian(shared_ptr<texture> tex) : tex(tex) {}
    int32_t rtk_emit(rtk::scene_builder& sb) const override { return push(sb, RTK_MAT_LAMBERTIAN, tex->rtk_id(sb), color(), 0); }

private:
    shared_ptr<texture> tex;
};

class dielectric : public material {
public:
    dielectric(double refraction_index) : refraction_index(refraction_index) {}
    int32_t rtk_emit(rtk::scene_builder& sb) const override { return push(sb, RTK_MAT_DIELECTRIC, -1, color(), refraction_index); }

private:
    double refraction_index;
};

class metal : public material {
public:
    metal(const color& albedo, double fuzz) : albedo(albedo), fuzz(fuzz < 1 ? fuzz : 1) {}
    int32_t rtk_emit(rtk::scene_builder& sb) const override { return push(sb, RTK_MAT_METAL, -1, albedo, fuzz); }

private:
    color albedo;
    double fuzz;
};

class diffuse_light : public material {
public:
    diffuse_light(shared_ptr<texture> tex) : tex(tex) {}
    diffuse_light(const color& emit) : tex(make_shared<solid_color>(emit)) {}
    int32_t rtk_emit(rtk::scene_builder& sb) const override { return push(sb, RTK_MAT_DIFFUSE_LIGHT, tex->rtk_id(sb), color(), 0); }

private:
    shared_ptr<texture> tex;
};

// Behaviourally identical to diffuse_light in the reference (material.h:94-122):
// both emit tex->value and never scatter.
class emissive_light : public material {
public:
    emissive_light(shared_ptr<texture> tex) : tex(tex) {}
    emissive_light(const color& emit) : tex(make_shared<solid_color>(emit)) {}
    int32_t rtk_emit(rtk::scene_builder& sb) const override { return push(sb, RTK_MAT_DIFFUSE_LIGHT, tex->rtk_id(sb), color(), 0); }

private:
    shared_ptr<texture> tex;
};

class isotropic : public material {
public:
    isotropic(const color& albedo) : tex(make_shared<solid_color>(albedo)) {}
    isotropic(shared_ptr<texture> tex) : tex(tex) {}
    int32_t rtk_emit(rtk::scene_builder& sb) const override { return push(sb, RTK_MAT_ISOTROPIC, tex->rtk_id(sb), color(), 0); }

private:
    shared_ptr<texture> tex;
};

class specular : public material {
public:
    specular(const color& albedo, double shininess) : albedo(albedo), shininess(shininess) {}
    int32_t rtk_emit(rtk::scene_builder& sb) const override { return push(sb, RTK_MAT_SPECULAR, -1, albedo, shininess); }

private:
    color albedo;
    double shininess;
};

// ===========================================================================
// Hittables
// ===========================================================================
class hittable {
public:
    virtual ~hittable() = default;
    virtual aabb bounding_box() const = 0;
    virtual int32_t rtk_emit(rtk::scene_builder& sb) const = 0;
    int32_t rtk_id(rtk::scene_builder& sb) const {
        return sb.memo(sb.node_ids, this, [&] { return rtk_emit(sb); });
    }
};

class translate : public hittable {
public:
    translate(shared_ptr<hittable> object, const vec3& offset) : object(object), offset(offset) { bbox = object->bounding_box() + offset; }
    aabb bounding_box() const override { return bbox; }
    int32_t rtk_emit(rtk::scene_builder& sb) const override {
        int32_t child = object->rtk_id(sb);
        sb.translates.push_back(rtk_translate{rtk::to_abi(offset)});
        return sb.add_node(RTK_NODE_TRANSLATE, int32_t(sb.translates.size()) - 1, child);
    }

private:
    shared_ptr<hittable> object;
    vec3 offset;
    aabb bbox;
};

class rotate_y : public hittable {
public:
    rotate_y(shared_ptr<hittable> object, double angle) : object(object) {
        double radians = degrees_to_radians(angle);
        sin_theta = std::sin(radians);
        cos_theta = std::cos(radians);
        aabb in = object->bounding_box();
        point3 lo(infinity, infinity, infinity), hi(-infinity, -infinity, -infinity);
        // Box of the eight rotated corners, corner order i,j,k as hittable.h:78-95.
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                for (int k = 0; k < 2; k++) {
                    double cx = i * in.x.max + (1 - i) * in.x.min;
                    double cy = j * in.y.max + (1 - j) * in.y.min;
                    double cz = k * in.z.max + (1 - k) * in.z.min;
                    vec3 corner(cos_theta * cx + sin_theta * cz, cy, -sin_theta * cx + cos_theta * cz);
                    for (int c = 0; c < 3; c++) {
                        lo[c] = std::fmin(lo[c], corner[c]);
                        hi[c] = std::fmax(hi[c], corner[c]);
                    }
                }
        bbox = aabb(lo, hi);
    }
    aabb bounding_box() const override { return bbox; }
    int32_t rtk_emit(rtk::scene_builder& sb) const override {
        int32_t child = object->rtk_id(sb);
        sb.rotates.push_back(rtk_rotate_y{sin_theta, cos_theta});
        return sb.add_node(RTK_NODE_ROTATE_Y, int32_t(sb.rotates.size()) - 1, child);
    }

private:
    shared_ptr<hittable> object;
    double sin_theta, cos_theta;
    aabb bbox;
};

class hittable_list : public hittable {
public:
    std::vector<shared_ptr<hittable>> objects;
    hittable_list() {}
    hittable_list(shared_ptr<hittable> object) { add(object); }
    void clear() { objects.clear(); }
    void add(shared_ptr<hittable> object) {
        objects.push_back(object);
        bbox = aabb(bbox, object->bounding_box());
    }
    aabb bounding_box() const override { return bbox; }
    int32_t rtk_emit(rtk::scene_builder& sb) const override {
        std::vector<int32_t> ids;
        ids.reserve(objects.size());
        for (const auto& o : objects) ids.push_back(o->rtk_id(sb));
        int32_t first = int32_t(sb.list_children.size());
        sb.list_children.insert(sb.list_children.end(), ids.begin(), ids.end());
        return sb.add_node(RTK_NODE_LIST, first, int32_t(ids.size()));
    }

private:
    aabb bbox;
};

// The reference's builder (bvh.h:13-62): box of the span, longest axis of that
// box, std::sort on the objects' own box minimum, median split; a span of one
// puts the same object on both sides, a span of two is not sorted.  Using the
// same libstdc++ std::sort on the same keys gives the same topology as the
// reference compiled here.
class bvh_node : public hittable {
public:
    bvh_node(hittable_list list) : bvh_node(list.objects, 0, list.objects.size()) {}
    bvh_node(std::vector<shared_ptr<hittable>>& objects, size_t start, size_t end) {
        bbox = aabb::empty;
        for (size_t i = start; i < end; i++) bbox = aabb(bbox, objects[i]->bounding_box());
        const int axis = bbox.longest_axis();
        const size_t span = end - start;
        if (span == 0) {
            // bvh.h:38-43 recurses forever here; report instead.
            std::cerr << "bvh_node: empty object list\n";
            std::abort();
        }
        if (span == 1) {
            left = right = objects[start];
        } else if (span == 2) {
            left = objects[start];
            right = objects[start + 1];
        } else {
            std::sort(objects.begin() + start, objects.begin() + end,
                      [axis](const shared_ptr<hittable> a, const shared_ptr<hittable> b) {
                          return a->bounding_box().axis_interval(axis).min < b->bounding_box().axis_interval(axis).min;
                      });
            size_t mid = start + span / 2;
            left = make_shared<bvh_node>(objects, start, mid);
            right = make_shared<bvh_node>(objects, mid, end);
        }
    }
    aabb bounding_box() const override { return bbox; }
    int32_t rtk_emit(rtk::scene_builder& sb) const override {
        int32_t l = left->rtk_id(sb), r = right->rtk_id(sb);
        sb.bvh_boxes.push_back(rtk_aabb{bbox.x.min, bbox.x.max, bbox.y.min, bbox.y.max, bbox.z.min, bbox.z.max});
        return sb.add_node(RTK_NODE_BVH, l, r, int32_t(sb.bvh_boxes.size()) - 1);
    }

private:
    shared_ptr<hittable> left, right;
    aabb bbox;
};

class sphere : public hittable {
public:
    // The box uses the constructor ARGUMENT `radius`, not the clamped member
    // (the parameter shadows the member in sphere.h:15-16).
    sphere(const point3& static_center, double radius, shared_ptr<material> mat)
        : center(static_center, vec3(0, 0, 0)), radius(std::fmax(0, radius)), mat(mat) {
        vec3 rvec(radius, radius, radius);
        bbox = aabb(static_center - rvec, static_center + rvec);
    }
    sphere(const point3& center1, const point3& center2, double radius, shared_ptr<material> mat)
        : center(center1, center2 - center1), radius(std::fmax(0, radius)), mat(mat) {
        vec3 rvec(radius, radius, radius);
        aabb box1(center.at(0) - rvec, center.at(0) + rvec);
        aabb box2(center.at(1) - rvec, center.at(1) + rvec);
        bbox = aabb(box1, box2);
    }
    aabb bounding_box() const override { return bbox; }
    int32_t rtk_emit(rtk::scene_builder& sb) const override {
        sb.spheres.push_back(rtk_sphere{rtk::to_abi(center.origin()), rtk::to_abi(center.direction()), radius, mat->rtk_id(sb), 0});
        return sb.add_node(RTK_NODE_SPHERE, int32_t(sb.spheres.size()) - 1);
    }

private:
    ray center;
    double radius;
    shared_ptr<material> mat;
    aabb bbox;
};

class quad : public hittable {
public:
    quad(const point3& Q, const vec3& u, const vec3& v, shared_ptr<material> mat) : Q(Q), u(u), v(v), mat(mat) {
        vec3 n = cross(u, v);
        normal = unit_vector(n);
        D = dot(normal, Q);
        w = n / dot(n, n);
        set_bounding_box();
    }
    virtual void set_bounding_box() {
        aabb d1(Q, Q + u + v), d2(Q + u, Q + v);
        bbox = aabb(d1, d2);
    }
    aabb bounding_box() const override { return bbox; }
    int32_t rtk_emit(rtk::scene_builder& sb) const override {
        sb.quads.push_back(rtk_quad{rtk::to_abi(Q), rtk::to_abi(u), rtk::to_abi(v), rtk::to_abi(w), rtk::to_abi(normal), D, mat->rtk_id(sb), 0});
        return sb.add_node(RTK_NODE_QUAD, int32_t(sb.quads.size()) - 1);
    }

private:
    point3 Q;
    vec3 u, v, w;
    shared_ptr<material> mat;
    aabb bbox;
    vec3 normal;
    double D;
};

// quad.h:86-108: six quads, front/right/back/left/top/bottom.
inline shared_ptr<hittable_list> box(const point3& a, const point3& b, shared_ptr<material> mat) {
    auto sides = make_shared<hittable_list>();
    point3 lo(std::fmin(a.x(), b.x()), std::fmin(a.y(), b.y()), std::fmin(a.z(), b.z()));
    point3 hi(std::fmax(a.x(), b.x()), std::fmax(a.y(), b.y()), std::fmax(a.z(), b.z()));
    vec3 dx(hi.x() - lo.x(), 0, 0), dy(0, hi.y() - lo.y(), 0), dz(0, 0, hi.z() - lo.z());
    sides->add(make_shared<quad>(point3(lo.x(), lo.y(), hi.z()), dx, dy, mat));
    sides->add(make_shared<quad>(point3(hi.x(), lo.y(), hi.z()), -dz, dy, mat));
    sides->add(make_shared<quad>(point3(hi.x(), lo.y(), lo.z()), -dx, dy, mat));
    sides->add(make_shared<quad>(point3(lo.x(), lo.y(), lo.z()), dz, dy, mat));
    sides->add(make_shared<quad>(point3(lo.x(), hi.y(), hi.z()), dx, -dz, mat));
    sides->add(make_shared<quad>(point3(lo.x(), lo.y(), lo.z()), dx, dz, mat));
    return sides;
}

class triangle : public hittable {
public:
    triangle(vec3 p0, vec3 p1, vec3 p2, std::shared_ptr<material> mat)
        : triangle(p0, p1, p2, mat, glm::vec2(0, 0), glm::vec2(1, 0), glm::vec2(0, 1)) {}
    // The reference's "wrap UVs into [0,1)" assigns to the shadowing parameters
    // (triangle.h:40-42), so the stored UVs are the raw ones.
    triangle(vec3 p0, vec3 p1, vec3 p2, std::shared_ptr<material> mat, glm::vec2 uv0, glm::vec2 uv1, glm::vec2 uv2)
        : p0(p0), p1(p1), p2(p2), mat(mat), uv0(uv0), uv1(uv1), uv2(uv2) {
        normal = unit_vector(cross(p1 - p0, p2 - p0));
        set_bounding_box();
    }
    // Two-point box: NOT padded (aabb.h:21-45).  Only used as a sort
    // key / merge input on the host; the device never slab-tests it.
    virtual void set_bounding_box() {
        vec3 lo(std::min({p0.x(), p1.x(), p2.x()}), std::min({p0.y(), p1.y(), p2.y()}), std::min({p0.z(), p1.z(), p2.z()}));
        vec3 hi(std::max({p0.x(), p1.x(), p2.x()}), std::max({p0.y(), p1.y(), p2.y()}), std::max({p0.z(), p1.z(), p2.z()}));
        bbox = aabb(lo, hi);
    }
    aabb bounding_box() const override { return bbox; }
    int32_t rtk_emit(rtk::scene_builder& sb) const override {
        rtk_triangle t;
        t.p0 = rtk::to_abi(p0);
        t.p1 = rtk::to_abi(p1);
        t.p2 = rtk::to_abi(p2);
        t.normal = rtk::to_abi(normal);
        t.uv0[0] = uv0.x; t.uv0[1] = uv0.y;
        t.uv1[0] = uv1.x; t.uv1[1] = uv1.y;
        t.uv2[0] = uv2.x; t.uv2[1] = uv2.y;
        t.material = mat->rtk_id(sb);
        t._pad = 0;
        sb.triangles.push_back(t);
        return sb.add_node(RTK_NODE_TRIANGLE, int32_t(sb.triangles.size()) - 1);
    }

private:
    vec3 p0, p1, p2;
    std::shared_ptr<material> mat;
    aabb bbox;
    glm::vec2 uv0, uv1, uv2;
    vec3 normal;
};

// triangle.h:146-169, including its `height + orig.x()` slip.
inline std::shared_ptr<hittable_list> triangle_quad(const point3& orig, double height, double width, shared_ptr<material> mat) {
    auto sides = make_shared<hittable_list>();
    sides->add(make_shared<triangle>(point3(orig), vec3(orig.x(), height + orig.x(), orig.z()), vec3(width + orig.x(), orig.y(), orig.z()), mat));
    sides->add(make_shared<triangle>(point3(orig.x() + width, orig.y(), orig.z()), vec3(orig.x() + width, orig.y() + height, orig.z()),
                                     vec3(orig.x(), height + orig.y(), orig.z()), mat));
    return sides;
}

class constant_medium : public hittable {
public:
    constant_medium(shared_ptr<hittable> boundary, double density, shared_ptr<texture> tex)
        : boundary(boundary), neg_inv_density(-1 / density), phase_function(make_shared<isotropic>(tex)) {}
    constant_medium(shared_ptr<hittable> boundary, double density, const color& albedo)
        : boundary(boundary), neg_inv_density(-1 / density), phase_function(make_shared<isotropic>(albedo)) {}
    aabb bounding_box() const override { return boundary->bounding_box(); }
    int32_t rtk_emit(rtk::scene_builder& sb) const override {
        int32_t b = boundary->rtk_id(sb);
        sb.media.push_back(rtk_medium{neg_inv_density, phase_function->rtk_id(sb), 0});
        return sb.add_node(RTK_NODE_MEDIUM, int32_t(sb.media.size()) - 1, b);
    }

private:
    shared_ptr<hittable> boundary;
    double neg_inv_density;
    shared_ptr<material> phase_function;
};

// point_light.h:9-28 without the CUDA qualifiers.
class point_light {
public:
    point_light(point3 position, color intensity, double size) : position(position), intensity(intensity), size(size) {}
    point3 get_position() const { return position; }
    color get_intensity() const { return intensity; }
    double get_size() const { return size; }

private:
    point3 position;
    color intensity;
    double size;
};

namespace rtk {
// Flatten `world` (+ lights) into builder tables; returns the ABI view.  The
// builder must outlive every use of the returned description.
inline rtk_scene_desc flatten(const hittable& world, const std::vector<point_light>& lights, scene_builder& sb) {
    sb.root = world.rtk_id(sb);
    for (const auto& l : lights) sb.lights.push_back(rtk_point_light{to_abi(l.get_position()), to_abi(l.get_intensity()), l.get_size()});
    return sb.desc();
}
}  // namespace rtk

#endif  // RTK_SCENE_API_H
