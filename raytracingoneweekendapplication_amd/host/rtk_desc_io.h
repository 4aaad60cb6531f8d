// rtk_desc_io.h -- save / load an rtk_scene_desc as one flat binary file.
//
// Depends only on include/rtk.h, so both the product's host library and the
// reference-side oracle driver can write the same format; tests compare the two
// files byte for byte (host flattener + BVH builder parity) and can feed either
// to the device and to the CPU oracle.
//
// Layout: "RTKSCN1\0" | int32 root, int32 counts[15] (order of rtk_scene_desc)
//         | int64 n_texel_bytes | the 15 tables, raw, in rtk_scene_desc order.
#ifndef RTK_DESC_IO_H
#define RTK_DESC_IO_H

#include <cstdio>
#include <cstring>
#include <vector>

#include "rtk.h"

namespace rtk {

inline bool save_desc(const rtk_scene_desc& d, const char* path) {
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    const char magic[8] = {'R', 'T', 'K', 'S', 'C', 'N', '1', 0};
    std::fwrite(magic, 1, 8, f);
    const int32_t head[16] = {d.root, d.n_nodes, d.n_list_children, d.n_spheres, d.n_quads, d.n_triangles, d.n_bvh_boxes, d.n_translates,
                              d.n_rotates, d.n_media, d.n_materials, d.n_textures, d.n_images, d.n_perlins, d.n_lights, 0};
    std::fwrite(head, sizeof(int32_t), 16, f);
    std::fwrite(&d.n_texel_bytes, sizeof(int64_t), 1, f);
    auto put = [&](const void* p, size_t elem, int64_t n) {
        if (n > 0) std::fwrite(p, elem, size_t(n), f);
    };
    put(d.nodes, sizeof(rtk_node), d.n_nodes);
    put(d.list_children, sizeof(int32_t), d.n_list_children);
    put(d.spheres, sizeof(rtk_sphere), d.n_spheres);
    put(d.quads, sizeof(rtk_quad), d.n_quads);
    put(d.triangles, sizeof(rtk_triangle), d.n_triangles);
    put(d.bvh_boxes, sizeof(rtk_aabb), d.n_bvh_boxes);
    put(d.translates, sizeof(rtk_translate), d.n_translates);
    put(d.rotates, sizeof(rtk_rotate_y), d.n_rotates);
    put(d.media, sizeof(rtk_medium), d.n_media);
    put(d.materials, sizeof(rtk_material), d.n_materials);
    put(d.textures, sizeof(rtk_texture), d.n_textures);
    put(d.images, sizeof(rtk_image), d.n_images);
    put(d.texels, 1, d.n_texel_bytes);
    put(d.perlins, sizeof(rtk_perlin), d.n_perlins);
    put(d.lights, sizeof(rtk_point_light), d.n_lights);
    bool ok = std::ferror(f) == 0;
    std::fclose(f);
    return ok;
}

// Owns the tables of a loaded description.
struct desc_storage {
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
    rtk_scene_desc desc;

    bool load(const char* path) {
        FILE* f = std::fopen(path, "rb");
        if (!f) return false;
        char magic[8];
        int32_t head[16];
        int64_t ntex = 0;
        bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "RTKSCN1", 8) == 0 && std::fread(head, sizeof(int32_t), 16, f) == 16 &&
                  std::fread(&ntex, sizeof(int64_t), 1, f) == 1;
        // a count is believed only if the rest of the file can hold that many records (a negative or absurd count in a
        // damaged file must not size an allocation)
        int64_t remaining = 0;
        if (ok) {
            const long here = std::ftell(f);
            if (here < 0 || std::fseek(f, 0, SEEK_END) != 0) ok = false;
            const long size = ok ? std::ftell(f) : -1;
            if (size < here || std::fseek(f, here, SEEK_SET) != 0) ok = false;
            remaining = ok ? int64_t(size - here) : 0;
        }
        auto get = [&](auto& vec, int64_t n) {
            vec.clear();
            if (!ok) return;
            const int64_t bytes_each = int64_t(sizeof(vec[0]));
            if (n < 0 || n > remaining / bytes_each) {
                ok = false;
                return;
            }
            vec.resize(size_t(n));
            if (n > 0) ok = std::fread(vec.data(), sizeof(vec[0]), size_t(n), f) == size_t(n);
            remaining -= n * bytes_each;
        };
        if (ok) {
            get(nodes, head[1]); get(list_children, head[2]); get(spheres, head[3]); get(quads, head[4]);
            get(triangles, head[5]); get(bvh_boxes, head[6]); get(translates, head[7]); get(rotates, head[8]);
            get(media, head[9]); get(materials, head[10]); get(textures, head[11]); get(images, head[12]);
            get(texels, ntex); get(perlins, head[13]); get(lights, head[14]);
        }
        std::fclose(f);
        if (!ok) return false;
        std::memset(&desc, 0, sizeof desc);
        desc.abi_version = RTK_ABI_VERSION;
        desc.root = head[0];
        desc.n_nodes = head[1]; desc.n_list_children = head[2]; desc.n_spheres = head[3]; desc.n_quads = head[4];
        desc.n_triangles = head[5]; desc.n_bvh_boxes = head[6]; desc.n_translates = head[7]; desc.n_rotates = head[8];
        desc.n_media = head[9]; desc.n_materials = head[10]; desc.n_textures = head[11]; desc.n_images = head[12];
        desc.n_perlins = head[13]; desc.n_lights = head[14];
        desc.n_texel_bytes = ntex;
        desc.nodes = nodes.data(); desc.list_children = list_children.data(); desc.spheres = spheres.data();
        desc.quads = quads.data(); desc.triangles = triangles.data(); desc.bvh_boxes = bvh_boxes.data();
        desc.translates = translates.data(); desc.rotates = rotates.data(); desc.media = media.data();
        desc.materials = materials.data(); desc.textures = textures.data(); desc.images = images.data();
        desc.texels = texels.data(); desc.perlins = perlins.data(); desc.lights = lights.data();
        return true;
    }
};

}  // namespace rtk

#endif  // RTK_DESC_IO_H
