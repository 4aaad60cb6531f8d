// rtk_scenes.cpp -- host library (librtk_host.so): the BASELINE.json scenes built
// through the drop-in scene API of this package and flattened for the device.
//
// A small C interface lets bench.py / the tests obtain the flattened scene and
// the derived camera without a C++ toolchain on the GPU box.  Nothing here traces
// rays; rendering is rtk_render_* of librtk_hip.so.
#include <cstring>
#include <string>
#include <vector>

#include "camera.h"
#include "mesh.h"
#include "rtk_desc_io.h"
#include "scenes/scene_library.h"

struct rtkh_scene {
    rtk::scene_builder builder;   // owns tables when built from the API
    rtk::desc_storage storage;    // owns tables when loaded from a file
    rtk_scene_desc desc;
    rtk_view view;
    bool has_view = false;
    uint64_t scene_rng_draws = 0;
};

extern "C" {

// Build one of the named scenes (scene_library.h).  scene_seed seeds the
// construction RNG (random spheres, box heights, Perlin tables).
rtkh_scene* rtkh_scene_build(const char* name, uint32_t scene_seed, const char* image_file) {
    auto* s = new rtkh_scene;
    rtk::seed_scene_rng(scene_seed);
    rtk_scene_def def;
    if (!rtk_build_named_scene(name, image_file ? image_file : "", def)) {
        delete s;
        return nullptr;
    }
    std::vector<point_light> lights;
    for (const auto& l : def.lights) lights.emplace_back(l.position, l.intensity, l.size);
    s->desc = rtk::flatten(def.world, lights, s->builder);
    s->view = def.view;
    s->has_view = true;
    s->scene_rng_draws = rtk::host_rng().draws;
    return s;
}

rtkh_scene* rtkh_scene_load(const char* path) {
    auto* s = new rtkh_scene;
    if (!s->storage.load(path)) {
        delete s;
        return nullptr;
    }
    s->desc = s->storage.desc;
    return s;
}

void rtkh_scene_free(rtkh_scene* s) { delete s; }

const rtk_scene_desc* rtkh_scene_desc(const rtkh_scene* s) { return s ? &s->desc : nullptr; }

int rtkh_scene_save(const rtkh_scene* s, const char* path) { return (s && rtk::save_desc(s->desc, path)) ? 0 : -1; }

uint64_t rtkh_scene_rng_draws(const rtkh_scene* s) { return s ? s->scene_rng_draws : 0; }

// Derived camera for the scene's view; width/height/spp/depth <= 0 keep the
// scene's own (BASELINE config) values.  Goes through camera::derive(), i.e. the
// same code path camera::render() uses.
int rtkh_scene_camera(const rtkh_scene* s, int width, int height, int spp, int depth, rtk_camera* out) {
    if (!s || !s->has_view || !out) return -1;
    rtk_view v = s->view;
    if (width > 0) v.image_width = width;
    if (height > 0) v.image_height = height;
    if (spp > 0) v.samples_per_pixel = spp;
    if (depth > 0) v.max_depth = depth;
    camera cam;
    cam.image_width = v.image_width;
    cam.aspect_ratio = double(v.image_width) / double(v.image_height);
    cam.samples_per_pixel = v.samples_per_pixel;
    cam.max_depth = v.max_depth;
    cam.background = v.background;
    cam.vfov = v.vfov;
    cam.lookfrom = v.lookfrom;
    cam.lookat = v.lookat;
    cam.vup = v.vup;
    cam.defocus_angle = v.defocus_angle;
    cam.focus_dist = v.focus_dist;
    *out = cam.derive();
    if (out->image_height != v.image_height) {
        // int(W / (W/H)) can round down by one (Camera.txt:137); the viewport
        // maths below it only depends on the final height, so set it and redo.
        cam.aspect_ratio = double(v.image_width) / (double(v.image_height) + 0.5);
        *out = cam.derive();
    }
    return out->image_height == v.image_height ? 0 : -2;
}

// camera::derive() (= camera::initialize(), Camera.txt:136-175) for caller-supplied public camera fields: the
// image_height = int(image_width / aspect_ratio), "< 1 -> 1" rule (Camera.txt:137-138) and the viewport vectors,
// exactly as camera::render() computes them.  For tests of that rule over arbitrary widths and aspect ratios.
int rtkh_camera_derive(int image_width, double aspect_ratio, int samples_per_pixel, int max_depth, double vfov, const double* lookfrom,
                       const double* lookat, const double* vup, double defocus_angle, double focus_dist, rtk_camera* out) {
    if (!lookfrom || !lookat || !vup || !out) return -1;
    camera cam;
    cam.image_width = image_width;
    cam.aspect_ratio = aspect_ratio;
    cam.samples_per_pixel = samples_per_pixel;
    cam.max_depth = max_depth;
    cam.vfov = vfov;
    cam.lookfrom = point3(lookfrom[0], lookfrom[1], lookfrom[2]);
    cam.lookat = point3(lookat[0], lookat[1], lookat[2]);
    cam.vup = vec3(vup[0], vup[1], vup[2]);
    cam.defocus_angle = defocus_angle;
    cam.focus_dist = focus_dist;
    *out = cam.derive();
    return 0;
}

// The texels rtw_image holds for an image file (PPM or baseline JPEG), i.e. what image_texture::value reads
// (texture.h:90-104).  Returns the byte count (width * height * 3) and fills `out` when it is large enough;
// -1 when the file cannot be loaded (rtw_stb_image.h:62: width() == 0).
int64_t rtkh_image_texels(const char* path, int* width, int* height, uint8_t* out, int64_t capacity) {
    rtw_image im;
    if (!path || !im.load(path)) return -1;
    if (width) *width = im.width();
    if (height) *height = im.height();
    const int64_t n = int64_t(im.data().size());
    if (out && capacity >= n) std::memcpy(out, im.data().data(), size_t(n));
    return n;
}

}  // extern "C"
