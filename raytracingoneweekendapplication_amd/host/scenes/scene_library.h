// scene_library.h -- the benchmark and test scenes, written against the
// reference's scene API *names only* (hittable_list, sphere, quad, box, triangle,
// translate, rotate_y, constant_medium, bvh_node, the materials and textures).
//
// This file deliberately includes nothing: the including translation unit
// chooses the API implementation.
//   - the product (host/rtk_scenes.cpp) includes the drop-in headers of this
//     package first  -> scenes for the MI355X kernels;
//   - oracle/ref_driver.cpp includes the reference's own headers from
//     /root/reference first -> the same scenes built from the reference's classes.
// That one source compiles against both is the drop-in claim of INTEGRATION.md,
// and it lets tests compare the two flattened scenes byte for byte.
//
// Scene definitions follow BASELINE.json `configs` / SURVEY.md 8(d).  Where a
// scene draws random numbers the draws are taken into named locals one per
// statement, so the draw order does not depend on the compiler's argument
// evaluation order.
#ifndef RTK_SCENE_LIBRARY_H
#define RTK_SCENE_LIBRARY_H

struct rtk_view {
    int image_width = 400, image_height = 225;
    int samples_per_pixel = 10, max_depth = 10;
    color background = color(0, 0, 0);
    double vfov = 90;
    point3 lookfrom = point3(0, 0, 0), lookat = point3(0, 0, -1);
    vec3 vup = vec3(0, 1, 0);
    double defocus_angle = 0, focus_dist = 10;
};

struct rtk_light_def {
    point3 position;
    color intensity;
    double size;
};

struct rtk_scene_def {
    hittable_list world;                 // already wrapped as main.cpp:442 does
    std::vector<rtk_light_def> lights;   // the `lights` argument of camera::render
    rtk_view view;
};

inline void rtk_wrap_in_bvh(rtk_scene_def& s) { s.world = hittable_list(make_shared<bvh_node>(s.world)); }  // main.cpp:442

// C1 -- three spheres + ground (RTIOW ch. 11 scene; 400x225x10, depth 10).
inline void rtk_scene_three_spheres(rtk_scene_def& s) {
    auto ground = make_shared<lambertian>(color(0.8, 0.8, 0.0));
    auto center = make_shared<lambertian>(color(0.1, 0.2, 0.5));
    auto left = make_shared<dielectric>(1.5);
    auto right = make_shared<metal>(color(0.8, 0.6, 0.2), 1.0);
    s.world.add(make_shared<sphere>(point3(0.0, -100.5, -1.0), 100.0, ground));
    s.world.add(make_shared<sphere>(point3(0.0, 0.0, -1.2), 0.5, center));
    s.world.add(make_shared<sphere>(point3(-1.0, 0.0, -1.0), 0.5, left));
    s.world.add(make_shared<sphere>(point3(1.0, 0.0, -1.0), 0.5, right));
    rtk_wrap_in_bvh(s);
    s.view.image_width = 400; s.view.image_height = 225;
    s.view.samples_per_pixel = 10; s.view.max_depth = 10;
    s.view.background = color(0.7, 0.8, 1.0);
    s.view.vfov = 90; s.view.lookfrom = point3(0, 0, 0); s.view.lookat = point3(0, 0, -1);
    s.view.focus_dist = 1.0; s.view.defocus_angle = 0;
}

// C2 -- RTIOW book-1 cover scene (1920x1080x100, depth 50).
inline void rtk_scene_book1_final(rtk_scene_def& s) {
    s.world.add(make_shared<sphere>(point3(0, -1000, 0), 1000, make_shared<lambertian>(color(0.5, 0.5, 0.5))));
    for (int a = -11; a < 11; a++) {
        for (int b = -11; b < 11; b++) {
            double choose_mat = random_double();
            double jx = random_double();
            double jz = random_double();
            point3 center(a + 0.9 * jx, 0.2, b + 0.9 * jz);
            if ((center - point3(4, 0.2, 0)).length() > 0.9) {
                shared_ptr<material> sphere_material;
                if (choose_mat < 0.8) {
                    color c1 = color::random();
                    color c2 = color::random();
                    sphere_material = make_shared<lambertian>(c1 * c2);
                } else if (choose_mat < 0.95) {
                    color albedo = color::random(0.5, 1);
                    double fuzz = random_double(0, 0.5);
                    sphere_material = make_shared<metal>(albedo, fuzz);
                } else {
                    sphere_material = make_shared<dielectric>(1.5);
                }
                s.world.add(make_shared<sphere>(center, 0.2, sphere_material));
            }
        }
    }
    s.world.add(make_shared<sphere>(point3(0, 1, 0), 1.0, make_shared<dielectric>(1.5)));
    s.world.add(make_shared<sphere>(point3(-4, 1, 0), 1.0, make_shared<lambertian>(color(0.4, 0.2, 0.1))));
    s.world.add(make_shared<sphere>(point3(4, 1, 0), 1.0, make_shared<metal>(color(0.7, 0.6, 0.5), 0.0)));
    rtk_wrap_in_bvh(s);
    s.view.image_width = 1920; s.view.image_height = 1080;
    s.view.samples_per_pixel = 100; s.view.max_depth = 50;
    s.view.background = color(0.70, 0.80, 1.00);
    s.view.vfov = 20; s.view.lookfrom = point3(13, 2, 3); s.view.lookat = point3(0, 0, 0);
    s.view.defocus_angle = 0.6; s.view.focus_dist = 10.0;
}

// C3 -- Cornell box with two rotated/translated boxes (main.cpp:208-243;
// 800x800x1000, depth 25).
inline void rtk_scene_cornell_box(rtk_scene_def& s) {
    auto red = make_shared<lambertian>(color(.65, .05, .05));
    auto white = make_shared<lambertian>(color(.73, .73, .73));
    auto green = make_shared<lambertian>(color(.12, .45, .15));
    auto light = make_shared<diffuse_light>(color(15, 15, 15));
    s.world.add(make_shared<quad>(point3(555, 0, 0), vec3(0, 555, 0), vec3(0, 0, 555), green));
    s.world.add(make_shared<quad>(point3(0, 0, 0), vec3(0, 555, 0), vec3(0, 0, 555), red));
    s.world.add(make_shared<quad>(point3(343, 554, 332), vec3(-130, 0, 0), vec3(0, 0, -105), light));
    s.world.add(make_shared<quad>(point3(0, 0, 0), vec3(555, 0, 0), vec3(0, 0, 555), white));
    s.world.add(make_shared<quad>(point3(555, 555, 555), vec3(-555, 0, 0), vec3(0, 0, -555), white));
    s.world.add(make_shared<quad>(point3(0, 0, 555), vec3(555, 0, 0), vec3(0, 555, 0), white));
    shared_ptr<hittable> tall = box(point3(0, 0, 0), point3(165, 330, 165), white);
    tall = make_shared<rotate_y>(tall, 15);
    tall = make_shared<translate>(tall, vec3(265, 0, 295));
    s.world.add(tall);
    shared_ptr<hittable> cube = box(point3(0, 0, 0), point3(165, 165, 165), white);
    cube = make_shared<rotate_y>(cube, -18);
    cube = make_shared<translate>(cube, vec3(130, 0, 65));
    s.world.add(cube);
    rtk_wrap_in_bvh(s);
    s.view.image_width = 800; s.view.image_height = 800;
    s.view.samples_per_pixel = 1000; s.view.max_depth = 25;
    s.view.background = color(0, 0, 0);
    s.view.vfov = 40; s.view.lookfrom = point3(278, 278, -800); s.view.lookat = point3(278, 278, 0);
    s.view.defocus_angle = 0; s.view.focus_dist = 10;
}

// Procedural stand-in for the reference's monkey.obj (968 triangles): a
// three-times subdivided icosahedron (1280 triangles) with a deterministic
// radial displacement and per-vertex UVs.  The reference asset cannot travel
// to the GPU box; the hot path is indifferent to which
// ~1k-triangle mesh it traverses.
inline void rtk_add_blob_mesh(hittable_list& world, shared_ptr<material> mat, point3 centre, double radius) {
    std::vector<vec3> verts;
    std::vector<int> tris;
    const double t = (1.0 + std::sqrt(5.0)) / 2.0;
    const double ico[12][3] = {{-1, t, 0}, {1, t, 0}, {-1, -t, 0}, {1, -t, 0}, {0, -1, t}, {0, 1, t},
                               {0, -1, -t}, {0, 1, -t}, {t, 0, -1}, {t, 0, 1}, {-t, 0, -1}, {-t, 0, 1}};
    const int faces[20][3] = {{0, 11, 5}, {0, 5, 1}, {0, 1, 7}, {0, 7, 10}, {0, 10, 11}, {1, 5, 9}, {5, 11, 4},
                              {11, 10, 2}, {10, 7, 6}, {7, 1, 8}, {3, 9, 4}, {3, 4, 2}, {3, 2, 6}, {3, 6, 8},
                              {3, 8, 9}, {4, 9, 5}, {2, 4, 11}, {6, 2, 10}, {8, 6, 7}, {9, 8, 1}};
    for (auto& v : ico) verts.push_back(unit_vector(vec3(v[0], v[1], v[2])));
    for (auto& f : faces) { tris.push_back(f[0]); tris.push_back(f[1]); tris.push_back(f[2]); }
    for (int level = 0; level < 3; level++) {
        std::vector<int> next;
        std::vector<long long> keys;
        std::vector<int> mids;
        auto midpoint = [&](int a, int b) {
            long long key = a < b ? (long long)a * 100000 + b : (long long)b * 100000 + a;
            for (size_t k = 0; k < keys.size(); k++)
                if (keys[k] == key) return mids[k];
            verts.push_back(unit_vector(0.5 * (verts[a] + verts[b])));
            keys.push_back(key);
            mids.push_back(int(verts.size()) - 1);
            return mids.back();
        };
        for (size_t k = 0; k < tris.size(); k += 3) {
            int a = tris[k], b = tris[k + 1], c = tris[k + 2];
            int ab = midpoint(a, b), bc = midpoint(b, c), ca = midpoint(c, a);
            const int sub[12] = {a, ab, ca, b, bc, ab, c, ca, bc, ab, bc, ca};
            next.insert(next.end(), sub, sub + 12);
        }
        tris.swap(next);
    }
    std::vector<vec3> pos(verts.size());
    std::vector<glm::vec2> uvs(verts.size());
    for (size_t k = 0; k < verts.size(); k++) {
        const vec3& n = verts[k];
        double bump = 1.0 + 0.18 * std::sin(5.0 * n.x()) * std::cos(4.0 * n.y()) + 0.10 * std::sin(7.0 * n.z() + 1.0);
        pos[k] = centre + (radius * bump) * n;
        uvs[k] = glm::vec2(float(0.5 + 0.5 * n.x()), float(0.5 + 0.5 * n.y()));
    }
    for (size_t k = 0; k < tris.size(); k += 3)
        world.add(make_shared<triangle>(pos[tris[k]], pos[tris[k + 1]], pos[tris[k + 2]], mat, uvs[tris[k]], uvs[tris[k + 1]], uvs[tris[k + 2]]));
}

// C4 -- triangle mesh + ground sphere + spherical light (1920x1080x256, depth 10).
inline void rtk_scene_mesh(rtk_scene_def& s) {
    auto grey = make_shared<lambertian>(color(0.5, 0.5, 0.5));
    auto clay = make_shared<lambertian>(color(0.7, 0.35, 0.2));
    auto lamp = make_shared<diffuse_light>(color(12, 12, 12));
    rtk_add_blob_mesh(s.world, clay, point3(0, 0, 0), 1.0);
    s.world.add(make_shared<sphere>(point3(0, -1001.3, 0), 1000, grey));
    s.world.add(make_shared<sphere>(point3(2.5, 4, 2), 1.5, lamp));
    rtk_wrap_in_bvh(s);
    s.view.image_width = 1920; s.view.image_height = 1080;
    s.view.samples_per_pixel = 256; s.view.max_depth = 10;
    s.view.background = color(0.05, 0.06, 0.10);
    s.view.vfov = 40; s.view.lookfrom = point3(1.5, 1, 4); s.view.lookat = point3(0, 0, 0);
    s.view.defocus_angle = 0; s.view.focus_dist = 10;
}

// C5 -- book-2 final scene as main.cpp:268-340 intends it (the `world`
// shadowing at main.cpp:288 removed).  earth_image names the
// texture file for the globe (a synthetic PPM in tests and benchmarks).
inline void rtk_scene_book2_final(rtk_scene_def& s, const char* earth_image) {
    hittable_list boxes1;
    auto ground = make_shared<lambertian>(color(0.48, 0.83, 0.53));
    const int boxes_per_side = 20;
    for (int i = 0; i < boxes_per_side; i++) {
        for (int j = 0; j < boxes_per_side; j++) {
            double w = 100.0;
            double x0 = -1000.0 + i * w;
            double z0 = -1000.0 + j * w;
            double y0 = 0.0;
            double x1 = x0 + w;
            double y1 = random_double(1, 101);
            double z1 = z0 + w;
            boxes1.add(box(point3(x0, y0, z0), point3(x1, y1, z1), ground));
        }
    }
    s.world.add(make_shared<bvh_node>(boxes1));

    auto light = make_shared<diffuse_light>(color(7, 7, 7));
    s.world.add(make_shared<quad>(point3(123, 554, 147), vec3(300, 0, 0), vec3(0, 0, 265), light));

    point3 center1(400, 400, 200);
    point3 center2 = center1 + vec3(30, 0, 0);
    s.world.add(make_shared<sphere>(center1, center2, 50, make_shared<lambertian>(color(0.7, 0.3, 0.1))));

    s.world.add(make_shared<sphere>(point3(260, 150, 45), 50, make_shared<dielectric>(1.5)));
    s.world.add(make_shared<sphere>(point3(0, 150, 145), 50, make_shared<metal>(color(0.8, 0.8, 0.9), 1.0)));

    auto boundary = make_shared<sphere>(point3(360, 150, 145), 70, make_shared<dielectric>(1.5));
    s.world.add(boundary);
    s.world.add(make_shared<constant_medium>(boundary, 0.2, color(0.2, 0.4, 0.9)));
    boundary = make_shared<sphere>(point3(0, 0, 0), 5000, make_shared<dielectric>(1.5));
    s.world.add(make_shared<constant_medium>(boundary, .0001, color(1, 1, 1)));

    auto emat = make_shared<lambertian>(make_shared<image_texture>(earth_image));
    s.world.add(make_shared<sphere>(point3(400, 200, 400), 100, emat));
    auto pertext = make_shared<noise_texture>(0.2);
    s.world.add(make_shared<sphere>(point3(220, 280, 300), 80, make_shared<lambertian>(pertext)));

    hittable_list boxes2;
    auto white = make_shared<lambertian>(color(.73, .73, .73));
    for (int j = 0; j < 1000; j++) {
        point3 c = point3::random(0, 165);
        boxes2.add(make_shared<sphere>(c, 10, white));
    }
    s.world.add(make_shared<translate>(make_shared<rotate_y>(make_shared<bvh_node>(boxes2), 15), vec3(-100, 270, 395)));
    rtk_wrap_in_bvh(s);

    s.view.image_width = 1920; s.view.image_height = 1080;
    s.view.samples_per_pixel = 1000; s.view.max_depth = 10;
    s.view.background = color(0, 0, 0);
    s.view.vfov = 40; s.view.lookfrom = point3(478, 278, -600); s.view.lookat = point3(278, 278, 0);
    s.view.defocus_angle = 0; s.view.focus_dist = 10;
}

// C4 with a real asset: an OBJ file loaded through mesh::loadObj (mesh.h:22) under a rotate/scale/translate
// transform, as main.cpp:399-412 does for its (missing) corgi model.  `obj_file` is e.g. the reference's
// monkey.obj (968 triangles) where that file is available; tests also use a small hand-written OBJ.
inline void rtk_scene_obj_mesh(rtk_scene_def& s, const char* obj_file) {
    auto grey = make_shared<lambertian>(color(0.5, 0.5, 0.5));
    auto clay = make_shared<lambertian>(color(0.7, 0.35, 0.2));
    glm::mat4 transform = glm::mat4(1.0f);
    transform = glm::translate(transform, glm::vec3(0.25f, 0.1f, -0.5f));
    transform = glm::rotate(transform, glm::radians(30.0f), glm::vec3(0, 1, 0));
    transform = glm::scale(transform, glm::vec3(1.2f, 1.2f, 1.2f));
    mesh model;
    if (!model.loadObj(obj_file, s.world, clay, transform)) s.world.add(make_shared<sphere>(point3(0, 0, 0), 1.0, clay));
    s.world.add(make_shared<sphere>(point3(0, -1001.3, 0), 1000, grey));
    s.world.add(make_shared<sphere>(point3(2.5, 4, 2), 1.5, make_shared<diffuse_light>(color(12, 12, 12))));
    rtk_wrap_in_bvh(s);
    s.view.image_width = 1920; s.view.image_height = 1080;
    s.view.samples_per_pixel = 256; s.view.max_depth = 10;
    s.view.background = color(0.05, 0.06, 0.10);
    s.view.vfov = 40; s.view.lookfrom = point3(1.5, 1, 4); s.view.lookat = point3(0, 0, 0);
    s.view.defocus_angle = 0; s.view.focus_dist = 10;
}

// ---- parity-test scenes (small, cover what the configs do not) -------------

// Every material and every texture kind, triangles with UVs, a point light pair.
inline void rtk_scene_material_zoo(rtk_scene_def& s, const char* image_file) {
    auto checker = make_shared<checker_texture>(0.32, color(.2, .3, .1), color(.9, .9, .9));
    s.world.add(make_shared<sphere>(point3(0, -1000, 0), 1000, make_shared<lambertian>(checker)));
    s.world.add(make_shared<sphere>(point3(2, 1, 5), 1.0, make_shared<dielectric>(1.5)));
    s.world.add(make_shared<sphere>(point3(2, 1, 5), 0.8, make_shared<dielectric>(1.0 / 1.5)));
    s.world.add(make_shared<sphere>(point3(-2, 1, 5), 1.0, make_shared<lambertian>(make_shared<noise_texture>(4))));
    auto checkerT = make_shared<checker_texture_triangle>(0.5, color(0, 0, 0), color(.9, .9, .9));
    s.world.add(make_shared<triangle>(point3(4, 0, 8), point3(-4, 0, 8), point3(0, 6, 8), make_shared<lambertian>(checkerT)));
    s.world.add(make_shared<triangle>(point3(-5, 0, 7), point3(-3, 0, 9), point3(-4, 3, 8), make_shared<metal>(color(.8, .8, .9), 0.1),
                                      glm::vec2(0.1f, 0.2f), glm::vec2(1.4f, 0.3f), glm::vec2(0.5f, 2.2f)));
    s.world.add(make_shared<sphere>(point3(0, 1, 5), 1.0, make_shared<lambertian>(make_shared<image_texture>(image_file))));
    s.world.add(make_shared<sphere>(point3(0, 3.2, 5), 0.7, make_shared<specular>(color(1.0, 0.1, 0.1), 5)));
    s.world.add(make_shared<sphere>(point3(4.2, 0.8, 4), 0.8, make_shared<metal>(color(0.7, 0.6, 0.5), 0.0)));
    s.world.add(make_shared<sphere>(point3(-4.2, 0.8, 4), 0.8, make_shared<emissive_light>(color(4, 3, 2))));
    auto nested = make_shared<checker_texture>(1.5, checker, make_shared<solid_color>(0.1, 0.1, 0.8));
    s.world.add(make_shared<quad>(point3(-6, 0, 9), vec3(12, 0, 0), vec3(0, 5, 0), make_shared<lambertian>(nested)));
    s.world.add(make_shared<quad>(point3(5, 4, 3), vec3(1, 0, 0), vec3(0, 0, 1), make_shared<diffuse_light>(color(10, 10, 10))));
    rtk_wrap_in_bvh(s);
    s.lights.push_back(rtk_light_def{point3(0, 6, 2), color(3, 3, 3), 0.5});
    s.lights.push_back(rtk_light_def{point3(-3, 1.5, 4.5), color(1, 2, 1), 2.0});
    s.view.image_width = 96; s.view.image_height = 54;
    s.view.samples_per_pixel = 8; s.view.max_depth = 12;
    s.view.background = color(0.3, 0.4, 0.6);
    s.view.vfov = 35; s.view.lookfrom = point3(1, 4, -10); s.view.lookat = point3(0, 1, 5);
    s.view.defocus_angle = 0.3; s.view.focus_dist = 15.5;
}

// Cornell box with smoke boxes (main.cpp:342-380): constant_medium around
// translate(rotate_y(box)) -- media over instance transforms over lists.
inline void rtk_scene_cornell_smoke(rtk_scene_def& s) {
    auto red = make_shared<lambertian>(color(.65, .05, .05));
    auto white = make_shared<lambertian>(color(.73, .73, .73));
    auto green = make_shared<lambertian>(color(.12, .45, .15));
    auto light = make_shared<diffuse_light>(color(7, 7, 7));
    s.world.add(make_shared<quad>(point3(555, 0, 0), vec3(0, 555, 0), vec3(0, 0, 555), green));
    s.world.add(make_shared<quad>(point3(0, 0, 0), vec3(0, 555, 0), vec3(0, 0, 555), red));
    s.world.add(make_shared<quad>(point3(113, 554, 127), vec3(330, 0, 0), vec3(0, 0, 305), light));
    s.world.add(make_shared<quad>(point3(0, 555, 0), vec3(555, 0, 0), vec3(0, 0, 555), white));
    s.world.add(make_shared<quad>(point3(0, 0, 0), vec3(555, 0, 0), vec3(0, 0, 555), white));
    s.world.add(make_shared<quad>(point3(0, 0, 555), vec3(555, 0, 0), vec3(0, 555, 0), white));
    shared_ptr<hittable> box1 = box(point3(0, 0, 0), point3(165, 330, 165), white);
    box1 = make_shared<rotate_y>(box1, 15);
    box1 = make_shared<translate>(box1, vec3(265, 0, 295));
    shared_ptr<hittable> box2 = box(point3(0, 0, 0), point3(165, 165, 165), white);
    box2 = make_shared<rotate_y>(box2, -18);
    box2 = make_shared<translate>(box2, vec3(130, 0, 65));
    s.world.add(make_shared<constant_medium>(box1, 0.01, color(0, 0, 0)));
    s.world.add(make_shared<constant_medium>(box2, 0.01, color(1, 1, 1)));
    rtk_wrap_in_bvh(s);
    s.view.image_width = 64; s.view.image_height = 64;
    s.view.samples_per_pixel = 16; s.view.max_depth = 10;
    s.view.background = color(0, 0, 0);
    s.view.vfov = 40; s.view.lookfrom = point3(278, 278, -800); s.view.lookat = point3(278, 278, 0);
    s.view.defocus_angle = 0; s.view.focus_dist = 10;
}

// A single object in the world: the top-level bvh_node has a span of one, so
// the object is tested twice per ray (bvh.h:30-32), here a fog ball -- the
// double RNG draw of SURVEY Q7 in its smallest form.  Plus motion blur.
inline void rtk_scene_single_fog(rtk_scene_def& s) {
    auto shell = make_shared<sphere>(point3(0, 0, -3), point3(0.4, 0.2, -3), 1.0, make_shared<dielectric>(1.5));
    s.world.add(make_shared<constant_medium>(shell, 0.8, color(0.9, 0.5, 0.2)));
    rtk_wrap_in_bvh(s);
    s.view.image_width = 48; s.view.image_height = 32;
    s.view.samples_per_pixel = 16; s.view.max_depth = 8;
    s.view.background = color(0.6, 0.7, 0.9);
    s.view.vfov = 50; s.view.lookfrom = point3(0, 0, 1); s.view.lookat = point3(0, 0, -3);
    s.view.defocus_angle = 0; s.view.focus_dist = 4;
}

inline bool rtk_build_named_scene(const char* name, const char* image_file, rtk_scene_def& s) {
    std::string n(name);
    if (n == "three_spheres") rtk_scene_three_spheres(s);
    else if (n == "book1_final") rtk_scene_book1_final(s);
    else if (n == "cornell_box") rtk_scene_cornell_box(s);
    else if (n == "mesh") rtk_scene_mesh(s);
    else if (n == "book2_final") rtk_scene_book2_final(s, image_file);
    else if (n == "material_zoo") rtk_scene_material_zoo(s, image_file);
    else if (n == "cornell_smoke") rtk_scene_cornell_smoke(s);
    else if (n == "single_fog") rtk_scene_single_fog(s);
    else if (n == "obj_mesh") rtk_scene_obj_mesh(s, image_file);  // the file argument is the OBJ path here
    else return false;
    return true;
}

#endif  // RTK_SCENE_LIBRARY_H
