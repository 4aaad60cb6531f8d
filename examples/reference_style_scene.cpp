// A reference-style program (the shape of main.cpp:114-487 without its Win32 shell): scene setup with the
// drop-in API, then camera::render() -- which runs the sample loop on the MI355X through include/rtk.h.
// Build: see INTEGRATION.md section 2.
#include "rtweekend.h"

#include "bvh.h"
#include "camera.h"
#include "constant_medium.h"
#include "hittable.h"
#include "hittable_list.h"
#include "material.h"
#include "quad.h"
#include "sphere.h"
#include "texture.h"
#include "triangle.h"

int main(int argc, char** argv) {
    hittable_list world;
    std::vector<point_light> lights;

    auto checker = make_shared<checker_texture>(0.32, color(.2, .3, .1), color(.9, .9, .9));
    world.add(make_shared<sphere>(point3(0, -1000, 0), 1000, make_shared<lambertian>(checker)));
    world.add(make_shared<sphere>(point3(0, 1, 0), 1.0, make_shared<dielectric>(1.5)));
    world.add(make_shared<sphere>(point3(-4, 1, 0), 1.0, make_shared<lambertian>(make_shared<noise_texture>(4))));
    world.add(make_shared<sphere>(point3(4, 1, 0), 1.0, make_shared<metal>(color(0.7, 0.6, 0.5), 0.0)));
    world.add(make_shared<quad>(point3(-2, 3, -2), vec3(4, 0, 0), vec3(0, 0, 4), make_shared<diffuse_light>(color(4, 4, 4))));
    shared_ptr<hittable> crate = box(point3(0, 0, 0), point3(1, 1, 1), make_shared<lambertian>(color(.7, .3, .2)));
    crate = make_shared<translate>(make_shared<rotate_y>(crate, 30), vec3(1.5, 0, 2));
    world.add(crate);
    world.add(make_shared<constant_medium>(make_shared<sphere>(point3(-1.5, 0.6, 2.5), 0.6, make_shared<dielectric>(1.5)), 1.5, color(.9, .9, 1)));

    world = hittable_list(make_shared<bvh_node>(world));  // main.cpp:442

    camera cam;
    cam.image_width = argc > 1 ? std::atoi(argv[1]) : 640;
    cam.aspect_ratio = 16.0 / 9.0;
    cam.samples_per_pixel = argc > 2 ? std::atoi(argv[2]) : 64;
    cam.max_depth = 20;
    cam.background = color(0.7, 0.8, 1.0);
    cam.vfov = 25;
    cam.lookfrom = point3(13, 3, 4);
    cam.lookat = point3(0, 0.8, 0.5);
    cam.defocus_angle = 0.4;
    cam.focus_dist = 13.0;
    cam.image_name = "reference_style_scene.png";
    cam.render(world, lights);  // Camera.txt:54
    return 0;
}
