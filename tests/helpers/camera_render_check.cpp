// GPU test helper: the drop-in C++ path end to end -- scene through the reference-style API, camera::render_to in the
// reference order, in the automatic order and in the forced fast order -- and a one-line JSON verdict on stdout.
#include "rtweekend.h"

#include "bvh.h"
#include "camera.h"
#include "constant_medium.h"
#include "hittable_list.h"
#include "material.h"
#include "quad.h"
#include "sphere.h"

#include <cstring>

struct progress_log {
    int calls = 0;
    bool monotone = true;
    int64_t last = -1, total = 0;
};
static void on_progress(int64_t done, int64_t total, void* user) {
    auto* log = static_cast<progress_log*>(user);
    log->calls++;
    if (done < log->last || done > total) log->monotone = false;
    log->last = done;
    log->total = total;
}

static hittable_list spheres_scene(bool with_fog) {
    hittable_list world;
    world.add(make_shared<sphere>(point3(0, -1000, 0), 1000, make_shared<lambertian>(color(0.5, 0.5, 0.5))));
    for (int a = -4; a < 4; a++)
        for (int b = -4; b < 4; b++) {
            point3 centre(a + 0.9 * random_double(), 0.2, b + 0.9 * random_double());
            double pick = random_double();
            shared_ptr<material> m;
            if (pick < 0.6) m = make_shared<lambertian>(color(random_double(), random_double(), random_double()));
            else if (pick < 0.85) m = make_shared<metal>(color(0.7, 0.6, 0.5), 0.3 * random_double());
            else m = make_shared<dielectric>(1.5);
            world.add(make_shared<sphere>(centre, 0.2, m));
        }
    world.add(make_shared<sphere>(point3(0, 1, 0), 1.0, make_shared<dielectric>(1.5)));
    world.add(make_shared<quad>(point3(-2, 3, -2), vec3(4, 0, 0), vec3(0, 0, 4), make_shared<diffuse_light>(color(3, 3, 3))));
    if (with_fog) world.add(make_shared<constant_medium>(make_shared<sphere>(point3(2, 0.7, 1), 0.7, make_shared<dielectric>(1.5)), 1.2, color(.9, .9, 1)));
    return hittable_list(make_shared<bvh_node>(world));
}

int main() {
    rtk::seed_scene_rng(1234);
    std::vector<point_light> lights;
    camera cam;
    cam.image_width = 160;
    cam.aspect_ratio = 16.0 / 9.0;
    cam.samples_per_pixel = 8;
    cam.max_depth = 12;
    cam.background = color(0.7, 0.8, 1.0);
    cam.vfov = 30;
    cam.lookfrom = point3(9, 2.5, 4);
    cam.lookat = point3(0, 0.5, 0);
    cam.write_image = false;
    bool ok = true;
    std::printf("{");
    for (int fog = 0; fog < 2; fog++) {
        hittable_list world = spheres_scene(fog != 0);
        std::vector<double> ref, aut, fast;
        cam.order = camera::reference_order;
        int rc0 = cam.render_to(world, lights, &ref, nullptr);
        cam.order = camera::auto_order;
        int rc1 = cam.render_to(world, lights, &aut, nullptr);
        const bool auto_used_fast = cam.used_fast_order, exact = cam.fast_order_exact;
        cam.order = camera::fast_order;
        int rc2 = cam.render_to(world, lights, &fast, nullptr);
        // media re-grouped too: the forced fast order is then another image of the same estimator, and auto_order declines
        std::vector<double> free_fast, free_auto;
        cam.free_media_order = true;
        int rc6 = cam.render_to(world, lights, &free_fast, nullptr);
        cam.order = camera::auto_order;
        int rc7 = cam.render_to(world, lights, &free_auto, nullptr);
        const bool free_auto_used_fast = cam.used_fast_order, free_exact = cam.fast_order_exact;
        cam.free_media_order = false;
        ok = ok && rc6 == 0 && rc7 == 0;
        // camera::devices: render() owns the split over several GPUs (here the same GPU listed twice and three times:
        // two / three ranks, tile buffers, one gather, un-permute) -- the image must be the very same doubles and bytes
        std::vector<double> two, three;
        std::vector<uint8_t> bytes1, bytes3;
        cam.order = camera::reference_order;
        progress_log log;
        cam.progress = &on_progress;
        cam.progress_user = &log;
        cam.devices = {0, 0};
        int rc3 = cam.render_to(world, lights, &two, nullptr);
        const bool first_monotone = log.monotone && log.total > 0 && log.last == log.total;
        log = progress_log();   // each render reports from zero to its own total
        cam.devices = {0, 0, 0};
        cam.order = camera::auto_order;
        int rc4 = cam.render_to(world, lights, &three, &bytes3);
        cam.devices.clear();
        cam.progress = nullptr;
        int rc5 = cam.render_to(world, lights, nullptr, &bytes1);
        const bool same_two = ref.size() == two.size() && std::memcmp(ref.data(), two.data(), ref.size() * sizeof(double)) == 0;
        const bool same_three = ref.size() == three.size() && std::memcmp(ref.data(), three.data(), ref.size() * sizeof(double)) == 0;
        const bool same_bytes = bytes1.size() == bytes3.size() && !bytes1.empty() && std::memcmp(bytes1.data(), bytes3.data(), bytes1.size()) == 0;
        ok = ok && rc3 == 0 && rc4 == 0 && rc5 == 0;
        const bool same_auto = ref.size() == aut.size() && std::memcmp(ref.data(), aut.data(), ref.size() * sizeof(double)) == 0;
        const bool same_fast = ref.size() == fast.size() && std::memcmp(ref.data(), fast.data(), ref.size() * sizeof(double)) == 0;
        const bool same_free_fast = ref.size() == free_fast.size() && std::memcmp(ref.data(), free_fast.data(), ref.size() * sizeof(double)) == 0;
        const bool same_free_auto = ref.size() == free_auto.size() && std::memcmp(ref.data(), free_auto.data(), ref.size() * sizeof(double)) == 0;
        double mean_ref = 0, mean_fast = 0, mean_free = 0;
        for (double v : ref) mean_ref += v;
        for (double v : fast) mean_fast += v;
        for (double v : free_fast) mean_free += v;
        std::printf("%s\"fog%d\": {\"rc\": [%d, %d, %d], \"exact\": %s, \"auto_used_fast\": %s, \"auto_identical\": %s, \"fast_identical\": %s, "
                    "\"mean_ref\": %.6f, \"mean_fast\": %.6f, \"two_devices_identical\": %s, \"three_devices_identical\": %s, \"bytes_identical\": %s, "
                    "\"free_exact\": %s, \"free_auto_used_fast\": %s, \"free_auto_identical\": %s, \"free_fast_identical\": %s, \"mean_free\": %.6f, "
                    "\"progress_calls\": %d, \"progress_monotone\": %s, \"progress_reached_total\": %s}",
                    fog ? ", " : "", fog, rc0, rc1, rc2, exact ? "true" : "false", auto_used_fast ? "true" : "false", same_auto ? "true" : "false",
                    same_fast ? "true" : "false", mean_ref / ref.size(), mean_fast / fast.size(), same_two ? "true" : "false", same_three ? "true" : "false",
                    same_bytes ? "true" : "false", free_exact ? "true" : "false", free_auto_used_fast ? "true" : "false", same_free_auto ? "true" : "false",
                    same_free_fast ? "true" : "false", mean_free / ref.size(), log.calls, (log.monotone && first_monotone) ? "true" : "false", (log.total > 0 && log.last == log.total) ? "true" : "false");
        ok = ok && rc0 == 0 && rc1 == 0 && rc2 == 0;
    }
    std::printf("}\n");
    return ok ? 0 : 1;
}
