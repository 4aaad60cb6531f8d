// CPU sanitizer harness (tests/test_sanitizers.py builds it with -fsanitize=address,undefined): every parser of
// untrusted input the host side has, driven over files named on the command line.
//
//   parser_harness <mode> <file>...      mode = image | obj | rtks
//
//   image  rtw_image::load            host/rtk_jpeg.h, host/rtk_png.h, the PPM reader (what image_texture(const char*) runs)
//   obj    mesh::loadObj              host/mesh.h
//   rtks   rtk::desc_storage::load    host/rtk_desc_io.h, then rtk_scene_validate (csrc/rtk_api.cpp: the validator and the
//                                     program compiler rtk_scene_upload runs) and rtk_scene_optimize (csrc/rtk_optimize.cpp)
//
// Prints one line per file; a malformed file must end in "rejected" (or load as something harmless), never in a
// sanitizer report or a crash.  No device is touched: rtk_scene_validate / rtk_scene_optimize are host-only
// (tests/helpers/launch_stubs.cpp satisfies the linker for the kernels' launchers, which are never reached).
#include <cstdio>
#include <cstring>
#include <string>

#include "rtweekend.h"

#include "hittable_list.h"
#include "material.h"
#include "mesh.h"
#include "rtk_desc_io.h"
#include "rtw_stb_image.h"

static unsigned long long fnv(const unsigned char* p, size_t n) {
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}

namespace rtk {
size_t hot_program_lds_bytes(const rtk_scene_desc* scene);  // csrc/rtk_api.cpp (internal; what rtk_scene_optimize sizes the LDS part with)
}

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s image|obj|rtks <file>...\n", argv[0]);
        return 2;
    }
    const std::string mode = argv[1];
    for (int k = 2; k < argc; k++) {
        const char* path = argv[k];
        if (mode == "image") {
            rtw_image im;
            if (im.load(path)) std::printf("%s: image %dx%d fnv %016llx\n", path, im.width(), im.height(), fnv(im.data().data(), im.data().size()));
            else std::printf("%s: rejected\n", path);
        } else if (mode == "obj") {
            hittable_list world;
            mesh m;
            auto mat = make_shared<lambertian>(color(0.5, 0.5, 0.5));
            const bool ok = m.loadObj(path, world, mat, glm::mat4(1.0f));
            std::printf("%s: %s, %zu triangles\n", path, ok ? "read" : "rejected", world.objects.size());
        } else if (mode == "rtks") {
            rtk::desc_storage st;
            if (!st.load(path)) {
                std::printf("%s: rejected (file)\n", path);
                continue;
            }
            int32_t ops = 0;
            const int rc = rtk_scene_validate(&st.desc, &ops);
            int rc_opt = 1;
            if (rc == RTK_OK) {  // only descriptions the validator accepts reach the optimiser (rtk_scene_upload_fast's order)
                rtk_scene_desc* fast = nullptr;
                rtk_optimize_info info;
                rc_opt = rtk_scene_optimize(&st.desc, nullptr, &fast, &info);
                if (rc_opt == RTK_OK) {
                    int32_t fast_ops = 0;
                    rc_opt = rtk_scene_validate(fast, &fast_ops);
                    (void)rtk::hot_program_lds_bytes(fast);  // the hot/cold program builder (build_hot_cold_program) on every accepted scene
                    rtk_scene_optimized_free(fast);
                }
                (void)rtk::hot_program_lds_bytes(&st.desc);
            }
            std::printf("%s: validate %d (%d ops) optimize %d\n", path, rc, ops, rc_opt);
        } else {
            return 2;
        }
    }
    return 0;
}
