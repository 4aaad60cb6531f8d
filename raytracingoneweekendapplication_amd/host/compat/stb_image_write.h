// compat/stb_image_write.h -- main.cpp includes "stb_image_write.h" with STB_IMAGE_WRITE_IMPLEMENTATION
// (main.cpp:25-27).  The drop-in camera writes its PNG itself (rtk::write_png); this header only provides
// the one entry point by that name for code that calls it directly.
#pragma once
#include "../rtk_camera.h"

inline int stbi_write_png(const char* filename, int w, int h, int comp, const void* data, int stride_in_bytes) {
    if (comp != 3 || stride_in_bytes != w * 3) return 0;
    return rtk::write_png(filename, w, h, static_cast<const uint8_t*>(data)) ? 1 : 0;
}
