// Link-time stand-ins for the launch entry points of csrc/rtk_trace.hip, so that csrc/rtk_api.cpp can be linked into
// the CPU sanitizer harness (tests/helpers/parser_harness.cpp) without device code.  Test infrastructure only; the
// harness calls host-only entry points (rtk_scene_validate, rtk_scene_optimize) and never reaches these.
#include "rtk_trace.h"

namespace rtk {  // link-time stand-ins for the kernels' launchers (test harness only; never called)
template <typename real>
hipError_t launch_render(const SceneView<real>&, const CameraRec<real>*, const TileMap&, uint32_t, uint32_t, bool, bool, uint32_t, void*, unsigned long long*,
                         unsigned int*, const int32_t*, unsigned int*, hipStream_t) { return hipErrorNotSupported; }
template hipError_t launch_render<double>(const SceneView<double>&, const CameraRec<double>*, const TileMap&, uint32_t, uint32_t, bool, bool, uint32_t, void*,
                                          unsigned long long*, unsigned int*, const int32_t*, unsigned int*, hipStream_t);
template hipError_t launch_render<float>(const SceneView<float>&, const CameraRec<float>*, const TileMap&, uint32_t, uint32_t, bool, bool, uint32_t, void*,
                                         unsigned long long*, unsigned int*, const int32_t*, unsigned int*, hipStream_t);
hipError_t launch_tile_order(unsigned int*, int, int32_t*, hipStream_t) { return hipErrorNotSupported; }
template <typename real>
hipError_t launch_resolve(const void*, const TileMap&, int, int, double, void*, uint8_t*, void*, bool, bool, hipStream_t) { return hipErrorNotSupported; }
template hipError_t launch_resolve<double>(const void*, const TileMap&, int, int, double, void*, uint8_t*, void*, bool, bool, hipStream_t);
template hipError_t launch_resolve<float>(const void*, const TileMap&, int, int, double, void*, uint8_t*, void*, bool, bool, hipStream_t);
template <typename real>
hipError_t launch_debug_hit(const SceneView<real>&, int, const double*, const uint32_t*, double*, unsigned long long*, hipStream_t) { return hipErrorNotSupported; }
template hipError_t launch_debug_hit<double>(const SceneView<double>&, int, const double*, const uint32_t*, double*, unsigned long long*, hipStream_t);
template hipError_t launch_debug_hit<float>(const SceneView<float>&, int, const double*, const uint32_t*, double*, unsigned long long*, hipStream_t);
template <typename real>
hipError_t launch_debug_scatter(const SceneView<real>&, int, const int32_t*, const double*, const double*, const uint32_t*, double*, unsigned long long*, hipStream_t) { return hipErrorNotSupported; }
template hipError_t launch_debug_scatter<double>(const SceneView<double>&, int, const int32_t*, const double*, const double*, const uint32_t*, double*, unsigned long long*, hipStream_t);
template hipError_t launch_debug_scatter<float>(const SceneView<float>&, int, const int32_t*, const double*, const double*, const uint32_t*, double*, unsigned long long*, hipStream_t);
template <typename real>
hipError_t launch_debug_texture(const SceneView<real>&, int, const int32_t*, const double*, double*, unsigned long long*, hipStream_t) { return hipErrorNotSupported; }
template hipError_t launch_debug_texture<double>(const SceneView<double>&, int, const int32_t*, const double*, double*, unsigned long long*, hipStream_t);
template hipError_t launch_debug_texture<float>(const SceneView<float>&, int, const int32_t*, const double*, double*, unsigned long long*, hipStream_t);
template <typename real>
hipError_t launch_debug_get_ray(const CameraRec<real>&, uint32_t, int, const int32_t*, double*, unsigned long long*, hipStream_t) { return hipErrorNotSupported; }
template hipError_t launch_debug_get_ray<double>(const CameraRec<double>&, uint32_t, int, const int32_t*, double*, unsigned long long*, hipStream_t);
template hipError_t launch_debug_get_ray<float>(const CameraRec<float>&, uint32_t, int, const int32_t*, double*, unsigned long long*, hipStream_t);
template <typename real>
hipError_t launch_unpermute(const void*, int, int, int, long long, void*, uint8_t*, hipStream_t) { return hipErrorNotSupported; }
template hipError_t launch_unpermute<double>(const void*, int, int, int, long long, void*, uint8_t*, hipStream_t);
template hipError_t launch_unpermute<float>(const void*, int, int, int, long long, void*, uint8_t*, hipStream_t);
template <typename real>
const char* render_kernel_name(const SceneView<real>&, uint32_t, bool, bool, uint32_t) { return ""; }
template const char* render_kernel_name<double>(const SceneView<double>&, uint32_t, bool, bool, uint32_t);
template const char* render_kernel_name<float>(const SceneView<float>&, uint32_t, bool, bool, uint32_t);
}  // namespace rtk

