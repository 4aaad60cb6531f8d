// rtk_trace.h -- launch entry points of rtk_trace.hip for the C-ABI layer.
#ifndef RTK_TRACE_H
#define RTK_TRACE_H

#include <hip/hip_runtime.h>

#include "rtk_device_layout.h"

namespace rtk {

// Enqueue the render kernel for one rank's tiles.  `features` selects the
// kernel instantiation (kFeatLean or kFeatAll); `count` selects the
// work-counting instantiation (the full-feature kernel, or -- for a scene with a
// MIXED program -- the counting build of the F_F32_BOX kernel itself).  tile_counter is
// a device word the persistent waves pull tile indices from (zeroed on `stream`
// before the launch); d_cam points at the camera record in device memory.  `diag`
// bits 8..13 select scheduler thresholds for A/B runs from tools/ (images are unaffected).
template <typename real>
hipError_t launch_render(const SceneView<real>& sc, const CameraRec<real>* d_cam, const TileMap& tmap, uint32_t seed, uint32_t features, bool count,
                         bool allow_lds, uint32_t diag, void* partial, unsigned long long* counters, unsigned int* tile_counter,
                         const int32_t* tile_order, unsigned int* tile_cost, hipStream_t stream);

// cost[n] (segments per local tile, measured by the frame just rendered) -> order[n], most expensive first; clears cost.
hipError_t launch_tile_order(unsigned int* cost, int n, int32_t* order, hipStream_t stream);

// Partial sums [item][3][64] -> the row-major image (+ bytes) or this rank's compact tile buffer.  A frame rendered in several
// passes over consecutive chunk ranges carries its running sum in `acc` [local tile][3][64] (read unless first_pass, written
// unless last_pass); only the last pass writes the outputs.
template <typename real>
hipError_t launch_resolve(const void* partial, const TileMap& tmap, int width, int height, double samples_scale, void* out_linear, uint8_t* out_rgb8,
                          void* acc, bool first_pass, bool last_pass, hipStream_t stream);

// Known-answer helper: closest hit of the scene root for n caller-supplied rays (device buffers).
template <typename real>
hipError_t launch_debug_hit(const SceneView<real>& sc, int n, const double* d_rays, const uint32_t* d_keys, double* d_out, unsigned long long* d_draws,
                            hipStream_t stream);

// Known-answer helpers for the shading side: shade_surface, texture_value and begin_sample on caller-supplied inputs.
template <typename real>
hipError_t launch_debug_scatter(const SceneView<real>& sc, int n, const int32_t* d_mat, const double* d_ray, const double* d_rec, const uint32_t* d_keys, double* d_out,
                                unsigned long long* d_draws, hipStream_t stream);
template <typename real>
hipError_t launch_debug_texture(const SceneView<real>& sc, int n, const int32_t* d_tex, const double* d_uvp, double* d_out, unsigned long long* d_work, hipStream_t stream);
template <typename real>
hipError_t launch_debug_get_ray(const CameraRec<real>& cam, uint32_t seed, int n, const int32_t* d_ijs, double* d_out, unsigned long long* d_draws, hipStream_t stream);

template <typename real>
hipError_t launch_unpermute(const void* gathered, int width, int height, int n_ranks, long long tiles_per_rank, void* out_linear, uint8_t* out_rgb8,
                            hipStream_t stream);

// Symbol name of the render kernel launch_render would launch for these arguments (the one decision function serves both).
template <typename real>
const char* render_kernel_name(const SceneView<real>& sc, uint32_t features, bool count, bool allow_lds, uint32_t diag);

}  // namespace rtk

#endif  // RTK_TRACE_H
