// rtk_trace.h -- launch entry points of rtk_trace.hip for the C-ABI layer.
#ifndef RTK_TRACE_H
#define RTK_TRACE_H

#include <hip/hip_runtime.h>

#include "rtk_device_layout.h"

namespace rtk {

// Enqueue the render kernel for one rank's tiles.  `features` selects the
// kernel instantiation (kFeatLean or kFeatAll); `count` selects the
// work-counting instantiation (always the full-feature kernel).
template <typename real>
hipError_t launch_render(const SceneView<real>& sc, const CameraRec<real>& cam, const TileMap& tmap, uint32_t seed, uint32_t features, bool count,
                         void* out_linear, uint8_t* out_rgb8, unsigned long long* counters, hipStream_t stream);

template <typename real>
hipError_t launch_unpermute(const void* gathered, int width, int height, int n_ranks, long long tiles_per_rank, void* out_linear, uint8_t* out_rgb8,
                            hipStream_t stream);

const char* render_kernel_name(bool f64, uint32_t features, bool count);

}  // namespace rtk

#endif  // RTK_TRACE_H
