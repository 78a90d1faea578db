// srt_launch.h -- host-callable launchers of the gfx950 kernels.
#ifndef SRT_LAUNCH_H
#define SRT_LAUNCH_H

#include <hip/hip_runtime_api.h>

#include "srt_core.h"

namespace srt {

// stats: 8 x uint64 on the device: rays_path, rays_shadow, hit_nodes, box_tests, sphere_tests,
// tri_tests, rect_tests, (unused).  The kernel adds to them; the caller zeroes them.
hipError_t launch_trace(const SceneView& sc, const RenderArgs& ra, bool exact, bool counted, uint32_t* d_out,
                        unsigned long long* d_stats, hipStream_t stream);
hipError_t launch_assemble(const RenderArgs& ra, const uint32_t* d_gathered, size_t stride, uint32_t* d_frame, hipStream_t stream);

}  // namespace srt
#endif
