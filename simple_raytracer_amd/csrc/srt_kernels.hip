// srt_kernels.hip -- gfx950 kernels: the trace megakernel and the tile assembly kernel.
//
// One lane owns one pixel at a time.  A lane is a small state machine (srt_core.h, advance())
// that always has exactly one ray query pending -- the closest-hit query of a ray-tree node
// or the shadow query of one of its lights -- so the 64 lanes of a wavefront run the BVH
// traversal loop together whatever stage of their ray trees they are in.  The traversal
// stack is a per-wavefront slab of LDS with one column per lane (entry i of lane l at
// [i][l]: conflict-free), the ray-tree frames (one per pending recursion level) are private.
#include <hip/hip_runtime.h>

#include "srt_core.h"
#include "srt_launch.h"

namespace srt {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <bool EXACT, bool COUNT>
__global__ void __launch_bounds__(kBlock) trace_kernel(SceneView sc, RenderArgs ra, uint32_t* __restrict__ out, unsigned long long* __restrict__ stats)
{
    __shared__ uint32_t lds_stack[kWaves][STACK_DEPTH][64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* my_stack = &lds_stack[wave][0][lane];

    PathState st;
    Frame frames[FRAME_SLOTS];
    Counters cnt;
    cnt.box_tests = cnt.sphere_tests = cnt.tri_tests = cnt.rect_tests = 0u;
    st.rays_path = st.rays_shadow = st.hit_nodes = 0u;

    const uint32_t work = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
    bool alive = false;
    if (work < ra.n_work) {
        int32_t x, y;
        uint32_t oi;
        if (work_to_pixel(ra, work, x, y, oi)) {
            start_pixel(st, ra, oi, x, y);
            alive = true;
        } else if (ra.layout != 0) {
            out[oi] = 0u;  // padding pixel of an edge tile
        }
    }
    while (alive) {
        Hit h = closest_hit<EXACT, COUNT, 64>(sc, st.qo, st.qd, st.qtmin, my_stack, cnt);
        if (advance(st, h, sc, ra, frames)) {
            out[st.work] = finish_pixel(st);
            alive = false;
        }
    }

    unsigned long long v0 = wave_sum(st.rays_path), v1 = wave_sum(st.rays_shadow), v2 = wave_sum(st.hit_nodes);
    if (lane == 0u) {
        atomicAdd(&stats[0], v0);
        atomicAdd(&stats[1], v1);
        atomicAdd(&stats[2], v2);
    }
    if (COUNT) {
        unsigned long long c0 = wave_sum(cnt.box_tests), c1 = wave_sum(cnt.sphere_tests), c2 = wave_sum(cnt.tri_tests), c3 = wave_sum(cnt.rect_tests);
        if (lane == 0u) {
            atomicAdd(&stats[3], c0);
            atomicAdd(&stats[4], c1);
            atomicAdd(&stats[5], c2);
            atomicAdd(&stats[6], c3);
        }
    }
}

// Rank 0 of a multi-GPU job: pixel (x, y) of the frame lives in tile t = tx + ty*tiles_x, which
// rank t % world rendered as its (t / world)-th tile.
__global__ void __launch_bounds__(kBlock) assemble_kernel(RenderArgs ra, const uint32_t* __restrict__ gathered, size_t stride, uint32_t* __restrict__ frame)
{
    uint32_t i = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
    uint32_t total = (uint32_t)ra.width * (uint32_t)ra.height;
    if (i >= total) return;
    uint32_t y = i / (uint32_t)ra.width, x = i - y * (uint32_t)ra.width;
    uint32_t tx = x / (uint32_t)ra.tile_w, ty = y / (uint32_t)ra.tile_h;
    uint32_t lx = x - tx * (uint32_t)ra.tile_w, ly = y - ty * (uint32_t)ra.tile_h;
    uint32_t t = tx + ty * (uint32_t)ra.tiles_x;
    uint32_t r = t % (uint32_t)ra.world_size, j = t / (uint32_t)ra.world_size;
    frame[i] = gathered[(size_t)r * stride + (size_t)j * (uint32_t)(ra.tile_w * ra.tile_h) + ly * (uint32_t)ra.tile_w + lx];
}

hipError_t launch_trace(const SceneView& sc, const RenderArgs& ra, bool exact, bool counted, uint32_t* d_out, unsigned long long* d_stats, hipStream_t stream)
{
    if (ra.n_work == 0u) return hipSuccess;
    dim3 grid((ra.n_work + kBlock - 1) / kBlock), block(kBlock);
    if (exact) {
        if (counted) hipLaunchKernelGGL((trace_kernel<true, true>), grid, block, 0, stream, sc, ra, d_out, d_stats);
        else hipLaunchKernelGGL((trace_kernel<true, false>), grid, block, 0, stream, sc, ra, d_out, d_stats);
    } else {
        if (counted) hipLaunchKernelGGL((trace_kernel<false, true>), grid, block, 0, stream, sc, ra, d_out, d_stats);
        else hipLaunchKernelGGL((trace_kernel<false, false>), grid, block, 0, stream, sc, ra, d_out, d_stats);
    }
    return hipGetLastError();
}

hipError_t launch_assemble(const RenderArgs& ra, const uint32_t* d_gathered, size_t stride, uint32_t* d_frame, hipStream_t stream)
{
    uint32_t total = (uint32_t)ra.width * (uint32_t)ra.height;
    dim3 grid((total + kBlock - 1) / kBlock), block(kBlock);
    hipLaunchKernelGGL(assemble_kernel, grid, block, 0, stream, ra, d_gathered, stride, d_frame);
    return hipGetLastError();
}

}  // namespace srt
