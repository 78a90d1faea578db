// srt_multi.h -- one frame over the several devices of a context: the host-side orchestration, without any HIP call.
//
// The reference fans a render out over worker threads that write disjoint row bands of one shared Color[]
// (Game1.cs:161-167, 376-380).  Here the frame is cut into screen tiles, tile t goes to device t % n (interleaving
// balances the cheap sky against the expensive spheres; contiguous bands do not), every device traces its tiles into a
// tile-major buffer, the buffers are gathered on the first device and a last step scatters them into the row-major
// frame.  The RNG is keyed by the global pixel index, so the frame does not depend on n.
//
// plan_multi() derives what each device does; render_multi() runs the steps on a backend: the HIP devices of libsrt.so,
// or the virtual devices of the tests' CPU harness, which runs this same code without a GPU.
#ifndef SRT_MULTI_H
#define SRT_MULTI_H

#include <string>

#include "srt_scene.h"

namespace srt {

struct MultiPlan {
    int n = 0;
    srt_render_params part[SRT_MAX_DEVICES];   // what device k traces: SRT_LAYOUT_TILES, rank k of n
    int64_t pixels[SRT_MAX_DEVICES];           // size of device k's tile buffer (0: a device without tiles)
    int64_t stride = 0;                        // distance between the devices' buffers once gathered (device 0 owns the most tiles)
    int64_t frame_pixels = 0;
    RenderArgs assemble;                       // arguments of the scatter step (tile grid of the whole frame)
};

// p describes the whole frame (SRT_LAYOUT_FRAME, world_size <= 1); tile_w / tile_h choose the tile, 0 = 32.
inline int plan_multi(const srt_render_params* p, int n, MultiPlan& plan, std::string& err)
{
    if (!p) { err = "params is NULL"; return SRT_E_INVALID; }
    if (n < 1 || n > SRT_MAX_DEVICES) { err = "device count out of range"; return SRT_E_INVALID; }
    if (p->layout != SRT_LAYOUT_FRAME || p->world_size > 1 || p->rank != 0) {
        err = "a context with several devices partitions the frame itself: SRT_LAYOUT_FRAME, rank 0, world_size <= 1";
        return SRT_E_INVALID;
    }
    plan.n = n;
    for (int k = 0; k < n; k++) {
        srt_render_params q = *p;
        q.layout = SRT_LAYOUT_TILES;
        q.rank = k;
        q.world_size = n;
        plan.part[k] = q;
        plan.pixels[k] = output_pixels(&q);
        if (plan.pixels[k] < 0) { RenderArgs ra; Camera none{}; return make_render_args(&q, none, ra, err); }
    }
    plan.stride = plan.pixels[0];
    plan.frame_pixels = (int64_t)p->width * p->height;
    Camera none{};
    return make_render_args(&plan.part[0], none, plan.assemble, err);
}

// Backend: int prepare(const MultiPlan&); int trace_tiles(int k, const MultiPlan&); int gather(const MultiPlan&);
//          int assemble(const MultiPlan&).  Each returns SRT_OK or a negative SRT_E_* code.
template <class Backend>
int render_multi(Backend& b, const MultiPlan& plan)
{
    int rc = b.prepare(plan);
    for (int k = 0; k < plan.n && rc == SRT_OK; k++) rc = b.trace_tiles(k, plan);
    if (rc == SRT_OK) rc = b.gather(plan);
    if (rc == SRT_OK) rc = b.assemble(plan);
    return rc;
}

}  // namespace srt
#endif
