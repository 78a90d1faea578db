// srt_launch.h -- host-callable launchers of the gfx950 kernels.
#ifndef SRT_LAUNCH_H
#define SRT_LAUNCH_H

#include <hip/hip_runtime_api.h>

#include "srt_core.h"

namespace srt {

// One pass of the trace kernel over samples [ra.sample_first, ra.sample_first + ra.sample_count) of this
// rank's pixels.  d_samples: 3 floats per (sample, pixel), sample-major, ra.sample_count * ra.n_work entries.
// d_stats: 56 x uint64 the kernel adds to: rays_path, rays_shadow, hit_nodes, box_tests, sphere_tests,
// tri_tests, rect_tests, section counts, alarm; [16, 32) cycles per part, [32, 48) the same weighted by active lanes, 48 shadow queries answered without a traversal, 49-51 the level-by-level tracer's node records, work items and given-up chunks, 52 the largest ray tree a lane of mesh_trace_kernel finished (a maximum), 53 the level-by-level tracer's largest shadow queue of a chunk, summed over its chunks, 54 its closest-hit queries that took over the answer of the same ray's query one node earlier.  d_work_counter: one uint32, reset by the launcher on `stream`.
// d_timeline: trace_timeline_bytes() of device memory; only -DSRT_TIMELINE builds write to it (per-wavefront clock records).
// d_frame_slab: slab_bytes of device memory, where the wavefronts keep the frames of pending ray-tree nodes that do not
// fit their LDS: trace_frame_slab_block_bytes() per block of the persistent grid, trace_frame_slab_bytes() for the largest
// grid the kernels ever use; with less, the grid shrinks to the blocks the slab has room for (at least one).  Contents
// need not survive a launch, and launches that overlap must not share it.
// walk_tune (scenes whose tree is walked with a stack): which traversal quantum the kernel uses -- long, short, or
// "measure both at the start of the launch and take the faster".  A calibrating launch adds what it measured to d_stats
// words 12-15 (traversals finished, wall-clock ticks; preset 0 then 1); walk_tune_prefers_short() reads them.
enum { WALK_TUNE_LONG = 0, WALK_TUNE_SHORT = 1, WALK_TUNE_CALIBRATE = 2 };
// wavefront: memory for the level-by-level tracer (srt_wavefront.hip), which launch_trace uses for scenes it serves
// (wavefront_serves) when the pool holds at least a few thousand nodes; nullptr, or a pool without memory: the one-kernel
// tracers only.  nodes_per_item is the caller's estimate of the pool entries a work item needs (0: none yet), from d_stats
// words kStatsWavefrontNodes (or kStatsWavefrontQueue, if a level's shadow queries outnumber all node records: several
// lights, shallow trees) / kStatsWavefrontItems of earlier launches; it sizes the chunks of items that share the pool.  A chunk that outgrows the pool all the same is traced by mesh_trace_kernel (kStatsWavefrontGivenUp counts them).
// (Chunks on two streams, each in its own half of the pool, were measured too: the kernels of the two streams do run at
// the same time, each at half its speed -- 301 ms against 290 ms for the 10,000-sphere scene -- so the chunks run one
// after the other.)
// far_stacks: wavefront_far_stack_bytes() of device memory, where the traversal kernels keep the far ends of their lanes'
// stacks (the near ends are in LDS); contents need not survive a launch.
struct WavefrontPool { void* mem = nullptr; size_t bytes = 0; double nodes_per_item = 0.0; void* far_stacks = nullptr; };
// The pool starts with a header: one "given up" word per chunk of a launch, then the control block of the chunk at work.
enum : uint32_t { WAVEFRONT_LEVELS = 16, WAVEFRONT_LIGHTS = 4, WAVEFRONT_MAX_CHUNKS = 2048, WAVEFRONT_CONTROL_BYTES = 1024, WAVEFRONT_MAX_WALK_BLOCKS = 4096 };
size_t wavefront_header_bytes();
size_t wavefront_far_stack_bytes();
enum { kStatsUnaskedWord = 48, kStatsWavefrontNodes = 49, kStatsWavefrontItems = 50, kStatsWavefrontGivenUp = 51,
       kStatsLargestTree = 52, kStatsWavefrontQueue = 53, kStatsWavefrontTwins = 54 };   // 52: the largest ray tree (GetRayColor calls) a lane of mesh_trace_kernel finished (a maximum, not a sum)
bool wavefront_serves(const SceneView& sc, const RenderArgs& ra, bool exact);
size_t wavefront_node_bytes();                                              // pool bytes per node record
uint32_t wavefront_capacity(size_t pool_bytes);                             // nodes a pool of that size holds (0: too small to use)
const uint32_t* wavefront_chunk_flag(void* pool_mem, uint32_t chunk);       // device address of "chunk `chunk` of the launch was given up"
// Items [item0, item1) of the pass -- chunk `chunk` of the launch -- traced level by level; queues its kernels on
// `stream` and waits for nothing.  The caller zeroes the chunk flags.
hipError_t launch_wavefront_chunk(const SceneView& sc, const RenderArgs& ra, bool counted, float* d_samples, unsigned long long* d_stats,
                                  void* pool_mem, size_t pool_bytes, void* far_stacks, uint32_t chunk, uint32_t item0, uint32_t item1, hipStream_t stream);
hipError_t launch_trace(const SceneView& sc, const RenderArgs& ra, bool exact, bool counted, float* d_samples,
                        unsigned long long* d_stats, uint32_t* d_work_counter, float* d_timeline, void* d_frame_slab, size_t slab_bytes, hipStream_t stream,
                        int walk_tune = WALK_TUNE_LONG, const WavefrontPool* wavefront = nullptr);
size_t trace_frame_slab_bytes();
size_t trace_frame_slab_block_bytes();
// The decision rule, shared by the kernel and the host: short quanta must finish traversals at least 3 % faster.
SRT_HD bool walk_tune_prefers_short(double done_long, double ticks_long, double done_short, double ticks_short)
{
    return done_short * ticks_long > 1.03 * done_long * ticks_short;
}
constexpr unsigned long long kWalkTuneEnoughTicks = 15000000ull;   // 50 wavefront windows of 3 ms (100 MHz wall clock) per preset
// The address, on the current device `device`, of the library's static workspace of SRT_STATIC_SCRATCH_BYTES (the memory
// srt_render_device_static works in).  Called once per device at set-up; allocates nothing.
hipError_t static_scratch_setup(int device, unsigned char** ptr);
size_t trace_timeline_bytes();
size_t trace_timeline_records();  // experiments: how many records (4 x uint64 each, one per wavefront of the largest grid) there are
size_t trace_timeline_offset();   // experiments: where a -DSRT_TIMELINE build leaves its per-wavefront clock records in d_timeline
// Adds the pass's samples to the per-pixel running sums (3 floats per pixel; untouched when the pass is both the
// first and the last) and, on the last pass, writes the packed pixels.  preview_scale > 0: earlier passes write a
// preview (running sum times preview_scale) to d_out as well.
hipError_t launch_resolve(const RenderArgs& ra, const float* d_samples, float* d_running, uint32_t* d_out, bool first_pass, bool last_pass,
                          float preview_scale, hipStream_t stream);
// One in-place pass of a post-process filter (SRT_POST_*): d_in is the picture before the pass and stays untouched,
// d_out receives the picture after it.  weight is clamped here; param = contrast / threshold.  d_progress:
// post_progress_bytes() of device memory, zeroed by the caller before the first pass; its last word is non-zero
// afterwards if a band gave up waiting for the one above it (the picture is then incomplete).
size_t post_progress_bytes();
hipError_t launch_post_filter(int filter, const uint32_t* d_in, uint32_t* d_out, int width, int height, float weight, float param,
                              uint32_t* d_progress, hipStream_t stream);
// d_frame[i] = Color(d_frame[i].ToVector3() / 2 + d_buffer[i].ToVector3() / 2) (Game1.cs:209-220); first: d_frame counts as all zero.
hipError_t launch_mix(uint32_t* d_frame, const uint32_t* d_buffer, uint32_t n_pixels, bool first, hipStream_t stream);
hipError_t launch_assemble(const RenderArgs& ra, const uint32_t* d_gathered, size_t stride, uint32_t* d_frame, hipStream_t stream);

}  // namespace srt
#endif
