// srt_api.cpp -- the C ABI of libsrt.so (include/srt.h) over the gfx950 kernels.
//
// There is no CPU path here: every render goes through launch_trace() on a gfx950 device, and
// srt_create fails when there is none.
//
// A context owns one DeviceState per device of its configuration: a replica of the scene, a stream, the sample
// buffer and the counters.  A context with one device runs everything there.  A context with several devices
// splits a frame into screen tiles (srt_multi.h), traces them on all devices at once, gathers the tile buffers on
// the first device -- grouped ncclSend / ncclRecv over xGMI, or peer copies -- and assembles the frame there.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/srt.h"
#include "srt_launch.h"
#include "srt_multi.h"
#include "srt_scene.h"

using namespace srt;

namespace {

// RCCL is bound at run time, and only by contexts that gather with it: a process that renders on one GPU never
// loads it, and a process that already holds a copy (PyTorch ships one) keeps using that copy.
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;

    bool load(std::string& err)
    {
        if (handle) return true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) { err = std::string("librccl could not be loaded: ") + dlerror(); return false; }
        bool ok = true;
        auto sym = [&](const char* name) { void* p = dlsym(handle, name); if (!p) { ok = false; err = std::string("librccl lacks ") + name; } return p; };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        CommCount = reinterpret_cast<decltype(CommCount)>(sym("ncclCommCount"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!ok) {
            // a library without the whole interface is no library: the next context that asks gets the same answer
            // instead of a call through a null pointer
            dlclose(handle);
            handle = nullptr;
        }
        return ok;
    }
};
RcclApi g_rccl;

// What one device of a context holds.
struct DeviceState {
    int device = 0;                       // HIP ordinal
    hipStream_t stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    hipEvent_t ev_tiles = nullptr;        // peer-copy gather: this device's tiles have arrived on the first device
    q4 *d_nodes = nullptr, *d_prims = nullptr, *d_normals = nullptr, *d_materials = nullptr, *d_lights = nullptr, *d_tiny = nullptr;
    uint32_t* d_out = nullptr;            // the frame (one device) / this device's tile buffer (several)
    size_t out_capacity = 0;              // pixels
    unsigned long long* d_stats = nullptr;
    uint32_t* d_counter = nullptr;        // work counter of the persistent trace kernel
    float* d_timeline = nullptr;          // scratch of -DSRT_TIMELINE builds (per-wavefront clock records)
    void* d_frame_slab = nullptr;         // frames of pending ray-tree nodes, one part per wavefront of the grid
    size_t slab_bytes = 0;                // its size: trace_frame_slab_bytes(), or less when SRT_FRAME_SLAB_MB bounds it
    float* d_samples = nullptr;           // per-(sample, pixel) colours of one pass, 12 bytes each
    size_t samples_capacity = 0;          // bytes
    float* d_running = nullptr;           // per-pixel running sums, only for renders that need several passes
    size_t running_capacity = 0;          // bytes
    bool stats_pending = false;
    // The stream of the most recent launch.  The sample buffer, the running sums, the work counter and the two events
    // are shared by all launches of this device, so a launch on another stream first waits for the previous one.
    hipStream_t last_stream = nullptr;
    bool launched = false;
    srt_stats part{};                     // this device's share of the last collection
    unsigned char* static_scratch = nullptr;   // this GPU's copy of the library's static workspace (srt_render_device_static)
    // Traversal quantum of the mesh kernel for the committed scene: measured by the first launches after a commit
    // (WALK_TUNE_CALIBRATE), then fixed to what they found.
    int walk_tune = WALK_TUNE_CALIBRATE;
    // Node pool of the level-by-level tracer: allocated by the first render that tracer takes, grown by renders that need
    // more (see launch()), kept until the context goes.
    WavefrontPool wavefront;
    bool wavefront_full = false;          // the pool is as large as the device allows (or cannot be had at all)
    // GetRayColor calls per work item in the renders of the committed scene so far (0: none collected yet).  The
    // level-by-level tracer pays per item for streaming the first level through memory and gains per node of the deeper
    // levels, so it takes over once this is above SRT_WAVEFRONT_MIN_RAYS (default 5).
    double rays_per_item = 0.0;
    uint64_t items_since_collect = 0;
    // ... or as soon as one ray tree of SRT_WAVEFRONT_MIN_TREE (default 32) calls was seen: a deep tree is walked by one lane,
    // node after node, and thirty small glossy spheres beside a mesh (1.57 rays per sample) already make the one-kernel
    // tracer's frame 38.6 ms instead of 9.0, against 16.9 ms level by level.
    uint64_t largest_tree = 0;
};

}  // namespace

struct srt_ctx {
    std::vector<DeviceState> dev;         // dev[0] assembles the frame
    int gather = SRT_GATHER_PEER_COPY;    // how tiles reach dev[0] (several devices)
    ncclComm_t comms[SRT_MAX_DEVICES] = {};
    int rccl_ranks = 0;
    HostScene host;
    FlatScene flat;
    bool committed = false;
    int world_kind = SRT_WORLD_BVHNODE;
    Camera cam{};
    bool has_camera = false;
    uint64_t sample_budget = (uint64_t)8 << 30;  // bytes the sample buffer of a device may take (SRT_SAMPLE_BUFFER_MB); 8K at 1024 spp: 7.04 s with 2 GiB, 6.83 s with 8
    srt_stats stats{};
    // dev[0] only
    uint32_t* d_gathered = nullptr;       // several devices: every device's tile buffer, `stride` pixels apart
    size_t gathered_capacity = 0;         // pixels
    uint32_t* d_frame = nullptr;          // several devices: the assembled row-major frame
    size_t frame_capacity = 0;
    hipEvent_t ev_own_tiles = nullptr, ev_gathered = nullptr, ev_assembled = nullptr;   // several devices: marks on dev[0]'s stream
    bool multi_pending = false;           // the marks of a frame have not been read yet
    uint32_t* d_mix = nullptr;            // one buffer of the mix-buffer technique
    size_t mix_capacity = 0;              // in pixels
    uint32_t* d_post[2] = {nullptr, nullptr};   // the picture before / after a pass of a post-process filter
    size_t post_capacity = 0;             // pixels
    uint32_t* d_post_progress = nullptr;
    // asynchronous render (srt_render_async / srt_poll / srt_wait)
    uint32_t* h_pinned = nullptr;
    size_t pinned_capacity = 0;
    uint32_t* async_dst = nullptr;
    size_t async_pixels = 0;
    bool async_pending = false;
    // A new scene's asynchronous render starts with one probe sample per pixel; srt_poll / srt_wait queue the rest once
    // that sample is done and its ray count says which tracer the scene wants (see launch()).
    bool async_probing = false;
    srt_render_params async_params{};
    // progressive render: one event per pass, recorded after the pass's preview reached pinned memory
    enum { MAX_PASSES = 64 };
    hipEvent_t pass_done[MAX_PASSES] = {};
    int passes_total = 0;
    std::string error;
};

namespace {

constexpr int kStatsWords = 56;   // counters a trace kernel adds to (srt_launch.h)
thread_local std::string g_create_error;

int fail(srt_ctx* c, int code, const std::string& msg)
{
    if (c) c->error = msg;
    else g_create_error = msg;
    return code;
}

int hip_fail(srt_ctx* c, hipError_t e, const char* what)
{
    return fail(c, SRT_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define SRT_HIP(c, call)                                         \
    do {                                                         \
        hipError_t e_ = (call);                                  \
        if (e_ != hipSuccess) return hip_fail((c), e_, #call);   \
    } while (0)
#define SRT_NCCL(c, call)                                                                                             \
    do {                                                                                                              \
        ncclResult_t r_ = (call);                                                                                     \
        if (r_ != ncclSuccess) return fail((c), SRT_E_HIP, std::string(#call) + ": " + g_rccl.GetErrorString(r_));     \
    } while (0)

void free_scene_buffers(DeviceState& d)
{
    hipFree(d.d_nodes); hipFree(d.d_prims); hipFree(d.d_normals); hipFree(d.d_materials); hipFree(d.d_lights); hipFree(d.d_tiny);
    d.d_nodes = d.d_prims = d.d_normals = d.d_materials = d.d_lights = d.d_tiny = nullptr;
}

int upload(srt_ctx* c, const std::vector<q4>& src, q4** dst)
{
    SRT_HIP(c, hipMalloc((void**)dst, src.size() * sizeof(q4)));
    SRT_HIP(c, hipMemcpy(*dst, src.data(), src.size() * sizeof(q4), hipMemcpyHostToDevice));
    return SRT_OK;
}

SceneView scene_view(const srt_ctx* c, const DeviceState& d)
{
    return c->flat.view(d.d_nodes, d.d_prims, d.d_normals, d.d_materials, d.d_lights, d.d_tiny);
}

// grows a device buffer of 4-byte pixels (never shrinks); waits for work in flight before it frees the old one
int ensure_pixels(srt_ctx* c, uint32_t** buf, size_t* capacity, size_t pixels)
{
    if (pixels <= *capacity && *buf) return SRT_OK;
    if (*buf) { SRT_HIP(c, hipDeviceSynchronize()); (void)hipFree(*buf); }
    *buf = nullptr; *capacity = 0;
    SRT_HIP(c, hipMalloc((void**)buf, (pixels ? pixels : 1) * sizeof(uint32_t)));
    *capacity = pixels;
    return SRT_OK;
}

void zero_counts(srt_stats& s)
{
    s.rays_path = s.rays_shadow = s.hit_nodes = 0;
    s.box_tests = s.sphere_tests = s.tri_tests = s.rect_tests = 0;
    s.trace_sections = s.trace_lanes = s.light_sections = s.light_lanes = 0;
    for (int i = 0; i < 16; i++) s.section_cycles[i] = s.section_lane_cycles[i] = 0;
    s.shadow_unasked = 0;
    s.wavefront_items = s.wavefront_nodes = s.wavefront_given_up = s.wavefront_twins = 0;
    s.kernel_ms = 0.0;
}

// Fetches (and zeroes) the counters accumulated by the launches since the last collection, once the most recent
// render of every device has finished.  kernel_ms is the device time of that render (all its passes), the longest
// one's when several devices took part.
int collect_stats(srt_ctx* c)
{
    bool alarm = false;
    for (DeviceState& d : c->dev) {
        if (!d.stats_pending) continue;
        SRT_HIP(c, hipSetDevice(d.device));
        SRT_HIP(c, hipEventSynchronize(d.ev_end));
        unsigned long long s[kStatsWords];
        SRT_HIP(c, hipMemcpy(s, d.d_stats, sizeof(s), hipMemcpyDeviceToHost));
        SRT_HIP(c, hipMemsetAsync(d.d_stats, 0, sizeof(s), d.stream));
        SRT_HIP(c, hipStreamSynchronize(d.stream));
#ifdef SRT_TIMELINE
        if (const char* path = std::getenv("SRT_TIMELINE_FILE")) {
            std::vector<unsigned long long> tl(srt::trace_timeline_records() * 4);
            hipMemcpy(tl.data(), reinterpret_cast<char*>(d.d_timeline) + srt::trace_timeline_offset(), tl.size() * 8, hipMemcpyDeviceToHost);
            if (FILE* f = std::fopen(path, "wb")) { std::fwrite(tl.data(), 8, tl.size(), f); std::fclose(f); }
        }
#endif
        float ms = 0.0f;
        SRT_HIP(c, hipEventElapsedTime(&ms, d.ev_begin, d.ev_end));
        srt_stats& p = d.part;
        p.rays_path = s[0]; p.rays_shadow = s[1]; p.hit_nodes = s[2];
        p.box_tests = s[3]; p.sphere_tests = s[4]; p.tri_tests = s[5]; p.rect_tests = s[6];
        p.trace_sections = s[7]; p.trace_lanes = s[8]; p.light_sections = s[9]; p.light_lanes = s[10];
        for (int i = 0; i < 16; i++) { p.section_cycles[i] = s[16 + i]; p.section_lane_cycles[i] = s[32 + i]; }
        p.shadow_unasked = s[48];
        if (d.walk_tune == WALK_TUNE_CALIBRATE && s[13] >= kWalkTuneEnoughTicks && s[15] >= kWalkTuneEnoughTicks)
            d.walk_tune = walk_tune_prefers_short((double)s[12], (double)s[13], (double)s[14], (double)s[15]) ? WALK_TUNE_SHORT : WALK_TUNE_LONG;
        // pool entries per work item of the level-by-level tracer's chunks (node records, or one level's shadow queries
        // where those are more): the next launches size their chunks by it
        if (s[kStatsWavefrontItems] != 0ull) {
            const unsigned long long entries = s[kStatsWavefrontNodes] > s[kStatsWavefrontQueue] ? s[kStatsWavefrontNodes] : s[kStatsWavefrontQueue];
            d.wavefront.nodes_per_item = (double)entries / (double)s[kStatsWavefrontItems];
        }
        if (s[kStatsWavefrontGivenUp] != 0ull) d.wavefront.nodes_per_item = 2.0 * (d.wavefront.nodes_per_item > 0.0 ? d.wavefront.nodes_per_item : 24.0);
        p.wavefront_items = s[kStatsWavefrontItems]; p.wavefront_nodes = s[kStatsWavefrontNodes]; p.wavefront_given_up = s[kStatsWavefrontGivenUp]; p.wavefront_twins = s[kStatsWavefrontTwins];
        if (d.items_since_collect != 0) d.rays_per_item = (double)s[0] / (double)d.items_since_collect;
        d.items_since_collect = 0;
        if (s[kStatsLargestTree] > d.largest_tree) d.largest_tree = s[kStatsLargestTree];
        p.kernel_ms = ms;
        d.stats_pending = false;
        alarm = alarm || s[11] != 0ull;
    }
    if (c->multi_pending) {
        SRT_HIP(c, hipSetDevice(c->dev[0].device));
        SRT_HIP(c, hipEventSynchronize(c->ev_assembled));
        float g = 0.0f, a = 0.0f;
        SRT_HIP(c, hipEventElapsedTime(&g, c->ev_own_tiles, c->ev_gathered));
        SRT_HIP(c, hipEventElapsedTime(&a, c->ev_gathered, c->ev_assembled));
        c->stats.gather_ms = g; c->stats.assemble_ms = a;
        c->multi_pending = false;
    }
    zero_counts(c->stats);
    for (DeviceState& d : c->dev) {
        const srt_stats& p = d.part;
        srt_stats& t = c->stats;
        t.rays_path += p.rays_path; t.rays_shadow += p.rays_shadow; t.hit_nodes += p.hit_nodes;
        t.box_tests += p.box_tests; t.sphere_tests += p.sphere_tests; t.tri_tests += p.tri_tests; t.rect_tests += p.rect_tests;
        t.trace_sections += p.trace_sections; t.trace_lanes += p.trace_lanes; t.light_sections += p.light_sections; t.light_lanes += p.light_lanes;
        for (int i = 0; i < 16; i++) { t.section_cycles[i] += p.section_cycles[i]; t.section_lane_cycles[i] += p.section_lane_cycles[i]; }
        t.shadow_unasked += p.shadow_unasked;
        t.wavefront_items += p.wavefront_items; t.wavefront_nodes += p.wavefront_nodes; t.wavefront_given_up += p.wavefront_given_up; t.wavefront_twins += p.wavefront_twins;
        if (p.kernel_ms > t.kernel_ms) t.kernel_ms = p.kernel_ms;
    }
    c->stats.walk_tune = (uint32_t)c->dev[0].walk_tune;
    if (alarm) return fail(c, SRT_E_HIP, "trace kernel: a wavefront gave up waiting for a ray subtree that another lane took over; the picture is incomplete");
    return SRT_OK;
}

// Memory a launch works in besides the scene and the output.  The usual launch takes it from the device's own
// buffers, growing them as needed.
struct Workspace {
    float* samples = nullptr; size_t samples_bytes = 0;
    float* running = nullptr; size_t running_bytes = 0;
    uint32_t* counter = nullptr;
    bool fixed = false;     // true: never allocate; a render that does not fit is refused
};

// min_passes > 1 (progressive renders): at least that many passes, each followed by a preview in d_out, a copy of it
// to pinned memory and an event.
// One pass of a render whose passes the caller schedules itself (contexts with several devices: the mix-buffer technique and
// progressive renders, where something happens between the passes of the devices): samples [first_sample, first_sample + count),
// resolved into d_out -- added to the running sums unless it is both the first and the last pass.
struct PassSpec { int first_sample = 0, count = 1; bool first = true, last = true; float preview = 0.0f; };

// Whether the first render of a scene should find out with one sample per pixel which tracer the scene wants.
static double wavefront_min_rays()
{
    const char* knob = std::getenv("SRT_WAVEFRONT_MIN_RAYS");   // read per launch: tests switch tracers inside one process
    return knob ? std::atof(knob) : 5.0;
}
static bool wants_probe(srt_ctx* c, const DeviceState& d, const srt_render_params* p)
{
    if (!c->committed || !c->has_camera || c->dev.size() != 1 || d.rays_per_item != 0.0 || wavefront_min_rays() <= 0.0 || p->samples < 2) return false;
    RenderArgs ra;
    std::string ignored;
    if (make_render_args(p, c->cam, ra, ignored) != SRT_OK || ra.n_work == 0u) return false;
    return wavefront_serves(scene_view(c, d), ra, c->world_kind == SRT_WORLD_BVHNODE);
}

int launch(srt_ctx* c, DeviceState& d, const srt_render_params* p, uint32_t* d_out, size_t capacity, hipStream_t stream, bool counted,
           int min_passes = 1, int mix_buffers = 0, const Workspace* fixed_ws = nullptr, const PassSpec* one_pass = nullptr, bool may_wait = false)
{
    if (!c->committed) return fail(c, SRT_E_STATE, "srt_commit_scene has not been called");
    if (!c->has_camera) return fail(c, SRT_E_STATE, "srt_set_camera has not been called");
    RenderArgs ra;
    int rc = make_render_args(p, c->cam, ra, c->error);
    if (rc != SRT_OK) return rc;
    int64_t need = output_pixels(p);
    if (need < 0) return fail(c, (int)need, "bad parameters");
    if ((size_t)need > capacity) return fail(c, SRT_E_INVALID, "output buffer too small for this layout");
    if (!d_out && need > 0) return fail(c, SRT_E_INVALID, "output pointer is NULL");
    SRT_HIP(c, hipSetDevice(d.device));
    if (&d == &c->dev[0]) c->passes_total = 0;
    if (ra.n_work == 0u) {
        // a rank without tiles: nothing to launch, and the counters of the "render" are all zero
        if (!d.stats_pending) zero_counts(d.part);
        return SRT_OK;
    }
    // Samples per pass: as many as fit the sample buffer budget (12 bytes per pixel and sample) and the
    // 2^31 work-item limit of one launch.
    if ((uint64_t)ra.n_work > (((uint64_t)1 << 31) - 1)) return fail(c, SRT_E_LIMIT, "more than 2^31 pixels per rank");
    const uint64_t per_sample = (uint64_t)ra.n_work * 12u;
    const uint64_t budget = fixed_ws ? (uint64_t)fixed_ws->samples_bytes : c->sample_budget;
    const uint64_t fit = budget / per_sample;
    const uint64_t by_items = (((uint64_t)1 << 31) - 1) / ra.n_work;
    uint64_t per_pass = fit < by_items ? fit : by_items;
    if (per_pass < 1) {
        if (fixed_ws) return fail(c, SRT_E_LIMIT, "the render does not fit the fixed workspace");
        per_pass = 1;
    }
    if (per_pass > (uint64_t)ra.samples) per_pass = (uint64_t)ra.samples;
    if (one_pass) {
        // a span of samples that does not fit the sample buffer is traced in several passes; only its last pass can show a preview
        if ((uint64_t)one_pass->count > per_pass && one_pass->preview != 0.0f) return fail(c, SRT_E_LIMIT, "a pass of this size does not fit the sample buffer");
        if ((uint64_t)one_pass->count < per_pass) per_pass = (uint64_t)one_pass->count;
    }
    if (mix_buffers > 0) {
        if (ra.layout != 0 || ra.world_size != 1) return fail(c, SRT_E_INVALID, "the mix-buffer technique renders whole frames (SRT_LAYOUT_FRAME, world_size 1)");
        if ((uint32_t)ra.mode != MODE_COLOR) return fail(c, SRT_E_INVALID, "the mix-buffer technique calls GetRayColor (SRT_MODE_COLOR)");
        per_pass = 1;
        if ((rc = ensure_pixels(c, &c->d_mix, &c->mix_capacity, (size_t)need)) != SRT_OK) return rc;
    }
    const bool progressive = min_passes > 1;
    if (progressive) {
        uint64_t share = ((uint64_t)ra.samples + (uint64_t)min_passes - 1) / (uint64_t)min_passes;
        if (share < 1) share = 1;
        if (per_pass > share) per_pass = share;
        // never more passes than events: later passes grow instead
        uint64_t floor_pp = ((uint64_t)ra.samples + srt_ctx::MAX_PASSES - 1) / srt_ctx::MAX_PASSES;
        if (per_pass < floor_pp) {
            if (floor_pp * per_sample > budget || floor_pp > by_items) return fail(c, SRT_E_LIMIT, "a progressive render of this size needs more than 64 passes");
            per_pass = floor_pp;
        }
    }
    const size_t need_samples = (size_t)(per_pass * per_sample);
    const SceneView view = scene_view(c, d);
    const bool exact = c->world_kind == SRT_WORLD_BVHNODE;
    const double min_rays = wavefront_min_rays();
    // The first render of a scene does not know yet how many rays a sample takes, and a host like the reference's renders
    // one frame per scene.  A caller that blocks anyway (may_wait) therefore gets the first sample of every pixel through the
    // one-kernel tracer, the library reads the ray count of that sample (one wait for the stream), and the remaining
    // samples go to whichever tracer that count calls for.  The picture is the same either way.
    const bool probe = may_wait && !fixed_ws && !one_pass && mix_buffers == 0 && min_passes <= 1 && wants_probe(c, d, p);
    const bool multi_pass = one_pass ? !(one_pass->first && one_pass->last) : (per_pass < (uint64_t)ra.samples || probe);
    Workspace ws;
    if (fixed_ws) {
        ws = *fixed_ws;
        if (need_samples > ws.samples_bytes || (multi_pass && (size_t)per_sample > ws.running_bytes))
            return fail(c, SRT_E_LIMIT, "the render does not fit the fixed workspace");
    } else {
        if (need_samples > d.samples_capacity) {
            // (re)allocation waits for work in flight; steady-state renders of one size never get here
            SRT_HIP(c, hipDeviceSynchronize());
            if (d.d_samples) hipFree(d.d_samples);
            d.d_samples = nullptr; d.samples_capacity = 0;
            SRT_HIP(c, hipMalloc((void**)&d.d_samples, need_samples));
            d.samples_capacity = need_samples;
        }
        if (multi_pass && (size_t)per_sample > d.running_capacity) {
            SRT_HIP(c, hipDeviceSynchronize());
            if (d.d_running) hipFree(d.d_running);
            d.d_running = nullptr; d.running_capacity = 0;
            SRT_HIP(c, hipMalloc((void**)&d.d_running, (size_t)per_sample));
            d.running_capacity = (size_t)per_sample;
        }
        ws.samples = d.d_samples; ws.samples_bytes = d.samples_capacity;
        ws.running = d.d_running; ws.running_bytes = d.running_capacity;
        ws.counter = d.d_counter;
    }
    // A launch that is being recorded into a HIP graph (stream capture) runs later, perhaps many times: it is neither
    // timed nor ordered against this device's earlier launches (the caller orders a graph against other work), and it
    // leaves the context's events alone -- an event recorded under capture can never be waited for or read, which would
    // fail every later collection.  The counters of its replays are added to the next collection's.
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    const bool recorded = hipStreamIsCapturing(stream, &capture) == hipSuccess && capture == hipStreamCaptureStatusActive;
    (void)hipGetLastError();
    if (recorded && (progressive || mix_buffers > 0)) return fail(c, SRT_E_STATE, "this render cannot be recorded into a graph");
    // launches of one device share its workspace, counters and events: one on another stream waits for the last one
    if (!recorded) {
        if (d.launched && d.last_stream != stream) SRT_HIP(c, hipStreamWaitEvent(stream, d.ev_end, 0));
        d.last_stream = stream;
        d.launched = true;
        if (!one_pass || one_pass->first) SRT_HIP(c, hipEventRecord(d.ev_begin, stream));
    }
    const WavefrontPool* wavefront = nullptr;
    uint32_t first_sample_of_loop = 0u;
    if (probe) {
        ra.sample_first = 0u;
        ra.sample_count = 1u;
        ra.n_items = ra.n_work;
        hipError_t e = launch_trace(view, ra, exact, counted, ws.samples, d.d_stats, ws.counter, d.d_timeline, d.d_frame_slab, d.slab_bytes, stream, d.walk_tune, nullptr);
        d.items_since_collect += ra.n_items;
        if (e != hipSuccess) return hip_fail(c, e, "trace kernel launch");
        e = launch_resolve(ra, ws.samples, ws.running, d_out, true, false, 0.0f, stream);
        if (e != hipSuccess) return hip_fail(c, e, "resolve kernel launch");
        unsigned long long seen[kStatsWords];
        SRT_HIP(c, hipMemcpyAsync(seen, d.d_stats, sizeof(seen), hipMemcpyDeviceToHost, stream));
        SRT_HIP(c, hipStreamSynchronize(stream));
        // the counters may hold earlier launches that were not collected yet: an estimate from above, which only ever
        // sends a scene to the level-by-level tracer one render early
        d.rays_per_item = (double)seen[0] / (double)d.items_since_collect;
        if (seen[kStatsLargestTree] > d.largest_tree) d.largest_tree = seen[kStatsLargestTree];
        first_sample_of_loop = 1u;
    }
    if (std::getenv("SRT_DEBUG_LAUNCH")) std::fprintf(stderr, "srt: rays per item %.2f (level-by-level tracer from %.2f), serves %d, probed %d\n", d.rays_per_item, min_rays, (int)wavefront_serves(view, ra, exact), (int)probe);
    const char* min_tree_knob = std::getenv("SRT_WAVEFRONT_MIN_TREE");
    const uint64_t min_tree = min_tree_knob ? (uint64_t)std::atoll(min_tree_knob) : 32u;
    if (!fixed_ws && (d.rays_per_item >= min_rays || (min_rays > 0.0 && d.largest_tree >= min_tree)) && wavefront_serves(view, ra, exact)) {
        // The pool is sized for what the largest pass of this render is expected to need (so that it is one chunk), within
        // SRT_WAVEFRONT_POOL_MB (default 192 GiB) and 60 % of the memory that is free; it grows when a later render needs
        // more and is never shrunk.  Few large chunks matter: the 10,000-sphere scene at 1080p and 16 samples takes
        // 411 ms with 4 GiB (40 chunks), 251 ms with 16 GiB (10) and 198 ms in one chunk.
        const double per_item = d.wavefront.nodes_per_item > 0.0 ? d.wavefront.nodes_per_item : 1.25 * (d.rays_per_item > 4.0 ? d.rays_per_item : 4.0);
        const uint64_t pass_items = (uint64_t)ra.n_work * (mix_buffers > 0 ? 1u : per_pass);
        const double want_nodes = 1.35 * per_item * (double)pass_items;
        size_t want = wavefront_header_bytes() + (size_t)(want_nodes * (double)wavefront_node_bytes());
        size_t cap = (size_t)196608 << 20;
        if (const char* mb = std::getenv("SRT_WAVEFRONT_POOL_MB")) cap = (size_t)std::atoll(mb) << 20;
        if (want > cap) want = cap;
        if (want < ((size_t)256 << 20)) want = cap < ((size_t)256 << 20) ? cap : (size_t)256 << 20;
        if (want > d.wavefront.bytes && !recorded && !d.wavefront_full) {
            size_t free_b = 0, total_b = 0;
            SRT_HIP(c, hipDeviceSynchronize());      // the old pool may be in use; steady-state renders of one size never get here
            if (d.wavefront.mem) { (void)hipFree(d.wavefront.mem); d.wavefront.mem = nullptr; d.wavefront.bytes = 0; }
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && want > free_b / 10 * 6) { want = free_b / 10 * 6; d.wavefront_full = true; }
            // (an allocation that fails although the memory is said to be free is tried again at half the size: a smaller pool
            // means more chunks, no pool means the one-kernel tracer for good)
            while (wavefront_capacity(want) != 0u) {
                if (hipMalloc(&d.wavefront.mem, want) == hipSuccess) { d.wavefront.bytes = want; break; }
                (void)hipGetLastError();
                d.wavefront.mem = nullptr;
                d.wavefront_full = true;
                want /= 2;
            }
            if (!d.wavefront.mem) d.wavefront_full = true;
            if (d.wavefront.mem && !d.wavefront.far_stacks && wavefront_far_stack_bytes() != 0 &&
                hipMalloc(&d.wavefront.far_stacks, wavefront_far_stack_bytes()) != hipSuccess) {
                (void)hipGetLastError();
                (void)hipFree(d.wavefront.mem);
                d.wavefront.mem = nullptr; d.wavefront.bytes = 0; d.wavefront.far_stacks = nullptr; d.wavefront_full = true;
            }
        }
        if (d.wavefront.mem) wavefront = &d.wavefront;
    }
    if (mix_buffers > 0) {
        // renderBlockFull (Game1.cs:320-374) once per buffer: one sample -- sample index b of the counter RNG stands for
        // thread b's System.Random -- divided by `samples`, gamma, packing; then the pairwise mix of Game1.cs:209-220
        for (int b = 0; b < mix_buffers; b++) {
            ra.sample_first = (uint32_t)b;
            ra.sample_count = 1u;
            ra.n_items = ra.n_work;
            hipError_t e = launch_trace(view, ra, exact, counted, ws.samples, d.d_stats, ws.counter, d.d_timeline, d.d_frame_slab, d.slab_bytes, stream, d.walk_tune, wavefront);
            d.items_since_collect += ra.n_items;
            if (e != hipSuccess) return hip_fail(c, e, "trace kernel launch");
            e = launch_resolve(ra, ws.samples, ws.running, c->d_mix, true, true, 0.0f, stream);
            if (e != hipSuccess) return hip_fail(c, e, "resolve kernel launch");
            e = launch_mix(d_out, c->d_mix, (uint32_t)need, b == 0, stream);
            if (e != hipSuccess) return hip_fail(c, e, "mix kernel launch");
        }
        SRT_HIP(c, hipEventRecord(d.ev_end, stream));
        d.stats_pending = true;
        return SRT_OK;
    }
    if (one_pass) {
        const uint32_t span_first = (uint32_t)one_pass->first_sample, span_end = span_first + (uint32_t)one_pass->count;
        for (uint32_t s0 = span_first; s0 < span_end; s0 += (uint32_t)per_pass) {
            ra.sample_first = s0;
            ra.sample_count = span_end - s0 < (uint32_t)per_pass ? span_end - s0 : (uint32_t)per_pass;
            ra.n_items = ra.n_work * ra.sample_count;
            hipError_t e = launch_trace(view, ra, exact, counted, ws.samples, d.d_stats, ws.counter, d.d_timeline, d.d_frame_slab, d.slab_bytes, stream, d.walk_tune, wavefront);
            d.items_since_collect += ra.n_items;
            if (e != hipSuccess) return hip_fail(c, e, "trace kernel launch");
            const bool ends = s0 + ra.sample_count >= span_end;
            e = launch_resolve(ra, ws.samples, ws.running, d_out, one_pass->first && s0 == span_first, one_pass->last && ends, ends ? one_pass->preview : 0.0f, stream);
            if (e != hipSuccess) return hip_fail(c, e, "resolve kernel launch");
        }
        if (!recorded) {
            SRT_HIP(c, hipEventRecord(d.ev_end, stream));
            d.stats_pending = true;
        }
        return SRT_OK;
    }
    int pass = 0;
    for (uint32_t s0 = first_sample_of_loop; s0 < (uint32_t)ra.samples; s0 += (uint32_t)per_pass, pass++) {
        ra.sample_first = s0;
        ra.sample_count = (uint32_t)ra.samples - s0 < (uint32_t)per_pass ? (uint32_t)ra.samples - s0 : (uint32_t)per_pass;
        ra.n_items = ra.n_work * ra.sample_count;
        hipError_t e = launch_trace(view, ra, exact, counted, ws.samples, d.d_stats, ws.counter, d.d_timeline, d.d_frame_slab, d.slab_bytes, stream, d.walk_tune, wavefront);
            d.items_since_collect += ra.n_items;
        if (e != hipSuccess) return hip_fail(c, e, "trace kernel launch");
        const uint32_t done = s0 + ra.sample_count;
        const float preview = progressive ? (float)ra.samples / (float)done : 0.0f;
        e = launch_resolve(ra, ws.samples, ws.running, d_out, s0 == 0, done >= (uint32_t)ra.samples, preview, stream);
        if (e != hipSuccess) return hip_fail(c, e, "resolve kernel launch");
        if (progressive) {
            SRT_HIP(c, hipMemcpyAsync(c->h_pinned, d_out, (size_t)need * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            if (!c->pass_done[pass]) SRT_HIP(c, hipEventCreateWithFlags(&c->pass_done[pass], hipEventDisableTiming));
            SRT_HIP(c, hipEventRecord(c->pass_done[pass], stream));
        }
    }
    if (&d == &c->dev[0]) c->passes_total = progressive ? pass : 0;
    if (!recorded) {
        SRT_HIP(c, hipEventRecord(d.ev_end, stream));
        d.stats_pending = true;
    }
    return SRT_OK;
}

// The devices of a context as the backend of render_multi(): every step only queues work on the devices' streams;
// when dev[0]'s stream has drained, the frame is in c->d_frame.
struct HipDevices {
    srt_ctx* c;
    bool counted;

    int prepare(const MultiPlan& plan)
    {
        DeviceState& root = c->dev[0];
        SRT_HIP(c, hipSetDevice(root.device));
        int rc = ensure_pixels(c, &c->d_gathered, &c->gathered_capacity, (size_t)plan.stride * (size_t)plan.n);
        if (rc == SRT_OK) rc = ensure_pixels(c, &c->d_frame, &c->frame_capacity, (size_t)plan.frame_pixels);
        for (int k = 1; k < plan.n && rc == SRT_OK; k++) {
            DeviceState& d = c->dev[k];
            SRT_HIP(c, hipSetDevice(d.device));
            rc = ensure_pixels(c, &d.d_out, &d.out_capacity, (size_t)plan.pixels[k]);
        }
        return rc;
    }
    // device 0 traces straight into its slot of the gathered array
    int trace_tiles(int k, const MultiPlan& plan)
    {
        DeviceState& d = c->dev[k];
        uint32_t* out = k == 0 ? c->d_gathered : d.d_out;
        return launch(c, d, &plan.part[k], out, (size_t)plan.pixels[k], d.stream, counted);
    }
    // one receive on the first device, one send on device k (inside the caller's group)
    int gather_pair(const MultiPlan& plan, int k)
    {
        DeviceState& root = c->dev[0];
        SRT_HIP(c, hipSetDevice(root.device));
        SRT_NCCL(c, g_rccl.Recv(c->d_gathered + (size_t)k * (size_t)plan.stride, (size_t)plan.pixels[k], ncclUint32, k, c->comms[0], root.stream));
        SRT_HIP(c, hipSetDevice(c->dev[k].device));
        SRT_NCCL(c, g_rccl.Send(c->dev[k].d_out, (size_t)plan.pixels[k], ncclUint32, 0, c->comms[k], c->dev[k].stream));
        return SRT_OK;
    }
    int gather(const MultiPlan& plan)
    {
        DeviceState& root = c->dev[0];
        SRT_HIP(c, hipSetDevice(root.device));
        SRT_HIP(c, hipEventRecord(c->ev_own_tiles, root.stream));
        int rc = gather_tiles(plan);
        if (rc != SRT_OK) return rc;
        SRT_HIP(c, hipSetDevice(root.device));
        SRT_HIP(c, hipEventRecord(c->ev_gathered, root.stream));
        return SRT_OK;
    }
    int gather_tiles(const MultiPlan& plan)
    {
        DeviceState& root = c->dev[0];
        if (c->gather == SRT_GATHER_RCCL) {
            // every peer has its own xGMI link to the first device: (n - 1) receives there, one send on each peer
            SRT_NCCL(c, g_rccl.GroupStart());
            int rc = SRT_OK;
            for (int k = 1; k < plan.n && rc == SRT_OK; k++) {
                if (plan.pixels[k] == 0) continue;
                rc = gather_pair(plan, k);
            }
            // the group is closed whatever happened inside it: an open group would swallow the thread's later RCCL calls
            const ncclResult_t end = g_rccl.GroupEnd();
            if (rc != SRT_OK) return rc;
            SRT_NCCL(c, end);
            return SRT_OK;
        }
        for (int k = 1; k < plan.n; k++) {
            if (plan.pixels[k] == 0) continue;
            DeviceState& d = c->dev[k];
            SRT_HIP(c, hipSetDevice(d.device));
            uint32_t* dst = c->d_gathered + (size_t)k * (size_t)plan.stride;
            const size_t bytes = (size_t)plan.pixels[k] * sizeof(uint32_t);
            if (d.device == root.device) SRT_HIP(c, hipMemcpyAsync(dst, d.d_out, bytes, hipMemcpyDeviceToDevice, d.stream));
            else SRT_HIP(c, hipMemcpyPeerAsync(dst, root.device, d.d_out, d.device, bytes, d.stream));
            SRT_HIP(c, hipEventRecord(d.ev_tiles, d.stream));
            SRT_HIP(c, hipSetDevice(root.device));
            SRT_HIP(c, hipStreamWaitEvent(root.stream, d.ev_tiles, 0));
        }
        return SRT_OK;
    }
    int assemble(const MultiPlan& plan)
    {
        DeviceState& root = c->dev[0];
        SRT_HIP(c, hipSetDevice(root.device));
        hipError_t e = launch_assemble(plan.assemble, c->d_gathered, (size_t)plan.stride, c->d_frame, root.stream);
        if (e != hipSuccess) return hip_fail(c, e, "assemble kernel launch");
        SRT_HIP(c, hipEventRecord(c->ev_assembled, root.stream));
        c->multi_pending = true;
        return SRT_OK;
    }
};

// Queues one frame on all devices of the context; the frame ends up in c->d_frame on dev[0]'s stream.
int launch_multi(srt_ctx* c, const srt_render_params* p, bool counted, MultiPlan& plan)
{
    int rc = plan_multi(p, (int)c->dev.size(), plan, c->error);
    if (rc != SRT_OK) return rc;
    HipDevices backend{c, counted};
    return render_multi(backend, plan);
}

// The mix-buffer technique on the devices of a context: the reference gives every thread a frame of its own to render
// (Game1.cs:174-175, 382-394), so every device renders whole one-sample frames -- buffer b on device b % n -- and hands
// them to the first device, which mixes them into the picture in buffer order, as Game1.Update does (:209-220; the mix
// rounds to 8 bits after every buffer, so the order is part of the result).  The picture ends up in c->d_frame.
int launch_mix_multi(srt_ctx* c, const srt_render_params* p, int n_buffers)
{
    const int n = (int)c->dev.size();
    if (p->layout != SRT_LAYOUT_FRAME || p->world_size > 1 || p->rank != 0)
        return fail(c, SRT_E_INVALID, "the mix-buffer technique renders whole frames (SRT_LAYOUT_FRAME, world_size 1)");
    if (p->mode != SRT_MODE_COLOR) return fail(c, SRT_E_INVALID, "the mix-buffer technique calls GetRayColor (SRT_MODE_COLOR)");
    const int64_t px = output_pixels(p);
    if (px < 0) { RenderArgs ra; return make_render_args(p, c->cam, ra, c->error); }
    MultiPlan plan;
    plan.n = n;
    plan.stride = px;
    plan.frame_pixels = px;
    HipDevices backend{c, false};
    for (int k = 0; k < n; k++) plan.pixels[k] = px;
    int rc = backend.prepare(plan);
    DeviceState& root = c->dev[0];
    for (int b0 = 0; b0 < n_buffers && rc == SRT_OK; b0 += n) {
        for (int k = 0; k < n && rc == SRT_OK; k++) {
            plan.pixels[k] = b0 + k < n_buffers ? px : 0;
            if (plan.pixels[k] == 0) continue;
            DeviceState& d = c->dev[k];
            // the first device keeps the buffers of a round side by side; nobody writes there before it has mixed the last round
            if (k > 0 && b0 > 0) { SRT_HIP(c, hipSetDevice(d.device)); SRT_HIP(c, hipStreamWaitEvent(d.stream, c->ev_assembled, 0)); }
            // renderBlockFull (Game1.cs:320-374) for one buffer: sample index b of the counter RNG stands for thread b's
            // System.Random; the colour is divided by `samples`, then gamma and packing
            PassSpec buffer;
            buffer.first_sample = b0 + k;
            rc = launch(c, d, p, k == 0 ? c->d_gathered : d.d_out, (size_t)px, d.stream, false, 1, 0, nullptr, &buffer);
        }
        if (rc == SRT_OK) rc = backend.gather(plan);
        if (rc != SRT_OK) break;
        SRT_HIP(c, hipSetDevice(root.device));
        for (int k = 0; k < n; k++) {
            if (plan.pixels[k] == 0) continue;
            hipError_t e = launch_mix(c->d_frame, c->d_gathered + (size_t)k * (size_t)px, (uint32_t)px, b0 + k == 0, root.stream);
            if (e != hipSuccess) return hip_fail(c, e, "mix kernel launch");
        }
        SRT_HIP(c, hipEventRecord(c->ev_assembled, root.stream));
        c->multi_pending = true;
    }
    return rc;
}

// A progressive render on the devices of a context: every pass is traced by all devices (each its own tiles, with running
// sums of its own), and after every pass the tiles are gathered and assembled on the first device and the preview goes to
// pinned host memory, with an event behind it.  The passes are sized for the first device, which owns the most tiles.
int launch_progressive_multi(srt_ctx* c, const srt_render_params* p, int min_passes)
{
    MultiPlan plan;
    int rc = plan_multi(p, (int)c->dev.size(), plan, c->error);
    if (rc != SRT_OK) return rc;
    HipDevices backend{c, false};
    if ((rc = backend.prepare(plan)) != SRT_OK) return rc;
    RenderArgs ra0;
    if ((rc = make_render_args(&plan.part[0], c->cam, ra0, c->error)) != SRT_OK) return rc;
    const uint64_t samples = (uint64_t)p->samples, per_sample = (uint64_t)(ra0.n_work ? ra0.n_work : 1u) * 12u;
    const uint64_t by_items = (((uint64_t)1 << 31) - 1) / (ra0.n_work ? ra0.n_work : 1u);
    uint64_t per_pass = c->sample_budget / per_sample;
    if (per_pass > by_items) per_pass = by_items;
    if (per_pass < 1) per_pass = 1;
    const uint64_t share = (samples + (uint64_t)min_passes - 1) / (uint64_t)min_passes;
    if (per_pass > share) per_pass = share;
    const uint64_t floor_pp = (samples + srt_ctx::MAX_PASSES - 1) / srt_ctx::MAX_PASSES;   // never more passes than events
    if (per_pass < floor_pp) {
        if (floor_pp * per_sample > c->sample_budget || floor_pp > by_items) return fail(c, SRT_E_LIMIT, "a progressive render of this size needs more than 64 passes");
        per_pass = floor_pp;
    }
    DeviceState& root = c->dev[0];
    const size_t frame_bytes = (size_t)plan.frame_pixels * sizeof(uint32_t);
    int pass = 0;
    for (uint64_t s0 = 0; s0 < samples; s0 += per_pass, pass++) {
        PassSpec spec;
        spec.first_sample = (int)s0;
        spec.count = (int)(samples - s0 < per_pass ? samples - s0 : per_pass);
        spec.first = s0 == 0;
        spec.last = s0 + (uint64_t)spec.count >= samples;
        spec.preview = (float)samples / (float)(s0 + (uint64_t)spec.count);
        for (int k = 0; k < plan.n; k++) {
            DeviceState& d = c->dev[k];
            if (plan.pixels[k] == 0) continue;
            // the first device keeps every device's tiles side by side: nobody writes there before it has assembled the last pass
            if (k > 0 && pass > 0) { SRT_HIP(c, hipSetDevice(d.device)); SRT_HIP(c, hipStreamWaitEvent(d.stream, c->ev_assembled, 0)); }
            rc = launch(c, d, &plan.part[k], k == 0 ? c->d_gathered : d.d_out, (size_t)plan.pixels[k], d.stream, false, 1, 0, nullptr, &spec);
            if (rc != SRT_OK) return rc;
        }
        if ((rc = backend.gather(plan)) != SRT_OK) return rc;
        if ((rc = backend.assemble(plan)) != SRT_OK) return rc;
        SRT_HIP(c, hipSetDevice(root.device));
        SRT_HIP(c, hipMemcpyAsync(c->h_pinned, c->d_frame, frame_bytes, hipMemcpyDeviceToHost, root.stream));
        if (!c->pass_done[pass]) SRT_HIP(c, hipEventCreateWithFlags(&c->pass_done[pass], hipEventDisableTiming));
        SRT_HIP(c, hipEventRecord(c->pass_done[pass], root.stream));
    }
    c->passes_total = pass;
    return SRT_OK;
}

int render_blocking(srt_ctx* c, const srt_render_params* p, uint32_t* rgba_out, bool counted, int mix_buffers = 0)
{
    if (!c) return SRT_E_INVALID;
    if (!rgba_out) return fail(c, SRT_E_INVALID, "rgba_out is NULL");
    if (c->async_pending) return fail(c, SRT_E_STATE, "an asynchronous render is in flight");
    int64_t n = output_pixels(p);
    if (n < 0) { RenderArgs ra; return make_render_args(p, c->cam, ra, c->error); }
    DeviceState& d = c->dev[0];
    SRT_HIP(c, hipSetDevice(d.device));
    int rc = collect_stats(c);  // a blocking render reports its own counters only
    if (rc != SRT_OK) return rc;
    const uint32_t* src;
    if (c->dev.size() > 1 && mix_buffers > 0) {
        if ((rc = launch_mix_multi(c, p, mix_buffers)) != SRT_OK) return rc;
        src = c->d_frame;
    } else if (c->dev.size() > 1) {
        MultiPlan plan;
        if ((rc = launch_multi(c, p, counted, plan)) != SRT_OK) return rc;
        src = c->d_frame;
    } else {
        if ((rc = ensure_pixels(c, &d.d_out, &d.out_capacity, (size_t)n)) != SRT_OK) return rc;
        if ((rc = launch(c, d, p, d.d_out, d.out_capacity, d.stream, counted, 1, mix_buffers, nullptr, nullptr, true)) != SRT_OK) return rc;
        src = d.d_out;
    }
    SRT_HIP(c, hipSetDevice(d.device));
    SRT_HIP(c, hipMemcpyAsync(rgba_out, src, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, d.stream));
    SRT_HIP(c, hipStreamSynchronize(d.stream));
    return collect_stats(c);
}

int create_device(DeviceState& d, int ordinal)
{
    d.device = ordinal;
    // The frame slab is the one large standing allocation of a device (320 MiB for the largest grid).  SRT_FRAME_SLAB_MB
    // bounds it, for hosts that share the GPU with a display: the persistent grid then has fewer blocks (256 KiB each).
    d.slab_bytes = trace_frame_slab_bytes();
    if (const char* mb = std::getenv("SRT_FRAME_SLAB_MB")) {
        const long v = std::atol(mb);
        if (v > 0 && ((size_t)v << 20) < d.slab_bytes) d.slab_bytes = (size_t)v << 20;
        if (d.slab_bytes < 16 * trace_frame_slab_block_bytes()) d.slab_bytes = 16 * trace_frame_slab_block_bytes();
    }
    bool ok = hipSetDevice(ordinal) == hipSuccess && hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&d.ev_begin) == hipSuccess && hipEventCreate(&d.ev_end) == hipSuccess &&
              hipEventCreateWithFlags(&d.ev_tiles, hipEventDisableTiming) == hipSuccess &&
              hipMalloc((void**)&d.d_stats, kStatsWords * sizeof(unsigned long long)) == hipSuccess &&
              hipMalloc((void**)&d.d_counter, sizeof(uint32_t)) == hipSuccess &&
#ifdef SRT_TIMELINE
              hipMalloc((void**)&d.d_timeline, trace_timeline_bytes()) == hipSuccess &&
#endif
              hipMalloc(&d.d_frame_slab, d.slab_bytes) == hipSuccess &&
              hipMemset(d.d_stats, 0, kStatsWords * sizeof(unsigned long long)) == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
              static_scratch_setup(ordinal, &d.static_scratch) == hipSuccess;
    return ok ? SRT_OK : SRT_E_HIP;
}

void destroy_device(DeviceState& d)
{
    hipSetDevice(d.device);
    if (d.stream) hipStreamSynchronize(d.stream);
    free_scene_buffers(d);
    if (d.d_out) hipFree(d.d_out);
    if (d.d_stats) hipFree(d.d_stats);
    if (d.d_counter) hipFree(d.d_counter);
    if (d.d_timeline) hipFree(d.d_timeline);
    if (d.wavefront.mem) hipFree(d.wavefront.mem);
    if (d.wavefront.far_stacks) hipFree(d.wavefront.far_stacks);
    if (d.d_frame_slab) hipFree(d.d_frame_slab);
    if (d.d_samples) hipFree(d.d_samples);
    if (d.d_running) hipFree(d.d_running);
    if (d.ev_begin) hipEventDestroy(d.ev_begin);
    if (d.ev_end) hipEventDestroy(d.ev_end);
    if (d.ev_tiles) hipEventDestroy(d.ev_tiles);
    if (d.stream) hipStreamDestroy(d.stream);
}

}  // namespace

extern "C" {

int srt_version(void) { return SRT_VERSION; }

const char* srt_last_error(const srt_ctx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int srt_create(const srt_config* cfg, srt_ctx** out)
{
    if (!out) return fail(nullptr, SRT_E_INVALID, "out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return fail(nullptr, SRT_E_NO_DEVICE, "no HIP device is visible; this library has no CPU path");
    // the device list
    std::vector<int> ordinals;
    int gather = SRT_GATHER_RCCL;
    const int n_cfg = cfg ? cfg->n_devices : 0;
    if (n_cfg == 0) ordinals.push_back(cfg ? cfg->device : 0);
    else if (n_cfg == SRT_ALL_DEVICES) {
        if (count > SRT_MAX_DEVICES) count = SRT_MAX_DEVICES;
        for (int i = 0; i < count; i++) ordinals.push_back(i);
    } else if (n_cfg >= 1 && n_cfg <= SRT_MAX_DEVICES) ordinals.assign(cfg->devices, cfg->devices + n_cfg);
    else return fail(nullptr, SRT_E_INVALID, "n_devices must be 0, SRT_ALL_DEVICES or 1..SRT_MAX_DEVICES");
    if (n_cfg != 0) {
        gather = cfg->gather;
        if (gather != SRT_GATHER_RCCL && gather != SRT_GATHER_PEER_COPY) return fail(nullptr, SRT_E_INVALID, "unknown gather kind");
        if (const char* g = std::getenv("SRT_GATHER")) {   // experiments: force one transport
            if (!std::strcmp(g, "peer")) gather = SRT_GATHER_PEER_COPY;
            else if (!std::strcmp(g, "rccl")) gather = SRT_GATHER_RCCL;
        }
    }
    bool distinct = true;
    for (size_t i = 0; i < ordinals.size(); i++) {
        if (ordinals[i] < 0 || ordinals[i] >= count) return fail(nullptr, SRT_E_INVALID, "device ordinal out of range");
        for (size_t j = 0; j < i; j++) distinct = distinct && ordinals[i] != ordinals[j];
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, ordinals[i]);
        if (e != hipSuccess) return hip_fail(nullptr, e, "hipGetDeviceProperties");
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(nullptr, SRT_E_NO_DEVICE, std::string("kernels are built for gfx950 only, device is ") + prop.gcnArchName);
    }
    srt_ctx* c = new srt_ctx();
    c->dev.resize(ordinals.size());
    for (size_t i = 0; i < ordinals.size(); i++) {
        if (create_device(c->dev[i], ordinals[i]) != SRT_OK) {
            srt_destroy(c);
            return fail(nullptr, SRT_E_HIP, "could not create stream / events / counters on the device");
        }
    }
    // RCCL refuses a communicator with one GPU in two ranks, and one device has nothing to gather
    c->gather = (ordinals.size() > 1 && distinct) ? gather : SRT_GATHER_PEER_COPY;
    // SRT_RCCL_SINGLE=1 (tests on a box with one GPU): a list of one device makes its one-rank communicator all the same,
    // which exercises loading librccl, ncclCommInitAll and the rank count check; there is still nothing to gather
    if (ordinals.size() == 1 && n_cfg != 0 && gather == SRT_GATHER_RCCL && std::getenv("SRT_RCCL_SINGLE")) c->gather = SRT_GATHER_RCCL;
    if (ordinals.size() > 1) {
        hipSetDevice(ordinals[0]);
        if (hipEventCreate(&c->ev_own_tiles) != hipSuccess || hipEventCreate(&c->ev_gathered) != hipSuccess || hipEventCreate(&c->ev_assembled) != hipSuccess) {
            srt_destroy(c);
            return fail(nullptr, SRT_E_HIP, "could not create the events of the gather");
        }
        // direct xGMI access between the first device and the others, where the platform offers it
        for (size_t i = 1; i < ordinals.size(); i++) {
            if (ordinals[i] == ordinals[0]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, ordinals[i], ordinals[0]) == hipSuccess && can) { hipSetDevice(ordinals[i]); (void)hipDeviceEnablePeerAccess(ordinals[0], 0); }
            if (hipDeviceCanAccessPeer(&can, ordinals[0], ordinals[i]) == hipSuccess && can) { hipSetDevice(ordinals[0]); (void)hipDeviceEnablePeerAccess(ordinals[i], 0); }
        }
        (void)hipGetLastError();   // "peer access already enabled" is not an error here
    }
    if (c->gather == SRT_GATHER_RCCL) {
        std::string err;
        if (!g_rccl.load(err)) { srt_destroy(c); return fail(nullptr, SRT_E_HIP, err + " (gather = SRT_GATHER_PEER_COPY needs no RCCL)"); }
        ncclResult_t r = g_rccl.CommInitAll(c->comms, (int)ordinals.size(), ordinals.data());
        if (r != ncclSuccess) {
            std::string msg = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r);
            for (ncclComm_t& k : c->comms) k = nullptr;
            srt_destroy(c);
            return fail(nullptr, SRT_E_HIP, msg);
        }
        int ranks = 0;
        r = g_rccl.CommCount(c->comms[0], &ranks);
        c->rccl_ranks = ranks;
        if (r != ncclSuccess || ranks != (int)ordinals.size()) {
            srt_destroy(c);
            return fail(nullptr, SRT_E_HIP, "the RCCL communicator does not have one rank per device of the context");
        }
    }
    if (const char* mb = std::getenv("SRT_SAMPLE_BUFFER_MB")) {
        long v = std::atol(mb);
        if (v > 0) c->sample_budget = (uint64_t)v << 20;
    }
    hipSetDevice(ordinals[0]);
    *out = c;
    return SRT_OK;
}

void srt_destroy(srt_ctx* c)
{
    if (!c) return;
    for (DeviceState& d : c->dev) if (d.stream) { hipSetDevice(d.device); hipStreamSynchronize(d.stream); }
    for (ncclComm_t k : c->comms) if (k) g_rccl.CommDestroy(k);
    if (!c->dev.empty()) {
        hipSetDevice(c->dev[0].device);
        if (c->d_gathered) hipFree(c->d_gathered);
        if (c->d_frame) hipFree(c->d_frame);
        if (c->d_mix) hipFree(c->d_mix);
        if (c->d_post[0]) hipFree(c->d_post[0]);
        if (c->d_post[1]) hipFree(c->d_post[1]);
        if (c->d_post_progress) hipFree(c->d_post_progress);
        if (c->h_pinned) hipHostFree(c->h_pinned);
        for (hipEvent_t e : c->pass_done) if (e) hipEventDestroy(e);
        for (hipEvent_t e : {c->ev_own_tiles, c->ev_gathered, c->ev_assembled}) if (e) hipEventDestroy(e);
    }
    for (DeviceState& d : c->dev) destroy_device(d);
    delete c;
}

int srt_get_device_info(const srt_ctx* c, srt_device_info* out)
{
    if (!c || !out) return SRT_E_INVALID;
    std::memset(out, 0, sizeof(*out));
    out->n_devices = (int32_t)c->dev.size();
    for (size_t i = 0; i < c->dev.size(); i++) out->devices[i] = c->dev[i].device;
    out->gather = c->gather;
    out->rccl_ranks = c->rccl_ranks;
    return SRT_OK;
}

int srt_set_camera(srt_ctx* c, const float position[3], const float lower_left[3], const float horizontal[3], const float vertical[3])
{
    if (!c) return SRT_E_INVALID;
    if (!position || !lower_left || !horizontal || !vertical) return fail(c, SRT_E_INVALID, "camera vector is NULL");
    c->cam.position = mk3(position[0], position[1], position[2]);
    c->cam.lower_left = mk3(lower_left[0], lower_left[1], lower_left[2]);
    c->cam.horizontal = mk3(horizontal[0], horizontal[1], horizontal[2]);
    c->cam.vertical = mk3(vertical[0], vertical[1], vertical[2]);
    c->has_camera = true;
    return SRT_OK;
}

int srt_set_materials(srt_ctx* c, const srt_material* m, uint32_t n)
{
    if (!c) return SRT_E_INVALID;
    c->committed = false;
    return host_set_materials(c->host, m, n, c->error);
}
int srt_set_spheres(srt_ctx* c, const float* cx, const float* cy, const float* cz, const float* r, const uint32_t* mat, uint32_t n)
{
    if (!c) return SRT_E_INVALID;
    c->committed = false;
    return host_set_spheres(c->host, cx, cy, cz, r, mat, n, c->error);
}
int srt_set_rects(srt_ctx* c, const float* x1, const float* x2, const float* y1, const float* y2, const float* k, const uint32_t* mat, uint32_t n)
{
    if (!c) return SRT_E_INVALID;
    c->committed = false;
    return host_set_rects(c->host, x1, x2, y1, y2, k, mat, n, c->error);
}
int srt_set_triangles(srt_ctx* c, const float* pos9, const float* nrm9, const uint32_t* mat, const uint8_t* flags, uint32_t n)
{
    if (!c) return SRT_E_INVALID;
    c->committed = false;
    return host_set_triangles(c->host, pos9, nrm9, mat, flags, n, c->error);
}
int srt_set_lights(srt_ctx* c, const srt_point_light* l, uint32_t n)
{
    if (!c) return SRT_E_INVALID;
    c->committed = false;
    return host_set_lights(c->host, l, n, c->error);
}

int srt_set_surface_order(srt_ctx* c, const uint8_t* type, const uint32_t* index, uint32_t n)
{
    if (!c) return SRT_E_INVALID;
    c->committed = false;
    return host_set_surface_order(c->host, type, index, n, c->error);
}

int srt_commit_scene(srt_ctx* c, int world_kind)
{
    if (!c) return SRT_E_INVALID;
    if (c->async_pending) return fail(c, SRT_E_STATE, "an asynchronous render is in flight");
    auto t0 = std::chrono::steady_clock::now();
    c->committed = false;
    int rc = flatten_scene(c->host, world_kind, c->flat, c->error);
    if (rc != SRT_OK) return rc;
    // one replica per device: no scene traffic between devices while tracing
    for (DeviceState& d : c->dev) {
        SRT_HIP(c, hipSetDevice(d.device));
        SRT_HIP(c, hipStreamSynchronize(d.stream));
        free_scene_buffers(d);
        d.walk_tune = WALK_TUNE_CALIBRATE;
        d.wavefront.nodes_per_item = 0.0;
        d.rays_per_item = 0.0;
        d.items_since_collect = 0;
        d.largest_tree = 0;
        if ((rc = upload(c, c->flat.nodes, &d.d_nodes)) != SRT_OK) return rc;
        if ((rc = upload(c, c->flat.prims, &d.d_prims)) != SRT_OK) return rc;
        if ((rc = upload(c, c->flat.normals, &d.d_normals)) != SRT_OK) return rc;
        if ((rc = upload(c, c->flat.materials, &d.d_materials)) != SRT_OK) return rc;
        if ((rc = upload(c, c->flat.lights, &d.d_lights)) != SRT_OK) return rc;
        if ((rc = upload(c, c->flat.tiny, &d.d_tiny)) != SRT_OK) return rc;
    }
    c->world_kind = world_kind;
    c->committed = true;
    c->stats.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    c->stats.bvh_nodes = c->flat.n_tree_nodes;
    c->stats.bvh_depth = c->flat.bvh_depth;
    c->stats.n_prims = c->flat.n_prims;
    // the host copies of the big arrays are no longer needed
    std::vector<q4>().swap(c->flat.nodes);
    std::vector<q4>().swap(c->flat.prims);
    std::vector<q4>().swap(c->flat.normals);
    return SRT_OK;
}

int srt_render(srt_ctx* c, const srt_render_params* p, uint32_t* rgba_out) { return render_blocking(c, p, rgba_out, false); }
int srt_render_counted(srt_ctx* c, const srt_render_params* p, uint32_t* rgba_out) { return render_blocking(c, p, rgba_out, true); }

int srt_render_device(srt_ctx* c, const srt_render_params* p, void* d_rgba_out, size_t capacity_pixels, void* hip_stream)
{
    if (!c) return SRT_E_INVALID;
    return launch(c, c->dev[0], p, (uint32_t*)d_rgba_out, capacity_pixels, (hipStream_t)hip_stream, false);
}

// The workspace comes out of the device's static array instead of hipMalloc: [work counter | running sums | samples].
// Nothing on this path allocates: the events, the counters and the array's address were made by srt_create.
int srt_render_device_static(srt_ctx* c, const srt_render_params* p, void* d_rgba_out, size_t capacity_pixels, void* hip_stream)
{
    if (!c) return SRT_E_INVALID;
    DeviceState& d = c->dev[0];
    if (!d.static_scratch) return fail(c, SRT_E_STATE, "the static workspace was not found on this device");
    RenderArgs ra;
    int rc = make_render_args(p, c->cam, ra, c->error);
    if (rc != SRT_OK) return rc;
    const size_t kHead = 512;   // the work counter, padded
    const size_t per_sample = (size_t)ra.n_work * 12u;
    const size_t room = (size_t)SRT_STATIC_SCRATCH_BYTES - kHead;
    Workspace ws;
    ws.fixed = true;
    ws.counter = reinterpret_cast<uint32_t*>(d.static_scratch);
    if (per_sample * (size_t)ra.samples <= room) {
        // every sample of the render in one pass: no running sums
        ws.samples = reinterpret_cast<float*>(d.static_scratch + kHead);
        ws.samples_bytes = room;
    } else {
        const size_t running = (per_sample + 255u) & ~(size_t)255u;
        if (running + per_sample > room) {
            char msg[160];
            std::snprintf(msg, sizeof msg, "the static workspace holds renders of up to %zu pixels per rank (24 bytes each), this one has %u",
                          (size_t)(room / 24u), ra.n_work);
            return fail(c, SRT_E_LIMIT, msg);
        }
        ws.running = reinterpret_cast<float*>(d.static_scratch + kHead);
        ws.running_bytes = running;
        ws.samples = reinterpret_cast<float*>(d.static_scratch + kHead + running);
        ws.samples_bytes = room - running;
    }
    return launch(c, d, p, (uint32_t*)d_rgba_out, capacity_pixels, (hipStream_t)hip_stream, false, 1, 0, &ws);
}

static int start_async(srt_ctx* c, const srt_render_params* p, uint32_t* rgba_out, int passes)
{
    if (!c) return SRT_E_INVALID;
    if (!rgba_out) return fail(c, SRT_E_INVALID, "rgba_out is NULL");
    if (c->async_pending) return fail(c, SRT_E_STATE, "an asynchronous render is already in flight");
    int64_t n = output_pixels(p);
    if (n < 0) { RenderArgs ra; return make_render_args(p, c->cam, ra, c->error); }
    DeviceState& d = c->dev[0];
    SRT_HIP(c, hipSetDevice(d.device));
    if ((size_t)n > c->pinned_capacity) {
        if (c->h_pinned) hipHostFree(c->h_pinned);
        c->h_pinned = nullptr; c->pinned_capacity = 0;
        SRT_HIP(c, hipHostMalloc((void**)&c->h_pinned, (size_t)n * sizeof(uint32_t), hipHostMallocDefault));
        c->pinned_capacity = (size_t)n;
    }
    int rc;
    const uint32_t* src;
    if (c->dev.size() > 1 && passes > 1) {
        if ((rc = launch_progressive_multi(c, p, passes)) != SRT_OK) return rc;
        src = c->d_frame;
        SRT_HIP(c, hipSetDevice(d.device));
    } else if (c->dev.size() > 1) {
        MultiPlan plan;
        if ((rc = launch_multi(c, p, false, plan)) != SRT_OK) return rc;
        src = c->d_frame;
        SRT_HIP(c, hipSetDevice(d.device));
    } else {
        if ((rc = ensure_pixels(c, &d.d_out, &d.out_capacity, (size_t)n)) != SRT_OK) return rc;
        if (passes <= 1 && wants_probe(c, d, p)) {
            // the first sample now; the others when srt_poll / srt_wait find it done
            const PassSpec probe{0, 1, true, false, 0.0f};
            if ((rc = launch(c, d, p, d.d_out, d.out_capacity, d.stream, false, 1, 0, nullptr, &probe)) != SRT_OK) return rc;
            c->async_probing = true;
            c->async_params = *p;
            c->async_dst = rgba_out;
            c->async_pixels = (size_t)n;
            c->async_pending = true;
            return SRT_OK;
        }
        if ((rc = launch(c, d, p, d.d_out, d.out_capacity, d.stream, false, passes)) != SRT_OK) return rc;
        src = d.d_out;
    }
    // ev_end marks the end of the kernels; the copy to pinned memory follows it on the same stream (a progressive
    // render has already queued that copy after its last pass)
    if (c->passes_total == 0) SRT_HIP(c, hipMemcpyAsync(c->h_pinned, src, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, d.stream));
    c->async_dst = rgba_out;
    c->async_pixels = (size_t)n;
    c->async_pending = true;
    return SRT_OK;
}

// The probe sample of an asynchronous render is done (the caller has seen the stream idle): its ray count, then the
// other samples and the copy of the frame.
static int continue_async(srt_ctx* c)
{
    DeviceState& d = c->dev[0];
    SRT_HIP(c, hipSetDevice(d.device));
    unsigned long long seen[kStatsWords];
    SRT_HIP(c, hipMemcpy(seen, d.d_stats, sizeof(seen), hipMemcpyDeviceToHost));
    if (d.items_since_collect != 0) d.rays_per_item = (double)seen[0] / (double)d.items_since_collect;
    if (seen[kStatsLargestTree] > d.largest_tree) d.largest_tree = seen[kStatsLargestTree];
    c->async_probing = false;
    const PassSpec rest{1, c->async_params.samples - 1, false, true, 0.0f};
    int rc = launch(c, d, &c->async_params, d.d_out, d.out_capacity, d.stream, false, 1, 0, nullptr, &rest);
    if (rc != SRT_OK) { c->async_pending = false; return rc; }
    SRT_HIP(c, hipMemcpyAsync(c->h_pinned, d.d_out, c->async_pixels * sizeof(uint32_t), hipMemcpyDeviceToHost, d.stream));
    return SRT_OK;
}

int srt_render_async(srt_ctx* c, const srt_render_params* p, uint32_t* rgba_out) { return start_async(c, p, rgba_out, 1); }

int srt_render_progressive(srt_ctx* c, const srt_render_params* p, uint32_t* rgba_out, int passes)
{
    if (c && (passes < 1 || passes > srt_ctx::MAX_PASSES)) return fail(c, SRT_E_INVALID, "passes must be 1..64");
    return start_async(c, p, rgba_out, passes);
}

int srt_progress(srt_ctx* c, int* passes_done, int* passes_total)
{
    if (!c) return SRT_E_INVALID;
    if (!passes_done || !passes_total) return fail(c, SRT_E_INVALID, "output pointer is NULL");
    *passes_total = c->passes_total;
    int n = 0;
    while (n < c->passes_total && hipEventQuery(c->pass_done[n]) == hipSuccess) n++;
    *passes_done = n;
    return SRT_OK;
}

int srt_read_preview(srt_ctx* c, uint32_t* rgba_out)
{
    if (!c) return SRT_E_INVALID;
    if (!rgba_out) return fail(c, SRT_E_INVALID, "rgba_out is NULL");
    if (!c->async_pending || c->passes_total == 0) return fail(c, SRT_E_STATE, "no progressive render is in flight");
    if (hipEventQuery(c->pass_done[0]) != hipSuccess) return fail(c, SRT_E_STATE, "no pass has finished yet");
    std::memcpy(rgba_out, c->h_pinned, c->async_pixels * sizeof(uint32_t));
    return SRT_OK;
}

static int finish_async(srt_ctx* c)
{
    std::memcpy(c->async_dst, c->h_pinned, c->async_pixels * sizeof(uint32_t));
    c->async_pending = false;
    return collect_stats(c);
}

// dev[0]'s stream has everything behind it: the other devices' tiles arrive there before the frame is assembled
int srt_poll(srt_ctx* c, int* done)
{
    if (!c) return SRT_E_INVALID;
    if (!done) return fail(c, SRT_E_INVALID, "done is NULL");
    if (!c->async_pending) { *done = 1; return SRT_OK; }
    hipError_t e = hipStreamQuery(c->dev[0].stream);
    if (e == hipErrorNotReady) { *done = 0; return SRT_OK; }
    if (e != hipSuccess) return hip_fail(c, e, "hipStreamQuery");
    if (c->async_probing) { *done = 0; return continue_async(c); }
    *done = 1;
    return finish_async(c);
}

int srt_wait(srt_ctx* c)
{
    if (!c) return SRT_E_INVALID;
    if (!c->async_pending) return SRT_OK;
    SRT_HIP(c, hipStreamSynchronize(c->dev[0].stream));
    if (c->async_probing) {
        int rc = continue_async(c);
        if (rc != SRT_OK) return rc;
        SRT_HIP(c, hipStreamSynchronize(c->dev[0].stream));
    }
    return finish_async(c);
}

int srt_get_stats(srt_ctx* c, srt_stats* out)
{
    if (!c) return SRT_E_INVALID;
    if (!out) return fail(c, SRT_E_INVALID, "out is NULL");
    int rc = collect_stats(c);
    if (rc != SRT_OK) return rc;
    *out = c->stats;
    return SRT_OK;
}

int srt_get_device_stats(srt_ctx* c, int32_t index, srt_stats* out)
{
    if (!c) return SRT_E_INVALID;
    if (!out) return fail(c, SRT_E_INVALID, "out is NULL");
    if (index < 0 || (size_t)index >= c->dev.size()) return fail(c, SRT_E_INVALID, "device index out of range");
    *out = c->dev[(size_t)index].part;
    out->build_ms = c->stats.build_ms; out->bvh_nodes = c->stats.bvh_nodes; out->bvh_depth = c->stats.bvh_depth; out->n_prims = c->stats.n_prims;
    return SRT_OK;
}

int64_t srt_local_pixels(const srt_render_params* p) { return output_pixels(p); }

int srt_assemble_tiles(srt_ctx* c, const srt_render_params* p, const void* d_gathered, size_t stride_pixels, void* d_frame, void* hip_stream)
{
    if (!c) return SRT_E_INVALID;
    if (!p) return fail(c, SRT_E_INVALID, "params is NULL");
    if (!d_gathered || !d_frame) return fail(c, SRT_E_INVALID, "device pointer is NULL");
    RenderArgs ra;
    Camera none{};
    srt_render_params q = *p;
    q.rank = 0;
    q.layout = SRT_LAYOUT_TILES;
    int rc = make_render_args(&q, none, ra, c->error);
    if (rc != SRT_OK) return rc;
    if ((int64_t)stride_pixels < output_pixels(&q)) return fail(c, SRT_E_INVALID, "stride_pixels smaller than rank 0's tile buffer");
    SRT_HIP(c, hipSetDevice(c->dev[0].device));
    hipStream_t s = (hipStream_t)hip_stream;
    hipError_t e = launch_assemble(ra, (const uint32_t*)d_gathered, stride_pixels, (uint32_t*)d_frame, s);
    if (e != hipSuccess) return hip_fail(c, e, "assemble kernel launch");
    return SRT_OK;
}

}  // extern "C"

namespace {

int post_filter(srt_ctx* c, int filter, uint32_t* array, int32_t width, int32_t height, float weight, float param, int32_t passes)
{
    if (!c) return SRT_E_INVALID;
    if (!array) return fail(c, SRT_E_INVALID, "array is NULL");
    if (width < 1 || height < 1) return fail(c, SRT_E_INVALID, "width and height must be positive");
    if (passes < 1) return fail(c, SRT_E_INVALID, "passes must be at least 1");
    if ((int64_t)width * (int64_t)height > ((int64_t)1 << 31) - 1) return fail(c, SRT_E_LIMIT, "more than 2^31 pixels");
    if (c->async_pending) return fail(c, SRT_E_STATE, "an asynchronous render is in flight");
    DeviceState& dv = c->dev[0];
    SRT_HIP(c, hipSetDevice(dv.device));
    const size_t pixels = (size_t)width * (size_t)height, bytes = pixels * sizeof(uint32_t);
    const size_t progress_bytes = post_progress_bytes();
    // the two pictures and the progress words stay with the context: filtering one size again allocates nothing
    if (pixels > c->post_capacity || !c->d_post[0] || !c->d_post[1]) {
        SRT_HIP(c, hipDeviceSynchronize());
        for (uint32_t*& b : c->d_post) { if (b) (void)hipFree(b); b = nullptr; }
        c->post_capacity = 0;
        SRT_HIP(c, hipMalloc((void**)&c->d_post[0], bytes));
        SRT_HIP(c, hipMalloc((void**)&c->d_post[1], bytes));
        c->post_capacity = pixels;
    }
    if (!c->d_post_progress) SRT_HIP(c, hipMalloc((void**)&c->d_post_progress, progress_bytes));
    uint32_t* const* d = c->d_post;
    uint32_t alarm = 0u;
    hipError_t e = hipMemsetAsync(c->d_post_progress, 0, progress_bytes, dv.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d[0], array, bytes, hipMemcpyHostToDevice, dv.stream);
    int from = 0;
    for (int32_t i = 0; i < passes && e == hipSuccess; i++, from ^= 1)
        e = launch_post_filter(filter, d[from], d[from ^ 1], width, height, weight, param, c->d_post_progress, dv.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(array, d[from], bytes, hipMemcpyDeviceToHost, dv.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&alarm, reinterpret_cast<char*>(c->d_post_progress) + progress_bytes - sizeof(uint32_t), sizeof(uint32_t), hipMemcpyDeviceToHost, dv.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(dv.stream);
    if (e != hipSuccess) return hip_fail(c, e, "post-process filter");
    if (alarm != 0u) return fail(c, SRT_E_HIP, "post-process filter: a band of rows gave up waiting for the band above it; the picture is incomplete");
    return SRT_OK;
}

}  // namespace

extern "C" {

int srt_render_mix_buffers(srt_ctx* ctx, const srt_render_params* p, int32_t n_buffers, uint32_t* rgba_out)
{
    if (!ctx) return SRT_E_INVALID;
    if (n_buffers < 1 || n_buffers > 65536) return fail(ctx, SRT_E_INVALID, "n_buffers must be within 1..65536");
    return render_blocking(ctx, p, rgba_out, false, n_buffers);
}

int srt_blur_array(srt_ctx* ctx, uint32_t* array, int32_t width, int32_t height, float weight, int32_t passes)
{
    return post_filter(ctx, SRT_POST_BLUR, array, width, height, weight, 0.0f, passes);
}
int srt_contrast_blur_array(srt_ctx* ctx, uint32_t* array, int32_t width, int32_t height, float weight, float contrast, int32_t passes)
{
    return post_filter(ctx, SRT_POST_CONTRAST_BLUR, array, width, height, weight, contrast, passes);
}
int srt_despeckle(srt_ctx* ctx, uint32_t* array, int32_t width, int32_t height, float weight, float threshold, int32_t passes)
{
    return post_filter(ctx, SRT_POST_DESPECKLE, array, width, height, weight, threshold, passes);
}

}  // extern "C"
