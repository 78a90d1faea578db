// srt_api.cpp -- the C ABI of libsrt.so (include/srt.h) over the gfx950 kernels.
//
// There is no CPU path here: every render goes through launch_trace() on a gfx950 device, and
// srt_create fails when there is none.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/srt.h"
#include "srt_launch.h"
#include "srt_scene.h"

using namespace srt;

struct srt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    HostScene host;
    FlatScene flat;
    bool committed = false;
    int world_kind = SRT_WORLD_BVHNODE;
    Camera cam{};
    bool has_camera = false;
    q4 *d_nodes = nullptr, *d_prims = nullptr, *d_normals = nullptr, *d_materials = nullptr, *d_lights = nullptr, *d_tiny = nullptr;
    uint32_t* d_out = nullptr;
    size_t out_capacity = 0;  // pixels
    unsigned long long* d_stats = nullptr;
    uint32_t* d_counter = nullptr;        // work counter of the persistent trace kernel
    uint32_t* d_mix = nullptr;            // one buffer of the mix-buffer technique
    size_t mix_capacity = 0;              // in pixels
    float* d_timeline = nullptr;              // scratch of -DSRT_TIMELINE builds (per-wavefront clock records)
    float* d_samples = nullptr;           // per-(sample, pixel) colours of one pass, 12 bytes each
    size_t samples_capacity = 0;          // bytes
    float* d_running = nullptr;           // per-pixel running sums, only for renders that need several passes
    size_t running_capacity = 0;          // bytes
    uint64_t sample_budget = (uint64_t)2 << 30;  // bytes the sample buffer may take (SRT_SAMPLE_BUFFER_MB)
    bool stats_pending = false;
    srt_stats stats{};
    // asynchronous render (srt_render_async / srt_poll / srt_wait)
    uint32_t* h_pinned = nullptr;
    size_t pinned_capacity = 0;
    uint32_t* async_dst = nullptr;
    size_t async_pixels = 0;
    bool async_pending = false;
    // progressive render: one event per pass, recorded after the pass's preview reached pinned memory
    enum { MAX_PASSES = 64 };
    hipEvent_t pass_done[MAX_PASSES] = {};
    int passes_total = 0;
    std::string error;
};

namespace {

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

void free_scene_buffers(srt_ctx* c)
{
    hipFree(c->d_nodes); hipFree(c->d_prims); hipFree(c->d_normals); hipFree(c->d_materials); hipFree(c->d_lights); hipFree(c->d_tiny);
    c->d_nodes = c->d_prims = c->d_normals = c->d_materials = c->d_lights = c->d_tiny = nullptr;
}

int upload(srt_ctx* c, const std::vector<q4>& src, q4** dst)
{
    SRT_HIP(c, hipMalloc((void**)dst, src.size() * sizeof(q4)));
    SRT_HIP(c, hipMemcpy(*dst, src.data(), src.size() * sizeof(q4), hipMemcpyHostToDevice));
    return SRT_OK;
}

SceneView scene_view(const srt_ctx* c)
{
    return c->flat.view(c->d_nodes, c->d_prims, c->d_normals, c->d_materials, c->d_lights, c->d_tiny);
}

// Fetches (and zeroes) the counters accumulated by the launches since the last collection, once
// the most recent render has finished.  kernel_ms is the device time of that render (all its passes).
int collect_stats(srt_ctx* c)
{
    if (!c->stats_pending) return SRT_OK;
    SRT_HIP(c, hipEventSynchronize(c->ev_end));
    unsigned long long s[32];
    SRT_HIP(c, hipMemcpy(s, c->d_stats, sizeof(s), hipMemcpyDeviceToHost));
    SRT_HIP(c, hipMemsetAsync(c->d_stats, 0, sizeof(s), c->stream));
    SRT_HIP(c, hipStreamSynchronize(c->stream));
#ifdef SRT_TIMELINE
    if (const char* path = std::getenv("SRT_TIMELINE_FILE")) {
        std::vector<unsigned long long> tl(4096 * 4);
        hipMemcpy(tl.data(), reinterpret_cast<char*>(c->d_timeline) + srt::trace_timeline_offset(), tl.size() * 8, hipMemcpyDeviceToHost);
        if (FILE* f = std::fopen(path, "wb")) { std::fwrite(tl.data(), 8, tl.size(), f); std::fclose(f); }
    }
#endif
    float ms = 0.0f;
    SRT_HIP(c, hipEventElapsedTime(&ms, c->ev_begin, c->ev_end));
    c->stats.rays_path = s[0]; c->stats.rays_shadow = s[1]; c->stats.hit_nodes = s[2];
    c->stats.box_tests = s[3]; c->stats.sphere_tests = s[4]; c->stats.tri_tests = s[5]; c->stats.rect_tests = s[6];
    c->stats.trace_sections = s[7]; c->stats.trace_lanes = s[8]; c->stats.light_sections = s[9]; c->stats.light_lanes = s[10];
    for (int i = 0; i < 16; i++) c->stats.section_cycles[i] = s[16 + i];
    c->stats.kernel_ms = ms;
    c->stats_pending = false;
    if (s[11] != 0ull) return fail(c, SRT_E_HIP, "trace kernel: a wavefront gave up waiting for a ray subtree that another lane took over; the picture is incomplete");
    return SRT_OK;
}

// min_passes > 1 (progressive renders): at least that many passes, each followed by a preview in d_out, a copy of it
// to pinned memory and an event.
int launch(srt_ctx* c, const srt_render_params* p, uint32_t* d_out, size_t capacity, hipStream_t stream, bool counted, int min_passes = 1, int mix_buffers = 0)
{
    if (!c->committed) return fail(c, SRT_E_STATE, "srt_commit_scene has not been called");
    if (!c->has_camera) return fail(c, SRT_E_STATE, "srt_set_camera has not been called");
    RenderArgs ra;
    int rc = make_render_args(p, c->cam, ra, c->error);
    if (rc != SRT_OK) return rc;
    int64_t need = output_pixels(p);
    if (need < 0) return fail(c, (int)need, "bad parameters");
    if ((size_t)need > capacity) return fail(c, SRT_E_INVALID, "output buffer too small for this layout");
    if (!d_out) return fail(c, SRT_E_INVALID, "output pointer is NULL");
    SRT_HIP(c, hipSetDevice(c->device));
    // Samples per pass: as many as fit the sample buffer budget (12 bytes per pixel and sample) and the
    // 2^31 work-item limit of one launch.
    if (ra.n_work == 0u) {
        // a rank without tiles: nothing to launch, and the counters of the "render" are all zero
        c->stats.rays_path = c->stats.rays_shadow = c->stats.hit_nodes = 0;
        c->stats.box_tests = c->stats.sphere_tests = c->stats.tri_tests = c->stats.rect_tests = 0;
        c->stats.trace_sections = c->stats.trace_lanes = c->stats.light_sections = c->stats.light_lanes = 0;
        for (int i = 0; i < 16; i++) c->stats.section_cycles[i] = 0;
        c->stats.kernel_ms = 0.0f;
        return SRT_OK;
    }
    uint64_t per_sample = (uint64_t)ra.n_work * 12u;
    uint64_t fit = c->sample_budget / per_sample;
    uint64_t by_items = (((uint64_t)1 << 31) - 1) / ra.n_work;
    uint64_t per_pass = fit < by_items ? fit : by_items;
    if (per_pass < 1) per_pass = 1;
    if (per_pass > (uint64_t)ra.samples) per_pass = (uint64_t)ra.samples;
    if (mix_buffers > 0) {
        if (ra.layout != 0 || ra.world_size != 1) return fail(c, SRT_E_INVALID, "the mix-buffer technique renders whole frames (SRT_LAYOUT_FRAME, world_size 1)");
        if ((uint32_t)ra.mode != MODE_COLOR) return fail(c, SRT_E_INVALID, "the mix-buffer technique calls GetRayColor (SRT_MODE_COLOR)");
        per_pass = 1;
        if ((size_t)need > c->mix_capacity) {
            SRT_HIP(c, hipDeviceSynchronize());
            if (c->d_mix) (void)hipFree(c->d_mix);
            c->d_mix = nullptr; c->mix_capacity = 0;
            SRT_HIP(c, hipMalloc((void**)&c->d_mix, (size_t)need * sizeof(uint32_t)));
            c->mix_capacity = (size_t)need;
        }
    }
    const bool progressive = min_passes > 1;
    if (progressive) {
        uint64_t share = ((uint64_t)ra.samples + (uint64_t)min_passes - 1) / (uint64_t)min_passes;
        if (share < 1) share = 1;
        if (per_pass > share) per_pass = share;
        // never more passes than events: later passes grow instead
        uint64_t floor_pp = ((uint64_t)ra.samples + srt_ctx::MAX_PASSES - 1) / srt_ctx::MAX_PASSES;
        if (per_pass < floor_pp) {
            if (floor_pp * per_sample > c->sample_budget) return fail(c, SRT_E_LIMIT, "a progressive render of this size needs more than 64 passes");
            per_pass = floor_pp;
        }
    }
    if ((uint64_t)ra.n_work > (((uint64_t)1 << 31) - 1)) return fail(c, SRT_E_LIMIT, "more than 2^31 pixels per rank");
    size_t need_samples = (size_t)(per_pass * per_sample);
    if (need_samples > c->samples_capacity) {
        // (re)allocation waits for work in flight; steady-state renders of one size never get here
        SRT_HIP(c, hipDeviceSynchronize());
        if (c->d_samples) hipFree(c->d_samples);
        c->d_samples = nullptr; c->samples_capacity = 0;
        SRT_HIP(c, hipMalloc((void**)&c->d_samples, need_samples));
        c->samples_capacity = need_samples;
    }
    const bool multi_pass = per_pass < (uint64_t)ra.samples;
    if (multi_pass && (size_t)per_sample > c->running_capacity) {
        SRT_HIP(c, hipDeviceSynchronize());
        if (c->d_running) hipFree(c->d_running);
        c->d_running = nullptr; c->running_capacity = 0;
        SRT_HIP(c, hipMalloc((void**)&c->d_running, (size_t)per_sample));
        c->running_capacity = (size_t)per_sample;
    }
    SRT_HIP(c, hipEventRecord(c->ev_begin, stream));
    const SceneView view = scene_view(c);
    if (mix_buffers > 0) {
        // renderBlockFull (Game1.cs:320-374) once per buffer: one sample -- sample index b of the counter RNG stands for
        // thread b's System.Random -- divided by `samples`, gamma, packing; then the pairwise mix of Game1.cs:209-220
        for (int b = 0; b < mix_buffers; b++) {
            ra.sample_first = (uint32_t)b;
            ra.sample_count = 1u;
            ra.n_items = ra.n_work;
            hipError_t e = launch_trace(view, ra, c->world_kind == SRT_WORLD_BVHNODE, counted, c->d_samples, c->d_stats, c->d_counter, c->d_timeline, stream);
            if (e != hipSuccess) return hip_fail(c, e, "trace kernel launch");
            e = launch_resolve(ra, c->d_samples, c->d_running, c->d_mix, true, true, 0.0f, stream);
            if (e != hipSuccess) return hip_fail(c, e, "resolve kernel launch");
            e = launch_mix(d_out, c->d_mix, (uint32_t)need, b == 0, stream);
            if (e != hipSuccess) return hip_fail(c, e, "mix kernel launch");
        }
        c->passes_total = 0;
        SRT_HIP(c, hipEventRecord(c->ev_end, stream));
        c->stats_pending = true;
        return SRT_OK;
    }
    int pass = 0;
    for (uint32_t s0 = 0; s0 < (uint32_t)ra.samples; s0 += (uint32_t)per_pass, pass++) {
        ra.sample_first = s0;
        ra.sample_count = (uint32_t)ra.samples - s0 < (uint32_t)per_pass ? (uint32_t)ra.samples - s0 : (uint32_t)per_pass;
        ra.n_items = ra.n_work * ra.sample_count;
        hipError_t e = launch_trace(view, ra, c->world_kind == SRT_WORLD_BVHNODE, counted, c->d_samples, c->d_stats, c->d_counter, c->d_timeline, stream);
        if (e != hipSuccess) return hip_fail(c, e, "trace kernel launch");
        const uint32_t done = s0 + ra.sample_count;
        const float preview = progressive ? (float)ra.samples / (float)done : 0.0f;
        e = launch_resolve(ra, c->d_samples, c->d_running, d_out, s0 == 0, done >= (uint32_t)ra.samples, preview, stream);
        if (e != hipSuccess) return hip_fail(c, e, "resolve kernel launch");
        if (progressive) {
            SRT_HIP(c, hipMemcpyAsync(c->h_pinned, d_out, (size_t)need * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            if (!c->pass_done[pass]) SRT_HIP(c, hipEventCreateWithFlags(&c->pass_done[pass], hipEventDisableTiming));
            SRT_HIP(c, hipEventRecord(c->pass_done[pass], stream));
        }
    }
    c->passes_total = progressive ? pass : 0;
    SRT_HIP(c, hipEventRecord(c->ev_end, stream));
    c->stats_pending = true;
    return SRT_OK;
}

int ensure_out(srt_ctx* c, size_t pixels)
{
    if (pixels <= c->out_capacity) return SRT_OK;
    if (c->d_out) hipFree(c->d_out);
    c->d_out = nullptr; c->out_capacity = 0;
    SRT_HIP(c, hipMalloc((void**)&c->d_out, pixels * sizeof(uint32_t)));
    c->out_capacity = pixels;
    return SRT_OK;
}

int render_blocking(srt_ctx* c, const srt_render_params* p, uint32_t* rgba_out, bool counted, int mix_buffers = 0)
{
    if (!c) return SRT_E_INVALID;
    if (!rgba_out) return fail(c, SRT_E_INVALID, "rgba_out is NULL");
    if (c->async_pending) return fail(c, SRT_E_STATE, "an asynchronous render is in flight");
    int64_t n = output_pixels(p);
    if (n < 0) { RenderArgs ra; return make_render_args(p, c->cam, ra, c->error); }
    SRT_HIP(c, hipSetDevice(c->device));
    int rc = ensure_out(c, (size_t)n);
    if (rc != SRT_OK) return rc;
    rc = collect_stats(c);  // a blocking render reports its own counters only
    if (rc != SRT_OK) return rc;
    rc = launch(c, p, c->d_out, c->out_capacity, c->stream, counted, 1, mix_buffers);
    if (rc != SRT_OK) return rc;
    SRT_HIP(c, hipMemcpyAsync(rgba_out, c->d_out, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    SRT_HIP(c, hipStreamSynchronize(c->stream));
    return collect_stats(c);
}

}  // namespace

extern "C" {

int srt_version(void) { return SRT_VERSION; }

const char* srt_last_error(const srt_ctx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int srt_create(const srt_config* cfg, srt_ctx** out)
{
    if (!out) return fail(nullptr, SRT_E_INVALID, "out is NULL");
    *out = nullptr;
    int device = cfg ? cfg->device : 0;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return fail(nullptr, SRT_E_NO_DEVICE, "no HIP device is visible; this library has no CPU path");
    if (device < 0 || device >= count) return fail(nullptr, SRT_E_INVALID, "device ordinal out of range");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return hip_fail(nullptr, e, "hipGetDeviceProperties");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, SRT_E_NO_DEVICE, std::string("kernels are built for gfx950 only, device is ") + prop.gcnArchName);
    srt_ctx* c = new srt_ctx();
    c->device = device;
    bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&c->ev_begin) == hipSuccess && hipEventCreate(&c->ev_end) == hipSuccess &&
              hipMalloc((void**)&c->d_stats, 32 * sizeof(unsigned long long)) == hipSuccess &&
              hipMalloc((void**)&c->d_counter, sizeof(uint32_t)) == hipSuccess &&
#ifdef SRT_TIMELINE
              hipMalloc((void**)&c->d_timeline, trace_timeline_bytes()) == hipSuccess &&
#endif
              hipMemset(c->d_stats, 0, 32 * sizeof(unsigned long long)) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    if (!ok) {
        srt_destroy(c);
        return fail(nullptr, SRT_E_HIP, "could not create stream / events / counters on the device");
    }
    if (const char* mb = std::getenv("SRT_SAMPLE_BUFFER_MB")) {
        long v = std::atol(mb);
        if (v > 0) c->sample_budget = (uint64_t)v << 20;
    }
    *out = c;
    return SRT_OK;
}

void srt_destroy(srt_ctx* c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    free_scene_buffers(c);
    if (c->d_out) hipFree(c->d_out);
    if (c->d_stats) hipFree(c->d_stats);
    if (c->d_counter) hipFree(c->d_counter);
    if (c->d_timeline) hipFree(c->d_timeline);
    if (c->d_mix) hipFree(c->d_mix);
    if (c->d_samples) hipFree(c->d_samples);
    if (c->d_running) hipFree(c->d_running);
    if (c->h_pinned) hipHostFree(c->h_pinned);
    for (hipEvent_t e : c->pass_done) if (e) hipEventDestroy(e);
    if (c->ev_begin) hipEventDestroy(c->ev_begin);
    if (c->ev_end) hipEventDestroy(c->ev_end);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
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

int srt_commit_scene(srt_ctx* c, int world_kind)
{
    if (!c) return SRT_E_INVALID;
    if (c->async_pending) return fail(c, SRT_E_STATE, "an asynchronous render is in flight");
    auto t0 = std::chrono::steady_clock::now();
    c->committed = false;
    int rc = flatten_scene(c->host, world_kind, c->flat, c->error);
    if (rc != SRT_OK) return rc;
    SRT_HIP(c, hipSetDevice(c->device));
    SRT_HIP(c, hipStreamSynchronize(c->stream));
    free_scene_buffers(c);
    if ((rc = upload(c, c->flat.nodes, &c->d_nodes)) != SRT_OK) return rc;
    if ((rc = upload(c, c->flat.prims, &c->d_prims)) != SRT_OK) return rc;
    if ((rc = upload(c, c->flat.normals, &c->d_normals)) != SRT_OK) return rc;
    if ((rc = upload(c, c->flat.materials, &c->d_materials)) != SRT_OK) return rc;
    if ((rc = upload(c, c->flat.lights, &c->d_lights)) != SRT_OK) return rc;
    if ((rc = upload(c, c->flat.tiny, &c->d_tiny)) != SRT_OK) return rc;
    c->world_kind = world_kind;
    c->committed = true;
    c->stats.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    c->stats.bvh_nodes = c->flat.n_nodes;
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
    return launch(c, p, (uint32_t*)d_rgba_out, capacity_pixels, (hipStream_t)hip_stream, false);
}

static int start_async(srt_ctx* c, const srt_render_params* p, uint32_t* rgba_out, int passes)
{
    if (!c) return SRT_E_INVALID;
    if (!rgba_out) return fail(c, SRT_E_INVALID, "rgba_out is NULL");
    if (c->async_pending) return fail(c, SRT_E_STATE, "an asynchronous render is already in flight");
    int64_t n = output_pixels(p);
    if (n < 0) { RenderArgs ra; return make_render_args(p, c->cam, ra, c->error); }
    SRT_HIP(c, hipSetDevice(c->device));
    int rc = ensure_out(c, (size_t)n);
    if (rc != SRT_OK) return rc;
    if ((size_t)n > c->pinned_capacity) {
        if (c->h_pinned) hipHostFree(c->h_pinned);
        c->h_pinned = nullptr; c->pinned_capacity = 0;
        SRT_HIP(c, hipHostMalloc((void**)&c->h_pinned, (size_t)n * sizeof(uint32_t), hipHostMallocDefault));
        c->pinned_capacity = (size_t)n;
    }
    rc = launch(c, p, c->d_out, c->out_capacity, c->stream, false, passes);
    if (rc != SRT_OK) return rc;
    // ev_end marks the end of the kernels; the copy to pinned memory follows it on the same stream (a progressive
    // render has already queued that copy after its last pass)
    if (c->passes_total == 0) SRT_HIP(c, hipMemcpyAsync(c->h_pinned, c->d_out, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    c->async_dst = rgba_out;
    c->async_pixels = (size_t)n;
    c->async_pending = true;
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

int srt_poll(srt_ctx* c, int* done)
{
    if (!c) return SRT_E_INVALID;
    if (!done) return fail(c, SRT_E_INVALID, "done is NULL");
    if (!c->async_pending) { *done = 1; return SRT_OK; }
    hipError_t e = hipStreamQuery(c->stream);
    if (e == hipErrorNotReady) { *done = 0; return SRT_OK; }
    if (e != hipSuccess) return hip_fail(c, e, "hipStreamQuery");
    *done = 1;
    return finish_async(c);
}

int srt_wait(srt_ctx* c)
{
    if (!c) return SRT_E_INVALID;
    if (!c->async_pending) return SRT_OK;
    SRT_HIP(c, hipStreamSynchronize(c->stream));
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

int64_t srt_local_pixels(const srt_render_params* p) { return output_pixels(p); }

int srt_assemble_tiles(srt_ctx* c, const srt_render_params* p, const void* d_gathered, size_t stride_pixels, void* d_frame, void* hip_stream)
{
    if (!c) return SRT_E_INVALID;
    if (!d_gathered || !d_frame) return fail(c, SRT_E_INVALID, "device pointer is NULL");
    RenderArgs ra;
    Camera none{};
    srt_render_params q = *p;
    q.rank = 0;
    int rc = make_render_args(&q, none, ra, c->error);
    if (rc != SRT_OK) return rc;
    srt_render_params worst = q;
    worst.layout = SRT_LAYOUT_TILES;
    if ((int64_t)stride_pixels < output_pixels(&worst)) return fail(c, SRT_E_INVALID, "stride_pixels smaller than rank 0's tile buffer");
    SRT_HIP(c, hipSetDevice(c->device));
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
    SRT_HIP(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)width * (size_t)height * sizeof(uint32_t);
    uint32_t* d[2] = {nullptr, nullptr};
    uint32_t* d_progress = nullptr;
    const size_t progress_bytes = post_progress_bytes();
    uint32_t alarm = 0u;
    hipError_t e = hipMalloc((void**)&d[0], bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&d[1], bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&d_progress, progress_bytes);
    if (e == hipSuccess) e = hipMemsetAsync(d_progress, 0, progress_bytes, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d[0], array, bytes, hipMemcpyHostToDevice, c->stream);
    int from = 0;
    for (int32_t i = 0; i < passes && e == hipSuccess; i++, from ^= 1)
        e = launch_post_filter(filter, d[from], d[from ^ 1], width, height, weight, param, d_progress, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(array, d[from], bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&alarm, reinterpret_cast<char*>(d_progress) + progress_bytes - sizeof(uint32_t), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (d[0]) (void)hipFree(d[0]);
    if (d[1]) (void)hipFree(d[1]);
    if (d_progress) (void)hipFree(d_progress);
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
