// emu_api.cpp -- TEST HARNESS, not part of the product.
//
// Runs the device code of simple_raytracer_amd/csrc/srt_core.h (the per-lane state machine, the
// traversal and the shading) and the host-side scene flattening of srt_scene.cpp on the CPU,
// one lane at a time, behind the same C ABI as libsrt.so.  It lets the CPU test-suite compare
// the kernel logic with the oracle in a container that has no GPU.  libsrt.so contains none
// of this; its renders always run on the device.
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/srt.h"
#include <algorithm>

#include "../../simple_raytracer_amd/csrc/srt_multi.h"
#include "../../simple_raytracer_amd/csrc/srt_path.h"
#include "../../simple_raytracer_amd/csrc/srt_post.h"
#include "../../simple_raytracer_amd/csrc/srt_scene.h"

using namespace srt;

struct srt_ctx {
    HostScene host;
    FlatScene flat;
    bool committed = false;
    int world_kind = 0;
    Camera cam{};
    bool has_camera = false;
    srt_stats stats{};
    int n_devices = 1;          // virtual devices: the frame is partitioned as libsrt.so partitions it over GPUs
    std::string error;
};

namespace {

// closest_hit() through the resumable traversal in the smallest possible quanta (one inner node per round), so
// that the CPU suite checks that splitting a traversal changes nothing.
template <bool EXACT>
Hit query(const SceneView& sc, f3 o, f3 d, float t_min, uint32_t* stack, Counters& cnt)
{
    Trav tr;
    tr.d = d;
    trav_begin<EXACT, true, 1>(sc, o, t_min, tr, stack, cnt);
    int budget = 1;
    while (tr.cur != TRAV_NONE) {
        if (!EXACT && tr.pad != 0.0f) trav_run<EXACT, true, 1, !EXACT>(sc, o, t_min, tr, stack, cnt, 1, budget);
        else trav_run<EXACT, true, 1>(sc, o, t_min, tr, stack, cnt, 1, budget);
        budget = budget % 3 + 1;
    }
    return tr.best;
}

struct Totals { uint64_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0}; };   // rays, hit nodes, tests; 7: shadow queries left out

// Items [i0, i1) of one pass, one lane at a time: the same per-lane functions the kernel calls, without the
// wavefront scheduling around them (which does not influence results).
template <bool EXACT>
void run_items(const SceneView& sc, const RenderArgs& ra, uint32_t i0, uint32_t i1, f3* samples, Totals* tot)
{
    std::vector<uint32_t> stack(STACK_DEPTH);
    Counters cnt{0, 0, 0, 0};
    uint64_t unasked = 0;
    float unit[UNIT_TABLE];
    fill_unit_table(unit);
    for (uint32_t item = i0; item < i1; item++) {
        Lane ln;
        Frame frame_mem[FRAME_SLOTS];
        PrivateFrames frames;
        frames.a = frame_mem;
        ln.rays_path = ln.rays_shadow = ln.hit_nodes = 0u;
        if (!item_start(ln, ra, item, unit)) continue;
        int phase = LANE_TRACE;
        f3 ret = mk3(0, 0, 0);
        while (phase != LANE_IDLE) {
            if (phase == LANE_RETURN) {
                phase = lane_return(ln, ret, sc, ra, frames);
            } else if (phase == LANE_TRACE) {
                Hit h = query<EXACT>(sc, ln.qo, ln.qd, ln.qtmin, stack.data(), cnt);
                phase = lane_on_closest(ln, h, ret, sc, ra, frames, unit);
            } else {
                uint32_t aux[4];
                rng_call(ln.key, CALL_AUX, aux);
                for (uint32_t li = 0; li < sc.n_lights; li++) {
                    if (!lane_shadow_matters(ln, sc, li)) { lane_skip_light(ln); unasked++; continue; }
                    f3 o, d;
                    lane_shadow_query(ln, sc, li, aux, unit, o, d);
                    Hit h = query<EXACT>(sc, o, d, 0.001f, stack.data(), cnt);
                    lane_on_shadow(ln, h, o, d, sc, li);
                }
                phase = lane_shade(ln, ln.former, ret, sc, ra, frames, aux, unit);
            }
        }
        samples[ln.slot] = ret;
        tot->v[0] += ln.rays_path; tot->v[1] += ln.rays_shadow; tot->v[2] += ln.hit_nodes;
    }
    tot->v[3] += cnt.box_tests; tot->v[4] += cnt.sphere_tests; tot->v[5] += cnt.tri_tests; tot->v[6] += cnt.rect_tests;
    tot->v[7] += unasked;
}

}  // namespace

namespace {

// The post-process filters: the kernel's per-pixel function (srt_post.h) over the picture in the reference's order,
// row-major and in place, so that a pixel sees the new values of its left neighbour and of the row above.
template <int FILTER>
static void post_pass(uint32_t* a, int32_t w, int32_t h, float weight, float param)
{
    weight = weight < 0.0f ? 0.0f : (weight > 1.0f ? 1.0f : weight);
    for (int32_t y = 1; y < h - 1; y++)
        for (int32_t x = 1; x < w - 1; x++) {
            Window win;
            win.left = a[x - 1 + y * w]; win.right = a[x + 1 + y * w];
            win.up = a[x + (y - 1) * w]; win.down = a[x + (y + 1) * w];
            win.left_up = a[x - 1 + (y - 1) * w]; win.right_up = a[x + 1 + (y - 1) * w];
            win.left_down = a[x - 1 + (y + 1) * w]; win.right_down = a[x + 1 + (y + 1) * w];
            a[x + y * w] = filtered<FILTER>(a[x + y * w], win, weight, param);
        }
}
static int post_check(srt_ctx* c, uint32_t* a, int32_t w, int32_t h, int32_t passes)
{
    if (!c || !a || w < 1 || h < 1 || passes < 1) return SRT_E_INVALID;
    return SRT_OK;
}
}  // namespace

extern "C" {

int srt_version(void) { return SRT_VERSION; }
const char* srt_last_error(const srt_ctx* c) { return c ? c->error.c_str() : ""; }
int srt_create(const srt_config* cfg, srt_ctx** out)
{
    *out = nullptr;
    int n = cfg ? cfg->n_devices : 0;
    if (n == 0 || n == SRT_ALL_DEVICES) n = 1;
    if (n < 1 || n > SRT_MAX_DEVICES) return SRT_E_INVALID;
    *out = new srt_ctx();
    (*out)->n_devices = n;
    return SRT_OK;
}
int srt_get_device_info(const srt_ctx* c, srt_device_info* out)
{
    if (!c || !out) return SRT_E_INVALID;
    std::memset(out, 0, sizeof(*out));
    out->n_devices = c->n_devices;
    out->gather = SRT_GATHER_PEER_COPY;
    return SRT_OK;
}
void srt_destroy(srt_ctx* c) { delete c; }

int srt_set_camera(srt_ctx* c, const float p[3], const float l[3], const float h[3], const float v[3])
{
    c->cam.position = mk3(p[0], p[1], p[2]); c->cam.lower_left = mk3(l[0], l[1], l[2]);
    c->cam.horizontal = mk3(h[0], h[1], h[2]); c->cam.vertical = mk3(v[0], v[1], v[2]);
    c->has_camera = true;
    return SRT_OK;
}
int srt_set_materials(srt_ctx* c, const srt_material* m, uint32_t n) { c->committed = false; return host_set_materials(c->host, m, n, c->error); }
int srt_set_spheres(srt_ctx* c, const float* cx, const float* cy, const float* cz, const float* r, const uint32_t* mat, uint32_t n)
{
    c->committed = false;
    return host_set_spheres(c->host, cx, cy, cz, r, mat, n, c->error);
}
int srt_set_rects(srt_ctx* c, const float* x1, const float* x2, const float* y1, const float* y2, const float* k, const uint32_t* mat, uint32_t n)
{
    c->committed = false;
    return host_set_rects(c->host, x1, x2, y1, y2, k, mat, n, c->error);
}
int srt_set_triangles(srt_ctx* c, const float* pos9, const float* nrm9, const uint32_t* mat, const uint8_t* flags, uint32_t n)
{
    c->committed = false;
    return host_set_triangles(c->host, pos9, nrm9, mat, flags, n, c->error);
}
int srt_set_lights(srt_ctx* c, const srt_point_light* l, uint32_t n) { c->committed = false; return host_set_lights(c->host, l, n, c->error); }
int srt_set_surface_order(srt_ctx* c, const uint8_t* type, const uint32_t* index, uint32_t n) { c->committed = false; return host_set_surface_order(c->host, type, index, n, c->error); }

int srt_commit_scene(srt_ctx* c, int world_kind)
{
    c->committed = false;
    int rc = flatten_scene(c->host, world_kind, c->flat, c->error);
    if (rc != SRT_OK) return rc;
    c->world_kind = world_kind;
    c->committed = true;
    c->stats.bvh_nodes = c->flat.n_tree_nodes; c->stats.bvh_depth = c->flat.bvh_depth; c->stats.n_prims = c->flat.n_prims;
    return SRT_OK;
}

static int render_passes(srt_ctx* c, const srt_render_params* p, uint32_t* out, int mix_buffers)
{
    if (!c->committed || !c->has_camera) { c->error = "scene or camera missing"; return SRT_E_STATE; }
    RenderArgs ra;
    int rc = make_render_args(p, c->cam, ra, c->error);
    if (rc != SRT_OK) return rc;
    SceneView sc = c->flat.view(c->flat.nodes.data(), c->flat.prims.data(), c->flat.normals.data(), c->flat.materials.data(),
                                c->flat.lights.data(), c->flat.tiny.data());
    unsigned n = std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    std::vector<Totals> tot(n);
    const bool exact = c->world_kind == SRT_WORLD_BVHNODE;
    // Passes of at most 3 samples, so that the running-sum path of the resolve step is exercised too.
    // mix_buffers > 0: the mix-buffer technique, one sample (index b) per buffer, each buffer finished on its own and
    // mixed into `out` (as srt_render_mix_buffers does with the trace, resolve and mix kernels)
    const uint32_t per_pass = mix_buffers > 0 ? 1 : 3;
    const uint32_t total = mix_buffers > 0 ? (uint32_t)mix_buffers : (uint32_t)ra.samples;
    std::vector<f3> samples((size_t)per_pass * ra.n_work), running(ra.n_work);
    std::vector<uint32_t> buffer(mix_buffers > 0 ? (size_t)ra.width * (size_t)ra.height : 0);
    uint32_t* frame = out;
    if (mix_buffers > 0) out = buffer.data();
    for (uint32_t s0 = 0; s0 < total; s0 += per_pass) {
        ra.sample_first = s0;
        ra.sample_count = mix_buffers > 0 ? 1u : std::min<uint32_t>(per_pass, (uint32_t)ra.samples - s0);
        set_item_order(ra, 2u);   // as the GPU launcher does for scenes without a traversal stack
        ra.n_items = ra.n_work * ra.sample_count;
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n; t++) {
            pool.emplace_back([&, t]() {
                for (uint64_t i0 = (uint64_t)t * 64; i0 < ra.n_items; i0 += (uint64_t)n * 64) {
                    uint32_t i1 = (uint32_t)std::min<uint64_t>(i0 + 64, ra.n_items);
                    if (exact) run_items<true>(sc, ra, (uint32_t)i0, i1, samples.data(), &tot[t]);
                    else run_items<false>(sc, ra, (uint32_t)i0, i1, samples.data(), &tot[t]);
                }
            });
        }
        for (auto& th : pool) th.join();
        const bool first = mix_buffers > 0 || s0 == 0, last = mix_buffers > 0 || s0 + ra.sample_count >= (uint32_t)ra.samples;
        for (uint32_t wp = 0; wp < ra.n_work; wp++) {   // what resolve_kernel does per thread
            int32_t x, y;
            uint32_t oi;
            if (!work_to_pixel(ra, wp, x, y, oi)) {
                if (ra.layout != 0 && last) out[oi] = 0u;
                continue;
            }
            f3 acc = first ? mk3(0.0f, 0.0f, 0.0f) : running[wp];
            acc = accumulate_samples(acc, samples.data(), ra.n_work, wp, ra.sample_count, 1.0f / (float)ra.samples);
            if (last) out[oi] = finish_color(acc);
            else running[wp] = acc;
        }
        if (mix_buffers > 0)
            for (size_t i = 0; i < buffer.size(); i++) frame[i] = mix_colors(s0 == 0 ? 0u : frame[i], buffer[i]);   // mix_kernel
    }
    uint64_t s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (auto& t : tot) for (int i = 0; i < 8; i++) s[i] += t.v[i];
    c->stats.rays_path = s[0]; c->stats.rays_shadow = s[1]; c->stats.hit_nodes = s[2];
    c->stats.box_tests = s[3]; c->stats.sphere_tests = s[4]; c->stats.tri_tests = s[5]; c->stats.rect_tests = s[6];
    c->stats.shadow_unasked = s[7];
    return SRT_OK;
}

// The virtual devices of a context as the backend of render_multi(): the orchestration libsrt.so runs over its GPUs
// (srt_multi.h), with host memory in place of device memory and copies in place of the gather.
struct EmuDevices {
    srt_ctx* c;
    uint32_t* frame;
    std::vector<uint32_t> gathered;
    std::vector<std::vector<uint32_t>> tiles;
    srt_stats total{};
    int prepare(const MultiPlan& plan)
    {
        gathered.assign((size_t)plan.stride * (size_t)plan.n, 0xDEADBEEFu);
        tiles.resize((size_t)plan.n);
        for (int k = 1; k < plan.n; k++) tiles[(size_t)k].assign((size_t)plan.pixels[k], 0xDEADBEEFu);
        return SRT_OK;
    }
    int trace_tiles(int k, const MultiPlan& plan)
    {
        if (plan.pixels[k] == 0) return SRT_OK;
        uint32_t* out = k == 0 ? gathered.data() : tiles[(size_t)k].data();
        int rc = render_passes(c, &plan.part[k], out, 0);
        total.rays_path += c->stats.rays_path; total.rays_shadow += c->stats.rays_shadow; total.hit_nodes += c->stats.hit_nodes;
        total.box_tests += c->stats.box_tests; total.sphere_tests += c->stats.sphere_tests; total.tri_tests += c->stats.tri_tests;
        total.rect_tests += c->stats.rect_tests; total.shadow_unasked += c->stats.shadow_unasked;
        return rc;
    }
    int gather(const MultiPlan& plan)
    {
        for (int k = 1; k < plan.n; k++)
            std::memcpy(gathered.data() + (size_t)k * (size_t)plan.stride, tiles[(size_t)k].data(), (size_t)plan.pixels[k] * sizeof(uint32_t));
        return SRT_OK;
    }
    int assemble(const MultiPlan& plan)
    {
        for (uint32_t i = 0; i < (uint32_t)plan.frame_pixels; i++) frame[i] = gathered[gathered_index(plan.assemble, i, (size_t)plan.stride)];   // assemble_kernel
        return SRT_OK;
    }
};

int srt_render(srt_ctx* c, const srt_render_params* p, uint32_t* out)
{
    if (c->n_devices == 1) return render_passes(c, p, out, 0);
    MultiPlan plan;
    int rc = plan_multi(p, c->n_devices, plan, c->error);
    if (rc != SRT_OK) return rc;
    EmuDevices backend{c, out, {}, {}, {}};
    rc = render_multi(backend, plan);
    const srt_stats& t = backend.total;
    c->stats.rays_path = t.rays_path; c->stats.rays_shadow = t.rays_shadow; c->stats.hit_nodes = t.hit_nodes;
    c->stats.box_tests = t.box_tests; c->stats.sphere_tests = t.sphere_tests; c->stats.tri_tests = t.tri_tests; c->stats.rect_tests = t.rect_tests;
    return rc;
}
int srt_render_mix_buffers(srt_ctx* c, const srt_render_params* p, int32_t n_buffers, uint32_t* out)
{
    if (n_buffers < 1 || p->layout != 0 || p->world_size > 1) return SRT_E_INVALID;
    return render_passes(c, p, out, n_buffers);
}
int srt_render_counted(srt_ctx* c, const srt_render_params* p, uint32_t* out) { return srt_render(c, p, out); }

// Test hook (not part of include/srt.h): the item order of one pass.  For every work item of a pass over all of
// p->samples with the last `scatter_samples` in scattered order: slots[item] = its sample-buffer slot, pixels[item] =
// x + y * width, or 0xFFFFFFFF for a padding pixel of an edge tile.  Returns the number of items.
int srt_blur_array(srt_ctx* c, uint32_t* a, int32_t w, int32_t h, float weight, int32_t passes)
{
    if (int rc = post_check(c, a, w, h, passes)) return rc;
    for (int32_t i = 0; i < passes; i++) post_pass<SRT_POST_BLUR>(a, w, h, weight, 0.0f);
    return SRT_OK;
}
int srt_contrast_blur_array(srt_ctx* c, uint32_t* a, int32_t w, int32_t h, float weight, float contrast, int32_t passes)
{
    if (int rc = post_check(c, a, w, h, passes)) return rc;
    for (int32_t i = 0; i < passes; i++) post_pass<SRT_POST_CONTRAST_BLUR>(a, w, h, weight, contrast);
    return SRT_OK;
}
int srt_despeckle(srt_ctx* c, uint32_t* a, int32_t w, int32_t h, float weight, float threshold, int32_t passes)
{
    if (int rc = post_check(c, a, w, h, passes)) return rc;
    for (int32_t i = 0; i < passes; i++) post_pass<SRT_POST_DESPECKLE>(a, w, h, weight, threshold);
    return SRT_OK;
}

int64_t srt_emu_item_order(srt_ctx* c, const srt_render_params* p, uint32_t scatter_samples, uint32_t* slots, uint32_t* pixels, uint64_t capacity)
{
    RenderArgs ra;
    if (make_render_args(p, c->cam, ra, c->error) != SRT_OK) return -1;
    ra.sample_first = 0u;
    ra.sample_count = (uint32_t)ra.samples;
    set_item_order(ra, scatter_samples);
    ra.n_items = ra.n_work * ra.sample_count;
    if ((uint64_t)ra.n_items > capacity) return -2;
    float unit[UNIT_TABLE];
    fill_unit_table(unit);
    for (uint32_t item = 0; item < ra.n_items; item++) {
        Lane ln;
        const bool real = item_start(ln, ra, item, unit);
        slots[item] = real ? ln.slot : 0xFFFFFFFFu;
        pixels[item] = real ? ln.key.pixel : 0xFFFFFFFFu;
    }
    return (int64_t)ra.n_items;
}
// Test hook: the surfaces of a committed SRT_WORLD_BVHNODE world in the order the tree's leaves hold them.
int64_t srt_emu_leaf_order(srt_ctx* c, uint32_t* out, uint64_t capacity)
{
    if (!c->committed || c->flat.leaf_order.size() > capacity) return -1;
    for (size_t i = 0; i < c->flat.leaf_order.size(); i++) out[i] = c->flat.leaf_order[i];
    return (int64_t)c->flat.leaf_order.size();
}
// FNV-1a over the flattened arrays of the committed scene (nodes, primitives, normals): two builds of one scene
// give the same digest exactly when they laid out the same tree.
uint64_t srt_emu_scene_digest(srt_ctx* c)
{
    if (!c->committed) return 0;
    uint64_t h = 1469598103934665603ull;
    auto eat = [&](const std::vector<q4>& v) {
        const unsigned char* b = reinterpret_cast<const unsigned char*>(v.data());
        for (size_t i = 0; i < v.size() * sizeof(q4); i++) { h ^= b[i]; h *= 1099511628211ull; }
    };
    eat(c->flat.nodes); eat(c->flat.prims); eat(c->flat.normals);
    return h;
}
int srt_get_stats(srt_ctx* c, srt_stats* out) { *out = c->stats; return SRT_OK; }
int64_t srt_local_pixels(const srt_render_params* p) { return output_pixels(p); }

}  // extern "C"
