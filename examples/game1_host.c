/*
 * game1_host.c -- what the reference's Game1 does around the trace loop, in plain C over the C ABI of libsrt.so:
 * build the scene of Game1.Initialize (Game1.cs:96-144), start the render (Game1.LoadContent, :157-175), poll it
 * like Game1.Update does every frame (:188-201), and save the frame (:239-244; here a binary PPM instead of a PNG).
 * It shows the call sequence a maintainer's P/Invoke layer makes and needs nothing but include/srt.h.
 *
 *   game1_host [width height samples max_depth seed out.ppm blur_passes devices]
 *
 * devices: how many GPUs share the frame, as Game1 shares it over numberOfAvailableThreads worker threads
 * (Game1.cs:89-92, 161-167): 0 = one GPU (device 0), N = devices 0..N-1, -1 = every visible GPU, and -N (N >= 2) = N
 * entries that all name device 0 (the several-device path on a box with a single GPU).  The frame is the same for
 * every choice.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/srt.h"

typedef struct { float x, y, z; } v3;

static v3 sub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static v3 scale(v3 a, float s) { v3 r = {a.x * s, a.y * s, a.z * s}; return r; }
static v3 cross(v3 a, v3 b) { v3 r = {a.y * b.z - b.y * a.z, -(a.x * b.z - b.x * a.z), a.x * b.y - b.x * a.y}; return r; }
static v3 normalize(v3 a)
{
    float f = 1.0f / (float)sqrt((double)((a.x * a.x) + (a.y * a.y) + (a.z * a.z)));
    return scale(a, f);
}

/* Camera(origin, lookAt, fov, ratio), RaytracingClasses.cs:34-50 */
static void camera(v3 position, v3 look_at, float fov, float ratio, float out[4][3])
{
    float theta = (float)((double)fov * 0.017453292519943295769236907684886);   /* MathHelper.ToRadians */
    float h = (float)tan((double)(theta / 2.0f));
    float viewport_height = 2.0f * h, viewport_width = viewport_height * ratio;
    v3 up = {0.0f, 1.0f, 0.0f};
    v3 w = normalize(sub(position, look_at));
    v3 u = normalize(cross(up, w));
    v3 v = cross(w, u);
    v3 horizontal = scale(u, viewport_width), vertical = scale(v, viewport_height);
    v3 llc = sub(sub(sub(position, scale(horizontal, 0.5f)), scale(vertical, 0.5f)), w);
    v3 all[4] = {position, llc, horizontal, vertical};
    for (int i = 0; i < 4; i++) { out[i][0] = all[i].x; out[i][1] = all[i].y; out[i][2] = all[i].z; }
}

#define CHECK(call) do { int rc_ = (call); if (rc_ != SRT_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, srt_last_error(ctx)); return 1; } } while (0)

int main(int argc, char** argv)
{
    int width = argc > 1 ? atoi(argv[1]) : 1280, height = argc > 2 ? atoi(argv[2]) : 720;   /* Game1.cs:41-42 */
    int samples = argc > 3 ? atoi(argv[3]) : 500, max_depth = argc > 4 ? atoi(argv[4]) : 4;  /* :43, :78 */
    unsigned seed = argc > 5 ? (unsigned)strtoul(argv[5], NULL, 10) : 1u;
    const char* path = argc > 6 ? argv[6] : "game1.ppm";
    int blur_passes = argc > 7 ? atoi(argv[7]) : 0;                                         /* Game1.cs:49: blurPasses = 0 */
    int devices = argc > 8 ? atoi(argv[8]) : 0;

    srt_ctx* ctx = NULL;
    srt_config cfg;
    memset(&cfg, 0, sizeof cfg);
    if (devices == -1) cfg.n_devices = SRT_ALL_DEVICES;
    else if (devices != 0) {
        int n = devices < 0 ? -devices : devices;
        if (n > SRT_MAX_DEVICES) { fprintf(stderr, "at most %d devices\n", SRT_MAX_DEVICES); return 1; }
        cfg.n_devices = n;
        for (int i = 0; i < n; i++) cfg.devices[i] = devices < 0 ? 0 : i;
    }
    cfg.gather = SRT_GATHER_RCCL;
    CHECK(srt_create(&cfg, &ctx));
    srt_device_info info;
    CHECK(srt_get_device_info(ctx, &info));
    fprintf(stderr, "%d device(s), tiles gathered by %s%s\n", info.n_devices, info.n_devices == 1 ? "nobody" : (info.gather == SRT_GATHER_RCCL ? "RCCL" : "peer copies"),
            info.rccl_ranks ? " (communicator made)" : "");

    /* Game1.cs:102-109 */
    srt_material mats[6] = {
        {{0.78f, 0.79f, 0.33f}, 0.0f, 0.995f, {0, 0, 0}}, {{1.0f, 0.65f, 0.62f}, 1.0f, 0.65f, {0, 0, 0}},
        {{0.5f, 0.425f, 0.15f}, 0.0f, 0.45f, {0, 0, 0}},  {{0, 0, 0}, 0.0f, 0.0f, {50.0f, 0.5f, 0.5f}},
        {{0, 0, 0}, 0.0f, 0.0f, {0.5f, 0.5f, 50.0f}},     {{1.0f, 0.5f, 0.5f}, 0.0f, 0.85f, {0, 0, 0}}};
    CHECK(srt_set_materials(ctx, mats, 6));
    /* Game1.cs:117-131 */
    float cx[5] = {2.05f, -4.5f, 0.0f, -20.0f, -15.0f}, cy[5] = {0.9f, 2.0f, 45.0f, 0.0f, 0.0f}, cz[5] = {-7.25f, -8.0f, -20.0f, 0.0f, 6.0f};
    float rad[5] = {3.5f, 2.3f, 42.5f, 2.0f, 6.0f};
    uint32_t sphere_mat[5] = {0, 1, 2, 3, 4};
    CHECK(srt_set_spheres(ctx, cx, cy, cz, rad, sphere_mat, 5));
    /* Game1.cs:140-141 */
    srt_point_light lights[2] = {{{6.0f, 2.0f, 3.0f}, 5.0f, {0, 0, 0}}, {{0.0f, 0.0f, -12.0f}, 2.5f, {0, 0, 0}}};
    CHECK(srt_set_lights(ctx, lights, 2));
    /* Game1.cs:86, 98 */
    float cam[4][3];
    v3 pos = {0.0f, 0.0f, 0.0f}, look = {-0.03f, 0.875f, -7.6f};
    camera(pos, look, 63.0f, (float)width / (float)height, cam);
    CHECK(srt_set_camera(ctx, cam[0], cam[1], cam[2], cam[3]));
    /* world.Init(); nodeWorld = new BVHNode(world), Game1.cs:136-138 */
    CHECK(srt_commit_scene(ctx, SRT_WORLD_BVHNODE));

    srt_render_params p;
    memset(&p, 0, sizeof p);
    p.width = width; p.height = height; p.samples = samples; p.max_depth = max_depth;
    p.color_clamp = 1.0f; p.seed = seed; p.mode = SRT_MODE_COLOR; p.layout = SRT_LAYOUT_FRAME;
    int64_t n = srt_local_pixels(&p);
    if (n < 0) { fprintf(stderr, "bad render parameters (%lld)\n", (long long)n); return 1; }
    uint32_t* frame = (uint32_t*)malloc((size_t)n * sizeof(uint32_t));
    if (!frame) return 1;

    /* LoadContent starts the render, Update polls it while the window stays responsive */
    CHECK(srt_render_async(ctx, &p, frame));
    int done = 0;
    unsigned long polls = 0;
    while (!done) { CHECK(srt_poll(ctx, &done)); polls++; }
    CHECK(srt_wait(ctx));
    srt_stats st;
    CHECK(srt_get_stats(ctx, &st));
    fprintf(stderr, "%dx%d, %d spp, depth %d: %.3f ms on the device, %llu rays, %.1f Mrays/s, %lu polls\n", width, height, samples,
            max_depth, st.kernel_ms, (unsigned long long)st.rays_path, (double)st.rays_path / st.kernel_ms / 1e3, polls);

    /* Game1.Update, once the render is done (Game1.cs:226-232): blurPasses times Sorting.Despeckle(buffer1Data, width,
       height, blurWeight = 1, speckleThreshold = 200) */
    if (blur_passes > 0) {
        CHECK(srt_despeckle(ctx, frame, width, height, 1.0f, 200.0f, blur_passes));
        fprintf(stderr, "despeckled, %d passes\n", blur_passes);
    }

    /* rows in buffer order, as the reference saves its Color[] (buffer row 0 is the bottom of the camera's view and
       the top of the saved picture: the scene is set up for that, see testrender.jpg) */
    FILE* f = fopen(path, "wb");
    if (!f) { perror(path); return 1; }
    fprintf(f, "P6\n%d %d\n255\n", width, height);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            uint32_t c = frame[(size_t)x + (size_t)y * (size_t)width];
            unsigned char rgb[3] = {(unsigned char)(c & 255u), (unsigned char)((c >> 8) & 255u), (unsigned char)((c >> 16) & 255u)};
            fwrite(rgb, 1, 3, f);
        }
    fclose(f);
    free(frame);
    srt_destroy(ctx);
    return 0;
}
