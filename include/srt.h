/*
 * srt.h -- C ABI of libsrt.so, the MI355X (gfx950) trace loop for helpvisa/simple_raytracer.
 *
 * The reference (C#, MonoGame) has no plugin or FFI seam.  This ABI is the seam a maintainer
 * adds: Game1.LoadContent stops spawning renderBlock threads (Game1.cs:161-175, 376-380) and
 * instead hands the scene it built in Game1.Initialize (Game1.cs:96-144) to this library, which
 * fills the same Color[] buffer (Game1.cs:157, PixelManagement.cs:8-23).  Every entry point
 * names the reference code it stands in for.
 *
 * Conventions
 *  - plain C, no C++ types or exceptions cross the boundary; every function returns 0
 *    (SRT_OK) or a negative SRT_E_* code, and srt_last_error() describes the last failure;
 *  - pointers passed in are caller-owned and only read during the call (scene arrays are
 *    copied); output buffers are caller-owned;
 *  - a context owns one HIP device or a list of them (srt_config) and is not thread-safe: the
 *    caller serialises calls, as the single UI thread of Game1 does;
 *  - there is no CPU fallback: without a usable gfx950 device srt_create fails.
 */
#ifndef SRT_H
#define SRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRT_VERSION 300

enum {
    SRT_OK = 0,
    SRT_E_INVALID = -1,      /* bad argument */
    SRT_E_NO_DEVICE = -2,    /* no usable HIP device / not gfx950 */
    SRT_E_HIP = -3,          /* a HIP runtime call failed */
    SRT_E_STATE = -4,        /* call sequence error (e.g. render before commit) */
    SRT_E_REFERENCE = -5,    /* the reference program would throw for this configuration */
    SRT_E_LIMIT = -6         /* a documented size limit is exceeded */
};

/* Which Surface the trace loop intersects, named after the reference object it reproduces. */
enum {
    /* GetRayColor(ray, nodeWorld, ...) as Game1.cs:293 calls it: the BVHNode tree built by the
     * reference's own builder (RaytracingClasses.cs:346-426) and walked with BVHNode.hit /
     * AABB.hit semantics (:428-442, :773-789).  A world holding a RectAxis is rejected with
     * SRT_E_REFERENCE (RectAxis has no boundingBox override, :503-544; the reference builder
     * dereferences its null bounds). */
    SRT_WORLD_BVHNODE = 0,
    /* GetRayColor(ray, world, ...): SurfaceList.hit semantics (RaytracingClasses.cs:298-316), the
     * closest hit over every surface in insertion order, later surface winning an exact tie.
     * The library answers this query through its own tree (binned SAH, collapsed to 4-wide quantised nodes), which
     * returns the same hit as the linear scan: boxes only cull, and they are padded by more than the float surface
     * tests can be wrong by -- Sphere.hit accepts rays that pass a sphere at a distance that grows with the distance
     * of the ray's origin (the rounding error of its discriminant), so sphere boxes are widened for every origin in
     * the ball around the scene, and a query that starts farther away widens the boxes it tests in proportion to its
     * distance; surfaces with coordinates beyond 2^40, and the queries whose tests can return NaN, fall back to the
     * scan itself. */
    SRT_WORLD_SURFACELIST = 1
};

/* Per-ray colour function (RaytracingOperations.cs), selected in the reference by editing Game1.cs:290-293. */
enum {
    SRT_MODE_COLOR = 0,      /* RayOperations.GetRayColor        RaytracingOperations.cs:75-195 */
    SRT_MODE_NORMAL = 1,     /* RayOperations.GetRayNormalColor  RaytracingOperations.cs:11-21 */
    SRT_MODE_DEPTH = 2,      /* RayOperations.GetRayDepthColor   RaytracingOperations.cs:24-35 */
    SRT_MODE_LIGHTS = 3,     /* RayOperations.GetLights          RaytracingOperations.cs:197-205 */
    SRT_MODE_DIFFUSE = 4,    /* RayOperations.GetRayColorDiffuse    RaytracingOperations.cs:38-54, depth = max_depth */
    SRT_MODE_REFLECTION = 5  /* RayOperations.GetRayColorReflection RaytracingOperations.cs:57-73, depth = max_depth */
};

/* Output layout of a render. */
enum {
    SRT_LAYOUT_FRAME = 0,    /* row-major W*H, pixel (x,y) at x + y*W (PixelManagement.cs:22) */
    SRT_LAYOUT_TILES = 1     /* this rank's tiles only, tile-major; see srt_render_params */
};

#define SRT_MAX_DEPTH 16      /* largest max_depth accepted */
#define SRT_MAX_LIGHTS 4096

/* Material after its constructor ran (clamped albedo / metalness / smoothness), RaytracingClasses.cs:221-262. 32 bytes. */
typedef struct { float albedo[3]; float metalness; float smoothness; float emission[3]; } srt_material;
/* PointLight after its constructor ran (brightness possibly rescaled), RaytracingClasses.cs:75-94,190-192. 28 bytes. */
typedef struct { float position[3]; float brightness; float color[3]; } srt_point_light;

/* How the devices of a context hand their screen tiles to the first device of the list. */
enum {
    SRT_GATHER_RCCL = 0,      /* grouped ncclSend / ncclRecv over a communicator made by ncclCommInitAll (the default) */
    SRT_GATHER_PEER_COPY = 1  /* hipMemcpyPeerAsync from every device into the first one's buffer */
};
#define SRT_MAX_DEVICES 8     /* one node */
#define SRT_ALL_DEVICES (-1)  /* srt_config.n_devices: every visible device */

/* The devices of a context.  The reference fans one render out over Environment.ProcessorCount threads inside one
 * process (Game1.cs:89-92, 161-167); a context fans it out over the listed GPUs the same way, inside the library:
 * every device holds the whole scene, traces the 32x32 screen tiles t with t % n_devices == its position in the list
 * (tile numbering: srt_render_params),
 * and hands them to the first device, which assembles the frame.
 *   n_devices == 0                    one device: `device` (callers that only know the first two fields keep working)
 *   n_devices == SRT_ALL_DEVICES      every visible device, in ordinal order
 *   n_devices 1..SRT_MAX_DEVICES      devices[0..n_devices).  An ordinal may be listed more than once: the entries then
 *                                     share that GPU (a way to run the multi-device path on a box with one GPU), and
 *                                     the tiles travel by device-to-device copies, since RCCL refuses duplicate devices. */
typedef struct {
    int32_t device;
    int32_t n_devices;
    int32_t devices[SRT_MAX_DEVICES];
    int32_t gather;           /* SRT_GATHER_* */
    int32_t reserved[5];      /* zero */
} srt_config;

/* What srt_create made of the configuration. */
typedef struct {
    int32_t n_devices;
    int32_t devices[SRT_MAX_DEVICES];   /* HIP ordinals, in list order; the first one assembles the frame */
    int32_t gather;           /* SRT_GATHER_* in use (peer copies when an ordinal is listed twice or RCCL is not needed) */
    int32_t rccl_ranks;       /* ncclCommCount of the communicator, 0 when none was made (one device, peer copies) */
    int32_t reserved[5];
} srt_device_info;

/* Render settings: the hard-coded fields of Game1 (Game1.cs:41-43,78-79) plus the RNG seed
 * (the reference seeds System.Random from entropy, Game1.cs:44,378) and the multi-GPU
 * partition (the reference's row bands per thread, Game1.cs:376-380, become screen tiles
 * per rank). */
typedef struct {
    int32_t  width, height;  /* full image, globalWidth / globalHeight */
    int32_t  samples;        /* Game1.samples */
    int32_t  max_depth;      /* Game1.maxDepth, 1..SRT_MAX_DEPTH */
    float    color_clamp;    /* Game1.colorClamp, >= 0 */
    uint32_t seed;
    int32_t  mode;           /* SRT_MODE_* */
    int32_t  layout;         /* SRT_LAYOUT_* */
    int32_t  tile_w, tile_h; /* screen tile, multiples of 8; 0 = 32.  Only used with SRT_LAYOUT_TILES. */
    int32_t  rank, world_size; /* tile t belongs to rank t % world_size.  Tiles are numbered row by row over the tile
                                  grid (tiles_x = ceil(width / tile_w) columns), starting ty columns in in row ty: tile t is
                                  in row ty = t / tiles_x, column (t % tiles_x + ty) % tiles_x, so that the ranks lie on
                                  diagonals of the grid whatever tiles_x is.  A rank's j-th tile (j = t / world_size)
                                  occupies pixels [j*tile_w*tile_h, (j+1)*tile_w*tile_h) of its buffer, row-major inside
                                  the tile, edge tiles padded.  world_size 0 or 1 = everything. */
} srt_render_params;

typedef struct {
    uint64_t rays_path;      /* GetRayColor calls with depth >= 1: primary + bounce rays */
    uint64_t rays_shadow;    /* shadow rays, RaytracingClasses.cs:174: one per light of every hit node, including those the
                                library answers without a traversal (shadow_unasked) */
    uint64_t hit_nodes;      /* GetRayColor calls that hit a surface */
    uint64_t box_tests;      /* BVH nodes whose box was tested   (only filled by srt_render_counted) */
    uint64_t sphere_tests, tri_tests, rect_tests;   /* (only filled by srt_render_counted) */
    double   kernel_ms;      /* device time of the last render (its trace and resolve kernels), from HIP events on its stream */
    double   build_ms;       /* host time of the last srt_commit_scene (BVH build + upload) */
    uint32_t bvh_nodes;      /* nodes of the committed tree: BVHNode objects of the reference tree (32 bytes each) or the library's 4-wide nodes (64) */
    uint32_t bvh_depth;
    uint32_t n_prims;
    uint32_t walk_tune;      /* scenes whose tree is walked with a stack: traversal quantum in use, 0 long, 1 short, 2 not measured yet
                                (the first launches after a commit measure both and keep the faster) */
    /* Wavefront occupancy of the trace kernel (only filled by srt_render_counted): how often a wavefront ran its
       closest-hit section / its light section, and how many of the 64 lanes took part, summed. */
    uint64_t trace_sections, trace_lanes, light_sections, light_lanes;
    /* Shader-clock cycles summed over wavefronts (only filled by srt_render_counted), by part of the kernel:
       0 dealing work items, 1 handing values up the tree, 2 closest-hit query, 3 bookkeeping after it,
       4 random words of a lit node, 5 shadow-ray set-up, 6 shadow query, 7 light evaluation,
       8 child directions, 9 fresnel, 10 starting a child / finishing a leaf;
       11-14 (stackless scenes) how the launch ended, in 100 MHz ticks a wavefront kept running after the work ran out:
       the longest, the sum, the wavefronts above 250 us, those above 1 ms;
       15 is a count, not cycles: subtrees that lanes handed to idle lanes of their wavefront (mesh scenes). */
    uint64_t section_cycles[16];
    /* Lane-cycles (only filled by srt_render_counted; stackless scenes): slots 0-10 as above but summed over the 64 lanes
       of every wavefront, counting a lane only while it worked in the part; 11-14 the lane-cycles spent waiting for the
       other lanes of the wavefront to finish handing values up / taking new items / their closest-hit part / their
       light part.  Slots 0-10 over the sum of 0-14 is the share of lane time that did work. */
    uint64_t section_lane_cycles[16];
    /* Shadow rays answered without a traversal (only filled by srt_render_counted; part of rays_shadow).  Scenes whose
       tree is walked with a stack leave out the shadow ray of a light behind the surface when every factor of
       PointLight.calculateLight is finite: with NdotL = 0 (RaytracingClasses.cs:113) the diffuse and the specular term
       (:160-161, :185) are zero whatever the ray meets, so the result is the reference's bit for bit. */
    uint64_t shadow_unasked;
    /* Contexts with several devices, last frame (0 otherwise), from HIP events on the first device's stream:
       gather_ms    from the end of the first device's own tiles to the arrival of every other device's tiles there --
                    the transfers and the wait for the slowest device;
       assemble_ms  the kernel that puts the gathered tiles into the row-major frame. */
    double   gather_ms, assemble_ms;
    /* Scenes whose tree is walked with a stack, once their renders have shown more than a few rays per sample: the ray
       trees are traced level by level through a pool of node records in device memory (one kernel per part of
       GetRayColor and level) instead of one tree per lane.  Work items (samples) traced that way in the last render, the
       node records they took (reservation holes included), and the chunks of items whose trees outgrew the pool and
       were traced by the one-kernel tracer after all (same colours, same counts).  The items include the padding pixels
       of edge tiles.
       wavefront_twins: GetRayColor calls of the last render whose world.hit was not computed again because the call
       before it asked the same question: the metalness pass and the reflectivity pass of a surface with 0 < metalness
       < 1 are both given CustomRay(record.point, reflectedDirection) (RaytracingOperations.cs:160,163), and world.hit
       is a function of the ray.  They are part of rays_path like every other call; srt_render_counted reports them
       too but walks the tree for them all the same (its test counts are those of a tracer that asks every question). */
    uint64_t wavefront_items, wavefront_nodes, wavefront_given_up, wavefront_twins;
} srt_stats;

typedef struct srt_ctx srt_ctx;

int  srt_version(void);

/* Replaces nothing in the reference: owns the device-side copies of what Game1 keeps on the heap, one replica per
 * device of the configuration (NULL = device 0). */
int  srt_create(const srt_config* cfg, srt_ctx** out);
void srt_destroy(srt_ctx* ctx);
/* The devices a context runs on (numberOfAvailableThreads of Game1.cs:89-92, for GPUs). */
int  srt_get_device_info(const srt_ctx* ctx, srt_device_info* out);
/* Thread-local text of the last error of `ctx` (or of srt_create when ctx is NULL). */
const char* srt_last_error(const srt_ctx* ctx);

/* Camera.position / lowerLeftCorner / horizontal / vertical as its constructor computed them
 * (RaytracingClasses.cs:34-50, 57-63); read by the ray generation of Game1.cs:283-287. */
int  srt_set_camera(srt_ctx* ctx, const float position[3], const float lower_left[3], const float horizontal[3], const float vertical[3]);
/* The Material objects referenced by the surfaces (RaytracingClasses.cs:221-262). */
int  srt_set_materials(srt_ctx* ctx, const srt_material* materials, uint32_t n);
/* world.surfaces entries of type Sphere (RaytracingClasses.cs:449-500, fields :497-499), structure of arrays. */
int  srt_set_spheres(srt_ctx* ctx, const float* cx, const float* cy, const float* cz, const float* radius, const uint32_t* material, uint32_t n);
/* world.surfaces entries of type RectAxis (RaytracingClasses.cs:503-544), structure of arrays. */
int  srt_set_rects(srt_ctx* ctx, const float* x1, const float* x2, const float* y1, const float* y2, const float* k, const uint32_t* material, uint32_t n);
/* world.surfaces entries of type Tri (RaytracingClasses.cs:547-680) as ModelLoader builds them
 * (ModelLoader.cs:39-64): pos9 = A,B,C positions, nrm9 = their normals, 9 floats per triangle;
 * flags bit0 = flipNormal, bit1 = blend (NULL = blend only, the loader's setting). */
int  srt_set_triangles(srt_ctx* ctx, const float* pos9, const float* nrm9, const uint32_t* material, const uint8_t* flags, uint32_t n);
/* lights.lights (LightList of PointLight, RaytracingClasses.cs:195-203; Game1.cs:140-144). */
int  srt_set_lights(srt_ctx* ctx, const srt_point_light* lights, uint32_t n);
/* The order of world.surfaces (RaytracingClasses.cs:340), one ordered list across the three types: SurfaceList.hit scans
 * it in that order and lets the later surface win an exact tie (:298-316), and the BVHNode constructor splits it by
 * position (:387-425), so the order decides the shape of the reference tree.  Entry i names the i-th surface added to
 * the world (world.surfaces.Add, Game1.cs:117-134; ModelLoader.cs:60): type[i] = 0 triangle, 1 sphere, 2 rect, and
 * index[i] = its position in the array given to srt_set_triangles / srt_set_spheres / srt_set_rects.  Every surface of
 * the world has to be named exactly once (srt_commit_scene checks it).  Optional: without it, or after a call with n = 0,
 * the order is all triangles, then all spheres, then all rects -- the statement order of Game1.cs:112-134. */
int  srt_set_surface_order(srt_ctx* ctx, const uint8_t* type, const uint32_t* index, uint32_t n);
/* world.Init() + new BVHNode(world) (Game1.cs:136-137): flattens the scene, builds and linearises the BVH
 * (32-byte nodes) and uploads everything. */
int  srt_commit_scene(srt_ctx* ctx, int world_kind);

/* Replaces Parallel.For(0, N, startThread) + renderBlock + WritePixel (Game1.cs:165, 264-317;
 * PixelManagement.cs:8-23).  Blocking; rgba_out is the Color[] buffer (packed R | G<<8 | B<<16 | A<<24)
 * sized for the layout: width*height pixels for SRT_LAYOUT_FRAME, srt_local_pixels() for SRT_LAYOUT_TILES.
 * All rows are rendered (the reference leaves the last height mod N rows black, Game1.cs:378).
 * A context with several devices splits the frame itself (SRT_LAYOUT_FRAME with world_size <= 1; tile_w / tile_h
 * choose the tile, 0 = 32): each device traces its tiles, the tiles are gathered on the first device, a kernel puts
 * them into the row-major frame and one copy brings it to rgba_out.  The picture does not depend on the number of
 * devices.  srt_render, srt_render_counted, srt_render_async / srt_poll / srt_wait, srt_render_progressive (tiles gathered
 * and assembled after every pass), srt_render_mix_buffers (every device renders whole one-sample buffers, the first
 * device mixes them in buffer order) and srt_get_stats work on all devices; srt_render_device,
 * srt_render_device_static, srt_assemble_tiles and the post-process filters run on the first device alone. */
int  srt_render(srt_ctx* ctx, const srt_render_params* p, uint32_t* rgba_out);
/* Same, writing to device memory on the caller's HIP stream (hipStream_t passed as void*; NULL is HIP's
 * default stream) without synchronising: the host side of a multi-GPU job keeps the tiles on the device
 * for the gather.  capacity_pixels is the size of d_rgba_out in pixels.  The counters of such launches
 * accumulate until srt_get_stats is called. */
int  srt_render_device(srt_ctx* ctx, const srt_render_params* p, void* d_rgba_out, size_t capacity_pixels, void* hip_stream);
/* No counterpart in the reference.  srt_render_device for callers that cannot allocate device memory while they
 * call it (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): the sample buffer,
 * the running sums and the work counter live in a static array of SRT_STATIC_SCRATCH_BYTES that every GPU the
 * library runs on holds until the process ends, and the call makes no hipMalloc / hipFree or other allocating call.
 * The array has to hold 512 bytes plus either every sample of the render (12 bytes per sample and pixel of the rank,
 * tile padding included) or, for a render in several passes, 24 bytes per pixel (one sample and the running sums):
 * renders of up to (SRT_STATIC_SCRATCH_BYTES - 512) / 24 = 2,796,181 pixels always fit, and one that fits neither
 * way is refused with SRT_E_LIMIT.  Output, counters (srt_get_stats) and stream behaviour are srt_render_device's; a
 * context with several devices runs it on its first.
 * All calls on one GPU share that GPU's array, whatever their context: calls that overlap in time on one GPU (different
 * streams, different contexts) race and give wrong pictures. */
#define SRT_STATIC_SCRATCH_BYTES (64u << 20)
int  srt_render_device_static(srt_ctx* ctx, const srt_render_params* p, void* d_rgba_out, size_t capacity_pixels, void* hip_stream);
/* srt_render with the event counters of srt_stats filled (a separately compiled, slower kernel). */
int  srt_render_counted(srt_ctx* ctx, const srt_render_params* p, uint32_t* rgba_out);

/* Replaces the threadArray[i].IsAlive polling of Game1.Update (Game1.cs:188-201): start a render,
 * ask whether it is done, wait for it.  rgba_out must stay valid until done.
 * The first render of a newly committed scene whose tree is walked with a stack may run in two stages: srt_render_async
 * queues the first sample of every pixel, and the srt_poll (or srt_wait) that finds it finished reads how many rays a
 * sample of this scene takes, queues the remaining samples for the tracer that suits the count, and reports "not done".
 * A host that calls srt_poll every frame, as Game1.Update does, notices nothing; one that never polls must call
 * srt_wait.  The frame is the same either way. */
int  srt_render_async(srt_ctx* ctx, const srt_render_params* p, uint32_t* rgba_out);
int  srt_poll(srt_ctx* ctx, int* done);
int  srt_wait(srt_ctx* ctx);
/* The picture filling in while it renders (Game1.Update uploads the shared Color[] every frame while the worker
 * threads write it, Game1.cs:188-201): like srt_render_async, but the samples are traced in at least `passes`
 * (1..64) passes and after each one a preview of the picture so far -- the running sums scaled by samples / samples
 * done -- is copied to host memory.  srt_progress reports how many passes have finished; srt_read_preview copies the
 * newest preview (it may mix two consecutive passes if the next copy is landing) and needs at least one finished
 * pass.  srt_poll / srt_wait complete the render as usual; the final frame is the one srt_render gives. */
int  srt_render_progressive(srt_ctx* ctx, const srt_render_params* p, uint32_t* rgba_out, int passes);
int  srt_progress(srt_ctx* ctx, int* passes_done, int* passes_total);
int  srt_read_preview(srt_ctx* ctx, uint32_t* rgba_out);

/* The reference's second threading technique, `mixBufferTechnique` (Game1.cs:56, :174-175): Parallel.For(startThreadFull)
 * gives each of n_buffers threads a Color[] of its own, renderBlockFull (:320-374) fills it with ONE sample per pixel --
 * the colour still divided by p->samples (:352), then gamma and packing -- and Game1.Update mixes the buffers into
 * buffer1Data one after the other in 8-bit colour: new Color(buffer1Data.ToVector3() / 2 + buffer.ToVector3() / 2)
 * (:209-220), starting from default (zero) Colors.  Buffer b is traced with sample index b of the counter RNG (thread
 * b's System.Random).  Blocking; SRT_LAYOUT_FRAME, world_size 1 and SRT_MODE_COLOR only.  A context with several devices
 * renders buffer b on device b % n_devices and mixes on the first device, in buffer order: the same picture. */
int  srt_render_mix_buffers(srt_ctx* ctx, const srt_render_params* p, int32_t n_buffers, uint32_t* rgba_out);

/* The post-process filters of the finished picture: Sorting.BlurArray (MiscFunctions.cs:124-191), Sorting.ContrastBlurArray
 * (:193-335) and Sorting.Despeckle (:337-416), which Game1.Update runs blurPasses times on buffer1Data once the render is
 * done (Game1.cs:226-232).  `array` is the caller's row-major width x height picture in the packing of srt_render
 * (MonoGame Color); it is filtered in place, `passes` times (>= 1), with the reference's in-place semantics: a pixel sees
 * the new values of its left neighbour and of the row above.  Only interior pixels change.  `weight` is clamped to
 * [0, 1] as the reference does; `contrast` / `threshold` are in byte units (0..255) of 0.3 R + 0.6 G + 0.11 B.
 * Blocking; not allowed while an asynchronous render is in flight. */
enum { SRT_POST_BLUR = 0, SRT_POST_CONTRAST_BLUR = 1, SRT_POST_DESPECKLE = 2 };
int  srt_blur_array(srt_ctx* ctx, uint32_t* array, int32_t width, int32_t height, float weight, int32_t passes);
int  srt_contrast_blur_array(srt_ctx* ctx, uint32_t* array, int32_t width, int32_t height, float weight, float contrast, int32_t passes);
int  srt_despeckle(srt_ctx* ctx, uint32_t* array, int32_t width, int32_t height, float weight, float threshold, int32_t passes);

/* Counters of the renders since the previous collection (a blocking render collects, so after srt_render
 * these are that render's counters).  Waits for the most recent launch.  With several devices the counts are
 * sums and kernel_ms is the longest device's; srt_get_device_stats gives one device's share (index = position in
 * the device list) of what srt_get_stats last collected. */
int  srt_get_stats(srt_ctx* ctx, srt_stats* out);
int  srt_get_device_stats(srt_ctx* ctx, int32_t index, srt_stats* out);

/* Size in pixels of the buffer a rank renders into for these parameters (tile-major, padded tiles). */
int64_t srt_local_pixels(const srt_render_params* p);
/* Rank 0 of a multi-GPU job: scatter the gathered per-rank tile buffers (rank r's buffer at
 * d_gathered + r*stride_pixels) into the row-major frame, i.e. what the shared Color[] of Game1.cs:157
 * is for the reference's threads.  Device pointers; runs on hip_stream without synchronising. */
int  srt_assemble_tiles(srt_ctx* ctx, const srt_render_params* p, const void* d_gathered, size_t stride_pixels, void* d_frame, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* SRT_H */
