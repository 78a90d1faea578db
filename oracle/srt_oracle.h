/*
 * srt_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the parity checker and the CPU
 * baseline for the MI355X trace-loop library (libsrt.so).  Nothing in the
 * product path may include, link or call it: only tests/, the smoke check and
 * the cpu_baseline leg of bench.py do.
 *
 * PARITY UNPINNED: the reference program (C#, MonoGame) ships no tests, golden
 * vectors or fixtures and cannot be built offline, so the restatement below is
 * pinned only by hand-derived known-answer tests, not by outputs of the real
 * reference binary.
 */
#ifndef SRT_ORACLE_H
#define SRT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Plain-data mirrors of the reference's scene classes (post-constructor state). */
typedef struct { float albedo[3]; float metalness; float smoothness; float emission[3]; } orc_material; /* RaytracingClasses.cs:221-262 */
typedef struct { float origin[3]; float radius; int32_t material; } orc_sphere;                          /* RaytracingClasses.cs:449-500 */
typedef struct { float x1, x2, y1, y2, k; int32_t material; } orc_rect;                                  /* RaytracingClasses.cs:503-544 */
typedef struct { float pos[9]; float nrm[9]; int32_t material; int32_t flip; int32_t blend; } orc_tri;   /* RaytracingClasses.cs:547-692 */
typedef struct { float position[3]; float brightness; float color[3]; } orc_light;                      /* RaytracingClasses.cs:75-94,190-192 */
typedef struct { float position[3]; float lower_left[3]; float horizontal[3]; float vertical[3]; } orc_camera; /* RaytracingClasses.cs:32-64 */

enum { ORC_WORLD_BVHNODE = 0, ORC_WORLD_SURFACELIST = 1 };
enum { ORC_MODE_COLOR = 0, ORC_MODE_NORMAL = 1, ORC_MODE_DEPTH = 2, ORC_MODE_LIGHTS = 3, ORC_MODE_DIFFUSE = 4, ORC_MODE_REFLECTION = 5 };

typedef struct {
    int32_t  width, height;      /* full image size (globalWidth/globalHeight, Game1.cs:283-284) */
    int32_t  samples;            /* Game1.cs:41 */
    int32_t  max_depth;          /* Game1.cs:42 */
    float    color_clamp;        /* Game1.cs:43 */
    uint32_t seed;               /* key of the shared counter RNG (replaces System.Random) */
    int32_t  mode;               /* ORC_MODE_* */
    int32_t  world_kind;         /* ORC_WORLD_*: which Surface GetRayColor is handed (Game1.cs:293 passes nodeWorld) */
    int32_t  threads;            /* 0 = hardware concurrency (Game1.cs:89) */
    int32_t  drop_remainder_rows;/* 1 = reproduce the H mod N unrendered rows of Game1.cs:378 */
    int32_t  x0, y0, x1, y1;     /* render window [x0,x1) x [y0,y1); all zero = full frame */
} orc_params;

typedef struct {
    uint64_t rays_path;          /* GetRayColor calls with depth >= 1 (one closest-hit query each) */
    uint64_t rays_shadow;        /* shadow rays (RaytracingClasses.cs:174) */
    uint64_t rays_redundant;     /* per-light re-intersections (RaytracingClasses.cs:211) */
    uint64_t box_tests;          /* AABB.hit calls */
    uint64_t sphere_tests, tri_tests, rect_tests;
    uint64_t hit_nodes;          /* GetRayColor calls that hit something */
    double   seconds;            /* wall time of the render call */
    int32_t  threads_used;
    int32_t  reserved;
    /* the part of box_tests / sphere_tests / tri_tests / rect_tests that repeats a test whose answer is known: all of a
       per-light re-intersection (RaytracingClasses.cs:211) and the second test of a single-surface leaf (:362-365).
       An implementation that reuses those answers performs exactly (tests - repeats) tests. */
    uint64_t box_repeats, sphere_repeats, tri_repeats, rect_repeats;
    /* Shadow rays that cannot change what PointLight.calculateLight returns (light behind the surface, all factors
       finite; the conditions are spelled out there), and the part of the four *_tests that belongs to their traversals
       and is not counted under *_repeats already.  The oracle traces them all the same; an implementation that leaves
       them out performs (tests - repeats - inert) tests. */
    uint64_t shadow_inert, box_inert, sphere_inert, tri_inert, rect_inert;
} orc_stats;

typedef struct orc_scene orc_scene;

/* Builds world (SurfaceList, insertion order: triangles, spheres, rects -- the statement
 * order of Game1.cs:112-134), runs world.Init() and, when there is no rect, new BVHNode(world). */
orc_scene* orc_scene_create(const orc_material* mats, int32_t n_mats,
                            const orc_tri* tris, int32_t n_tris,
                            const orc_sphere* spheres, int32_t n_spheres,
                            const orc_rect* rects, int32_t n_rects,
                            const orc_light* lights, int32_t n_lights);
/* The same with world.surfaces in the caller's order: entry r names the r-th surface added (type 0 triangle, 1 sphere,
 * 2 rect; index into that type's array); every surface exactly once.  n_order = 0: the order above. */
orc_scene* orc_scene_create_ordered(const orc_material* mats, int32_t n_mats, const orc_tri* tris, int32_t n_tris,
                                    const orc_sphere* spheres, int32_t n_spheres, const orc_rect* rects, int32_t n_rects,
                                    const orc_light* lights, int32_t n_lights,
                                    const uint8_t* order_type, const uint32_t* order_index, int32_t n_order);
void orc_scene_destroy(orc_scene*);
/* number of BVHNode objects / max depth of the reference tree (0 if not built) */
int32_t orc_scene_bvh_nodes(const orc_scene*);
int32_t orc_scene_bvh_depth(const orc_scene*);
/* Primitive ids (tri i -> i, sphere j -> n_tris + j) in the left-to-right leaf order of the
 * reference BVHNode tree; returns the number written (<= cap). */
int32_t orc_scene_bvh_leaf_order(const orc_scene*, int32_t* out, int32_t cap);

/* 0 ok; <0 error (-1 bad argument, -2 the reference would throw for this configuration). */
int orc_render(const orc_scene*, const orc_camera*, const orc_params*, uint32_t* rgba_out,
               uint32_t* nodes_per_pixel_or_null, orc_stats* stats_or_null);
/* The mix-buffer technique (Game1.cs:174-175, :209-220, :320-388): n_buffers one-sample renders of the whole picture,
 * each divided by p->samples as the reference does, mixed pairwise into `out` in 8-bit colour. */
int orc_render_mix_buffers(const orc_scene* s, const orc_camera* cam, const orc_params* p, int32_t n_buffers, uint32_t* out, orc_stats* stats);
const char* orc_last_error(void);
/* Tests only: renders made while bit 0 is set sum the lights plainly (f <- f + c) instead of the reference's
 * f <- 2 f + c (RaytracingClasses.cs:213) -- a negative control for the comparison with the reference's own picture. */
void orc_set_misreading(int bits);

/* Constructors of the host-side classes, for known-answer tests. */
void orc_camera_ctor(const float origin[3], const float look_at[3], float fov_deg, float ratio, orc_camera* out);
void orc_material_ctor(const float albedo[3], float metal, float smooth, const float emission[3], orc_material* out);
void orc_pointlight_ctor(const float origin[3], float brightness, const float* color_or_null, orc_light* out);

/* Single-function probes for known-answer tests.  rec = {t, Px,Py,Pz, Nx,Ny,Nz, frontFace, material}. */
int orc_sphere_hit(const orc_sphere*, const float o[3], const float d[3], float t_min, float t_max, float rec[9]);
int orc_rect_hit(const orc_rect*, const float o[3], const float d[3], float t_min, float t_max, float rec[9]);
int orc_tri_hit(const orc_tri*, const float o[3], const float d[3], float t_min, float t_max, float rec[9]);
int orc_aabb_hit(const float bmin[3], const float bmax[3], const float o[3], const float d[3]);
void orc_sky(const float d[3], float rgb[3]);
int  orc_depth_heuristic(int depth, float smoothness);
uint32_t orc_pack_color(float r, float g, float b);
/* Sorting.BlurArray / ContrastBlurArray / Despeckle (MiscFunctions.cs:124, :193, :337): one in-place pass over a
 * row-major packed Color array (R | G<<8 | B<<16 | A<<24). */
void orc_blur_array(uint32_t* array, int32_t width, int32_t height, float weight);
void orc_contrast_blur_array(uint32_t* array, int32_t width, int32_t height, float weight, float contrast);
void orc_despeckle(uint32_t* array, int32_t width, int32_t height, float weight, float threshold);
void orc_rng_words(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t node, uint32_t call, uint32_t out[4]);
/* Random.Next(lo, lo + range) as the next base-`range` digit of *word (which is advanced). */
int  orc_rng_next(uint32_t* word, int lo, int range);
/* RandomVector.ReturnRandomVector from one RNG word (three base-200 digits). */
void orc_random_vector(uint32_t word, float out[3]);
void orc_hemisphere_vector(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t node, const float n[3], float out[3]);
float orc_angle_between(const float a[3], const float b[3]);
/* LightList.calculateLight on a single hit; returns 0 if the primary ray misses. */
int orc_get_lights(const orc_scene*, int world_kind, const float o[3], const float d[3],
                   uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t node, float rgb[3]);
/* One GetRayColor call from an arbitrary ray (node id 1 = a primary ray). */
void orc_get_ray_color(const orc_scene*, int world_kind, const float o[3], const float d[3], int depth, float clamp,
                       uint32_t seed, uint32_t pixel, uint32_t sample, float rgb[3]);
void orc_get_debug_color(const orc_scene*, int world_kind, int mode, const float o[3], const float d[3], int depth, uint32_t seed,
                         uint32_t pixel, uint32_t sample, float rgb[3]);   /* GetRayColorDiffuse / GetRayColorReflection */

#ifdef __cplusplus
}
#endif
#endif
