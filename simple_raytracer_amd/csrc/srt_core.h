// srt_core.h -- building blocks of the trace loop: vector math, RNG, scene records, primitive and
// box tests, closest-hit queries, hit records and the shading formulas.
//
// Everything here is `SRT_HD` (host + device) so that the same source runs inside the gfx950
// kernels (srt_kernels.hip) and inside a CPU harness used by the tests to debug kernel logic
// without a GPU.  The harness is not part of the product library.
//
// Arithmetic contract: every float operation that feeds a comparison, a ray or a colour is
// written in the operation order of the reference C# (file:line cited at each function) and
// must be compiled with floating-point contraction off and correctly rounded division and
// square root.  Under that contract the image is a pure function of (scene, camera,
// parameters, seed) and is reproduced bit for bit by any conforming CPU or GPU.
#ifndef SRT_CORE_H
#define SRT_CORE_H

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SRT_HD __host__ __device__ __forceinline__
#else
#define SRT_HD inline
#endif

namespace srt {

// ------------------------------------------------------------------------------------------
// small vector type; op order follows MonoGame's Vector3 (3.8.0) helpers
// ------------------------------------------------------------------------------------------
struct f3 { float x, y, z; };
struct q4 { float x, y, z, w; };  // one 16-byte device word

SRT_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
SRT_HD f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
SRT_HD f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
SRT_HD f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
SRT_HD f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
SRT_HD f3 mul3(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }

SRT_HD uint32_t fbits(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
SRT_HD float bitsf(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }

// Correctly rounded square root, reciprocal, 1 / sqrt (the factor of Vector3.Normalize: two roundings) and the Smith
// term 2 / (1 + sqrt(1 + t)).
//
// The plain operators are IEEE on the host, and on the device under -fhip-fp32-correctly-rounded-divide-sqrt (HIP's
// __fsqrt_rn() is the native, approximate v_sqrt_f32 unless OCML_BASIC_ROUNDED_OPERATIONS is defined, and differs from
// IEEE in ~40 % of inputs by one ulp).  The compiler's device sequences are general: 11 instructions for a division and
// 16 for a square root, most of them scaling and fix-ups for operands at the ends of the exponent range.  A ray tree
// node costs about forty of these, so the device has a short form for operands of ordinary size:
//   1 / x    r = v_rcp_f32(x), then one Newton step in fused arithmetic: r + r (1 - x r)
//   sqrt x   y = v_rsq_f32(x), g = x y, then one step: g + (y / 2) (x - g g)
// A short form returns the IEEE result for EVERY float that passes its acceptance test, and the tests look at the
// result: a reciprocal that is a normal number; a 1 / sqrt below 2^46 (radicand above ~2^-92).  Zeros, denormals,
// infinities, NaNs, negative radicands and the ends of the exponent range fail them and get the plain operator.
// tests/gpu_probe/exhaustive proves this on all 2^32 bit patterns, against the compiler's general sequences and,
// through a checksum over the same 2^32 inputs, against the host's IEEE operations.  These are not approximations.
// The fallback is a branch, and in straight-line shading code a branch costs about what a short form saves (the
// compiler no longer moves, merges or hoists code across it, and the common case is the taken branch).  Measured on the
// 1080p 16-spp frame of the Game1 scene, 15.07 ms with the plain operators everywhere: every reciprocal, root and
// 1 / sqrt short WITHOUT acceptance tests (not exact; the bound) 13.76 ms; all of them with tests 15.1-15.8 ms; whole
// functions run speculatively with the refusals collected in a flag and redone with the plain operators when it is
// set 16.7 ms (the inputs of a function stay live for the redo: 59 spilled registers).  What pays is a test that
// covers several operations -- the 1 / sqrt of a normalisation (27 -> 10 instructions), the three reciprocals of a ray
// direction under one branch -- and the Smith term, which needs none: 14.77 ms (the reference's own settings, 720p 500
// spp depth 4: 95.9 -> 93.6 ms).
#if defined(__HIP_DEVICE_COMPILE__)
SRT_HD float rcp_short(float x)
{
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
SRT_HD float sqrt_short(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    return __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
}
SRT_HD bool rcp_accept(float r) { return __builtin_amdgcn_classf(r, 0x108); }   // -normal | +normal
SRT_HD bool rcp_sqrt_accept(float f) { return f < 0x1p46f; }                     // false for NaN
SRT_HD float rcp_sqrt_rn(float s) { const float f = rcp_short(sqrt_short(s)); return rcp_sqrt_accept(f) ? f : 1.0f / __builtin_sqrtf(s); }
// 2 / (1 + sqrt(1 + t)) for t in [+0, +infinity] or NaN (the Smith term: t = r2 (1 - c^2) / c^2 with c in [0, 1] and
// r2 a square).  The radicand is then >= 1 and the denominator in [2, 2^65], inside the range of both short forms,
// except for an infinite t, whose term is +0; doubling is exact.  No acceptance test is needed: the exhaustive test
// runs this function on every float t >= +0 and on all NaNs.
SRT_HD float smith_term(float t)
{
    const float s = 2.0f * rcp_short(1.0f + sqrt_short(1.0f + t));
    return t == INFINITY ? 0.0f : s;
}
#else
SRT_HD float rcp_sqrt_rn(float s) { return 1.0f / __builtin_sqrtf(s); }
SRT_HD float smith_term(float t) { return 2.0f / (1.0f + __builtin_sqrtf(1.0f + t)); }
#endif
SRT_HD float sqrt_rn(float v) { return __builtin_sqrtf(v); }
SRT_HD float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
SRT_HD float lensq3(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
SRT_HD f3 cross3(f3 a, f3 b)
{
    return mk3(a.y * b.z - b.y * a.z, -(a.x * b.z - b.x * a.z), a.x * b.y - b.x * a.y);
}
// Vector3.Normalize: factor = 1 / sqrt(x^2 + y^2 + z^2), then three multiplies.
SRT_HD f3 normalize3(f3 a)
{
    float f = rcp_sqrt_rn((a.x * a.x) + (a.y * a.y) + (a.z * a.z));
    return mk3(a.x * f, a.y * f, a.z * f);
}
// (1 / d.x, 1 / d.y, 1 / d.z) -- CustomRay.dirFrac
SRT_HD f3 rcp3_rn(f3 d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const f3 r = mk3(rcp_short(d.x), rcp_short(d.y), rcp_short(d.z));
    if (rcp_accept(r.x) & rcp_accept(r.y) & rcp_accept(r.z)) return r;
#endif
    return mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
}
SRT_HD float distsq3(f3 a, f3 b)
{
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return dx * dx + dy * dy + dz * dz;
}
// Vector3.Reflect(v, n) = v - (2 n) (v . n)
SRT_HD f3 reflect3(f3 v, f3 n)
{
    float d = ((v.x * n.x) + (v.y * n.y)) + (v.z * n.z);
    return mk3(v.x - (2.0f * n.x) * d, v.y - (2.0f * n.y) * d, v.z - (2.0f * n.z) * d);
}
SRT_HD float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }  // System.Math.Clamp: NaN passes
// new Color(float, float, float): (int)(c*255) truncated, clamped to a byte; NaN and values
// beyond the int range convert to int.MinValue on x64 and therefore clamp to 0.  Alpha 255.
SRT_HD uint32_t to_byte(float c)
{
    float v = c * 255.0f;
    if (!(v >= 0.0f && v < 2147483648.0f)) return 0u;
    int32_t i = (int32_t)v;
    return i > 255 ? 255u : (uint32_t)i;
}
SRT_HD uint32_t pack_color(float r, float g, float b) { return to_byte(r) | (to_byte(g) << 8) | (to_byte(b) << 16) | 0xFF000000u; }

// ------------------------------------------------------------------------------------------
// Counter-based RNG.  It stands in for System.Random, which the reference seeds from entropy
// per thread (Game1.cs:44,378) and which therefore cannot be reproduced.
//
//   call(key, c)   four 32-bit words = pcg4d(pixel, sample, node, c + 0x9E3779B9 * seed)
//   digit(w, B)    d = floor(w * B / 2^32), then w <- (w * B) mod 2^32: successive calls peel the
//                  base-B digits of the fraction w / 2^32, which are independent and uniform on
//                  [0, B) -- Random.Next(lo, lo + B) is lo + such a digit.
//   vector(w)      three base-200 digits d1 d2 d3 -> ((d1-100)/100, (d2-100)/100, (d3-100)/100),
//                  i.e. the three Next(-100,100)/100 draws of RandomVector.ReturnRandomVector
//                  (MiscFunctions.cs:12-15), X first.
// Which word feeds what, inside one ray-tree node:
//   call 0 word 0         the pixel jitter (node 0: two base-100 digits, u first, Game1.cs:283-284)
//                         or the reflection roughness vector (RaytracingOperations.cs:128)
//   call (1+l)>>2, word (1+l)&3     the shadow jitter of light l (RaytracingClasses.cs:174)
//   call 0x10000 + (i>>2), word i&3 the i-th try of the unit-sphere rejection loop (MiscFunctions.cs:26-34)
// Node ids: primary ray = 1; child k (0 diffuse, 1 metal, 2 reflect) of node n = 4n + k + 1.
// The value (d-100)/100 is read from a 200-entry table filled with exactly that division.
// ------------------------------------------------------------------------------------------
enum : uint32_t { CALL_AUX = 0u, CALL_SPHERE0 = 0x10000u };
enum { MAX_REJECTION_TRIES = 64, UNIT_TABLE = 200 };

struct RngKey { uint32_t seed, pixel, sample, node; };

SRT_HD void rng_call(const RngKey& k, uint32_t call, uint32_t w[4])
{
    uint32_t x = k.pixel, y = k.sample, z = k.node, t = call + 0x9E3779B9u * k.seed;
    x = x * 1664525u + 1013904223u;
    y = y * 1664525u + 1013904223u;
    z = z * 1664525u + 1013904223u;
    t = t * 1664525u + 1013904223u;
    x += y * t; y += z * x; z += x * y; t += y * z;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; t ^= t >> 16;
    x += y * t; y += z * x; z += x * y; t += y * z;
    w[0] = x; w[1] = y; w[2] = z; w[3] = t;
}
SRT_HD uint32_t rng_digit(uint32_t& w, uint32_t base)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d = __umulhi(w, base);
    w = w * base;
    return d;
#else
    uint64_t p = (uint64_t)w * (uint64_t)base;
    w = (uint32_t)p;
    return (uint32_t)(p >> 32);
#endif
}
// entry d of the table: (float)(d - 100) / 100
SRT_HD void fill_unit_table(float* table)
{
    for (int d = 0; d < UNIT_TABLE; d++) table[d] = (float)(d - 100) / 100.0f;
}
// RandomVector.ReturnRandomVector from one word
SRT_HD f3 vector_from_word(uint32_t w, const float* unit)
{
    uint32_t d1 = rng_digit(w, 200u), d2 = rng_digit(w, 200u), d3 = rng_digit(w, 200u);
    return mk3(unit[d1], unit[d2], unit[d3]);
}
// RandomVector.ReturnRandomRangedVector (MiscFunctions.cs:18-23)
SRT_HD f3 ranged_vector_from_word(uint32_t w, float multiplier, const float* unit)
{
    float factor = 1.0f - clampf(multiplier, 0.0f, 1.0f);
    f3 r = vector_from_word(w, unit);
    return mk3(r.x * factor, r.y * factor, r.z * factor);
}
// word of light l's shadow jitter, given the node's call-0 words
SRT_HD uint32_t light_word(const RngKey& k, const uint32_t aux[4], uint32_t l)
{
    uint32_t j = 1u + l;
    if (j < 4u) return aux[j];
    uint32_t w[4];
    rng_call(k, CALL_AUX + (j >> 2), w);
    return w[j & 3u];
}
// RandomVector.ReturnHemisphereRandomVector over ReturnRandomUnitSphereVector (MiscFunctions.cs:26-48).
// The rejection loop is bounded at MAX_REJECTION_TRIES (probability of getting there: (1 - pi/6)^64 ~ 2e-21).
SRT_HD f3 random_unit_sphere_vector(const RngKey& k, const float* unit)
{
    f3 p = mk3(0.0f, 0.0f, 0.0f);
    bool found = false;
    for (uint32_t c = 0; c < (uint32_t)MAX_REJECTION_TRIES / 4u && !found; c++) {
        uint32_t w[4];
        rng_call(k, CALL_SPHERE0 + c, w);
        for (int i = 0; i < 4 && !found; i++) {
            p = vector_from_word(w[i], unit);
            found = !(lensq3(p) >= 1.0f);
        }
    }
    return p;
}
SRT_HD f3 random_hemisphere_vector(const RngKey& k, f3 normal, const float* unit)
{
    f3 p = random_unit_sphere_vector(k, unit);
    if (dot3(p, normal) > 0.0f) return p;
    return neg3(p);
}

// ------------------------------------------------------------------------------------------
// Device scene: 16-byte words, one gather-friendly record per primitive / node.
// ------------------------------------------------------------------------------------------
enum : uint32_t { PRIM_TRI = 0u, PRIM_SPHERE = 1u, PRIM_RECT = 2u };
enum : uint32_t { INFO_FLIP = 4u, INFO_BLEND = 8u };
enum : uint32_t { DESC_LEAF = 0x80000000u, NO_PRIM = 0xFFFFFFFFu };
#ifndef SRT_STACK_DEPTH
#define SRT_STACK_DEPTH 36
#endif
enum { STACK_DEPTH = SRT_STACK_DEPTH, MAX_LEAF_PRIMS = 16, FRAME_SLOTS = 16, TINY_TREE_NODES = 64 };

// Primitive record, 3 words (48 bytes):
//   tri:    (A.x A.y A.z AB.x) (AB.y AB.z AC.x AC.y) (AC.z  info order normal_index)
//   sphere: (c.x c.y c.z r)    (-)                   (-     info order -)
//   rect:   (x1 x2 y1 y2)      (k - - -)             (-     info order -)
// info = type | flip<<2 | blend<<3 | material<<8.  order = rank used to break exact ties in t
// (the surface the reference tests later wins).
// BVH node, 2 words (32 bytes): (min.xyz, desc) (max.xyz, -).  desc of an inner node = index
// of its left child (right = left + 1); desc of a leaf = DESC_LEAF | (count-1)<<27 | first.
// Triangle normals, 3 words per triangle: (nA.xyz -) (nB.xyz -) (nC.xyz -).
// Material, 2 words: (albedo.xyz metalness) (smoothness emission.xyz).
// Light, 2 words: (position.xyz brightness) (color.xyz -).
// Scene tables that the kernels read with wave-uniform indices (nodes, the primitives of a leaf, lights).  Device code
// sees them in the constant address space: no kernel ever writes a scene table, and only with that stated in the
// pointer type do uniform reads stay scalar loads in a kernel that also stores to other global memory (the compiler
// cannot prove that those stores leave the tables alone, and falls back to vector loads plus readfirstlane).
#if defined(__HIP_DEVICE_COMPILE__)
typedef float table_f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) table_f4* scene_table;
SRT_HD scene_table as_scene_table(const q4* p) { return (scene_table)p; }
SRT_HD q4 table_word(scene_table t, uint32_t i) { const table_f4 v = t[i]; q4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r; }
#else
typedef const q4* scene_table;
SRT_HD scene_table as_scene_table(const q4* p) { return p; }
SRT_HD q4 table_word(scene_table t, uint32_t i) { return t[i]; }
#endif

struct SceneView {
    scene_table nodes;
    scene_table prims;
    scene_table normals;
    const q4* materials;
    scene_table lights;
    uint32_t n_lights;
    uint32_t list_first, list_count;  // primitives scanned linearly (rects; everything in tiny scenes)
    uint32_t has_bvh;                 // nodes[0] is the root when set
    uint32_t n_wide;                  // non-zero: `nodes` holds that many 64-byte wide nodes (4 words each) instead
    // Small reference trees only: the same nodes in depth-first (pre)order, 2 words each:
    // (min.xyz, desc) (max.xyz, index of the first node after this node's subtree).
    scene_table tiny;
    uint32_t n_tiny;
    uint32_t n_nodes, n_prims, n_materials;  // table sizes (nodes / primitive records / materials)
    // The primitive records again, for the per-lane lookups of the winning hit (same data as `prims`; the kernels
    // point it at an LDS copy when the scene is small).
    const q4* hit_prims;
    // Far-origin guard of the library's own tree (see trav_begin): the ball around the surfaces in the tree, the
    // largest origin distance L0 the boxes were padded for, and the padding per unit of distance beyond it.
    f3 far_center;
    float far_radius, far_L0, far_slope;
    // Non-zero: the kernel of this scene leaves out the shadow rays that cannot change their light's contribution
    // (shadow_ray_matters); the host sets it for scenes whose queries are stack traversals and whose material and
    // light values are all of ordinary size.
    uint32_t inert_lights;
};

struct Material { f3 albedo; float metalness; float smoothness; f3 emission; };
SRT_HD Material load_material(const SceneView& sc, uint32_t idx)
{
    q4 a = sc.materials[2 * idx], b = sc.materials[2 * idx + 1];
    Material m;
    m.albedo = mk3(a.x, a.y, a.z); m.metalness = a.w; m.smoothness = b.x; m.emission = mk3(b.y, b.z, b.w);
    return m;
}

struct Counters { uint32_t box_tests, sphere_tests, tri_tests, rect_tests; };

struct Hit { float t; uint32_t prim; float u, v; uint32_t order; };

// ------------------------------------------------------------------------------------------
// Primitive tests.  Each returns true and the candidate t (and u, v for triangles) when the
// reference's hit() would return true for the range [t_min, t_max].
// ------------------------------------------------------------------------------------------
// Sphere.hit (RaytracingClasses.cs:459-487); `a` = |d|^2 is hoisted by the caller.
SRT_HD bool test_sphere(q4 s, f3 o, f3 d, float a, float t_min, float t_max, float& t_out)
{
    f3 oc = mk3(o.x - s.x, o.y - s.y, o.z - s.z);
    float half_b = dot3(oc, d);
    float c = lensq3(oc) - s.w * s.w;
    float disc = half_b * half_b - a * c;
    if (disc < 0.0f) return false;
    float sq = sqrt_rn(disc);
    float root = (-half_b - sq) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sq) / a;
        if (root < t_min || t_max < root) return false;
    }
    t_out = root;
    return true;
}
// RectAxis.hit (RaytracingClasses.cs:516-536); t exactly as written there: -k - o.z / d.z.
SRT_HD bool test_rect(q4 r0, float k, f3 o, f3 d, float t_min, float t_max, float& t_out)
{
    float t = (-k - o.z / d.z);
    if (t < t_min || t > t_max) return false;
    float x = o.x + (t * d.x);
    float y = o.y + (t * d.y);
    if (x < r0.x || x > r0.y || y < r0.z || y > r0.w) return false;
    if (t < 0.0f) return false;
    t_out = t;
    return true;
}
// Tri.hit (RaytracingClasses.cs:597-627).  The C# compares the float determinant with the
// double 1e-8; 0x1.5798eep-27f is the largest float below 1e-8, so "(double)det < 1e-8" is
// "det <= that float" and "(double)det > 1e-8" is "det > that float".
SRT_HD bool test_tri(q4 p0, q4 p1, float acz, bool flip, f3 o, f3 d, float t_min, float t_max, float& t_out, float& u_out, float& v_out)
{
    const float kEps = 0x1.5798eep-27f;
    f3 A = mk3(p0.x, p0.y, p0.z), AB = mk3(p0.w, p1.x, p1.y), AC = mk3(p1.z, p1.w, acz);
    f3 pv = cross3(d, AC);
    float det = dot3(AB, pv);
    if (!flip) { if (det <= kEps) return false; }
    else { if (det > kEps) return false; }
    if (fabsf(det) <= kEps) return false;
    float inv = 1.0f / det;
    f3 tv = sub3(o, A);
    float u = dot3(tv, pv) * inv;
    if (u < 0.0f || u > 1.0f) return false;
    f3 qv = cross3(tv, AB);
    float v = dot3(d, qv) * inv;
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = dot3(AC, qv) * inv;
    if (t < 0.0f) return false;
    if (t < t_min || t > t_max) return false;
    t_out = t; u_out = u; v_out = v;
    return true;
}

// The test of primitive `idx`, whose record (three words) the caller has already fetched.
template <bool COUNT>
SRT_HD void test_prim_record(q4 p0, q4 p1, q4 p2, uint32_t idx, f3 o, f3 d, float a, float t_min, Hit& best, Counters& cnt)
{
    uint32_t info = fbits(p2.y), order = fbits(p2.z);
    uint32_t type = info & 3u;
    float t = 0.0f, u = 0.0f, v = 0.0f;
    bool h;
    if (type == PRIM_TRI) {
        if (COUNT) cnt.tri_tests++;
        h = test_tri(p0, p1, p2.x, (info & INFO_FLIP) != 0u, o, d, t_min, best.t, t, u, v);
    } else if (type == PRIM_SPHERE) {
        if (COUNT) cnt.sphere_tests++;
        h = test_sphere(p0, o, d, a, t_min, best.t, t);
    } else {
        if (COUNT) cnt.rect_tests++;
        h = test_rect(p0, p1.x, o, d, t_min, best.t, t);
    }
    // t <= best.t here.  On an exact tie the surface the reference tests later wins
    // (SurfaceList.hit / BVHNode.hit pass the running t_max and accept t == t_max).
    if (h && (t < best.t || best.prim == NO_PRIM || order >= best.order)) {
        best.t = t; best.prim = idx; best.u = u; best.v = v; best.order = order;
    }
}
template <bool COUNT>
SRT_HD void test_prim(const SceneView& sc, uint32_t idx, f3 o, f3 d, float a, float t_min, Hit& best, Counters& cnt)
{
    // all three words at once: one memory round trip per primitive, whatever its type turns out to be
    test_prim_record<COUNT>(table_word(sc.prims, 3 * idx), table_word(sc.prims, 3 * idx + 1), table_word(sc.prims, 3 * idx + 2), idx, o, d, a, t_min, best, cnt);
}

// SurfaceList.hit (RaytracingClasses.cs:296-316) over every surface of the world, for the queries of a
// SurfaceList world that the library's own tree must not answer.  The reference scans the list in order, handing
// each surface the closest t so far as its upper bound, and a surface test can report a hit at t = NaN (every
// rejection in Tri.hit / Sphere.hit / RectAxis.hit is a comparison, and all of them are false for a NaN): that hit
// is accepted whatever came before it, and the surface after it is accepted whatever its distance.  So the scan
// returns the closest hit among the surfaces AFTER the last NaN hit (or that NaN hit, if none follows); with no NaN
// hit that is the closest hit overall, which is what the tree computes in any order.  NaN distances need an
// overflow or 0 * inf inside a test.  The tree is only built over surfaces whose coordinates stay below
// kTameRange (the triple products of a test then stay finite), and it only answers rays that start inside that
// range, with a direction of ordinary length, after the surfaces outside the tree reported no NaN hit; every other
// query comes here: two passes over the surfaces in memory order, the first finds the last NaN hit in list order.
constexpr float kTameRange = 0x1p40f;
template <bool COUNT>
SRT_HD Hit scan_all_surfaces(const SceneView& sc, f3 o, f3 d, float a, float t_min, Counters& cnt)
{
    Hit none;
    none.t = INFINITY; none.prim = NO_PRIM; none.u = 0.0f; none.v = 0.0f; none.order = 0u;
    Hit best = none;
    for (uint32_t i = 0; i < sc.n_prims; i++) {
        Hit h = none;
        test_prim<COUNT>(sc, i, o, d, a, t_min, h, cnt);
        if (h.prim != NO_PRIM && h.t != h.t && (best.prim == NO_PRIM || h.order >= best.order)) best = h;
    }
    const bool after = best.prim != NO_PRIM;
    const uint32_t last_nan = best.order;
    for (uint32_t i = 0; i < sc.n_prims; i++) {
        if (after && fbits(table_word(sc.prims, 3 * i + 2).z) <= last_nan) continue;
        test_prim<COUNT>(sc, i, o, d, a, t_min, best, cnt);
    }
    return best;
}

// ------------------------------------------------------------------------------------------
// Box tests
// ------------------------------------------------------------------------------------------
SRT_HD float min2(float a, float b) { return fminf(a, b); }
SRT_HD float max2(float a, float b) { return fmaxf(a, b); }

// AABB.hit "approach 3" (RaytracingClasses.cs:773-789): the caller's range is ignored, and
// because System.Math.Min/Max propagate NaN (0 * inf when the origin lies on a slab plane of
// an axis-parallel ray) any NaN makes both rejections false, i.e. counts as a hit.
SRT_HD bool box_hit_reference(q4 n0, q4 n1, f3 o, f3 inv)
{
    float t1 = (n0.x - o.x) * inv.x, t2 = (n1.x - o.x) * inv.x;
    float t3 = (n0.y - o.y) * inv.y, t4 = (n1.y - o.y) * inv.y;
    float t5 = (n0.z - o.z) * inv.z, t6 = (n1.z - o.z) * inv.z;
    bool any_nan = (t1 != t1) | (t2 != t2) | (t3 != t3) | (t4 != t4) | (t5 != t5) | (t6 != t6);
    float tmin = max2(max2(min2(t1, t2), min2(t3, t4)), min2(t5, t6));
    float tmax = min2(min2(max2(t1, t2), max2(t3, t4)), max2(t5, t6));
    return any_nan | !((tmax < 0.0f) | (tmin > tmax));
}
// ------------------------------------------------------------------------------------------
// Closest-hit query = Surface.hit(ray, t_min, +inf, record) on the committed world.
//   EXACT:  BVHNode.hit order and semantics (left subtree, then right; no distance culling).
//   !EXACT: SurfaceList.hit semantics through the library's own tree (a padded SAH tree collapsed to wide
//           quantised nodes), nearest child first, culled against the closest hit so far.  fminf/fmaxf drop NaNs,
//           which treats a NaN slab as "inside" -- the conservative reading.
// closest_hit() answers worlds that need no traversal stack (flat lists, tiny reference trees); trees that do are
// walked by trav_begin() / trav_run() below, with the per-lane stack in `stack_mem` at element stride STRIDE (an
// LDS column on the GPU).
// ------------------------------------------------------------------------------------------
// NEAR < STACK_DEPTH: only the first NEAR entries live in `base` (an LDS column: less LDS per wavefront, more wavefronts
// per SIMD); deeper entries, which few traversals ever reach, go to `far` (a column of device memory, same stride).
template <int STRIDE, int NEAR = STACK_DEPTH>
struct LaneStack {
    uint32_t* base;
    uint32_t* far;
    int sp;
    SRT_HD void push(uint32_t v)
    {
        if (NEAR >= STACK_DEPTH || sp < NEAR) base[sp * STRIDE] = v;
        else far[(sp - NEAR) * STRIDE] = v;
        sp++;
    }
    SRT_HD uint32_t pop()
    {
        sp--;
        if (NEAR >= STACK_DEPTH) return base[sp * STRIDE];
        // the LDS read is unconditional and the far read a branch few lanes take: a select between the two addresses
        // would be one flat load, which goes through the texture-address unit for every lane
        uint32_t v = base[(sp < NEAR ? sp : NEAR - 1) * STRIDE];
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(v));     // (keeps the compiler from merging the two reads into that select all the same)
#endif
        if (sp >= NEAR) v = far[(sp - NEAR) * STRIDE];
        return v;
    }
};

template <bool EXACT, bool COUNT>
SRT_HD Hit closest_hit(const SceneView& sc, f3 o, f3 d, float t_min, Counters& cnt)
{
    Hit best;
    best.t = INFINITY; best.prim = NO_PRIM; best.u = 0.0f; best.v = 0.0f; best.order = 0u;
    const float a = lensq3(d);  // Sphere.hit's `a`

    for (uint32_t i = 0; i < sc.list_count; i++) test_prim<COUNT>(sc, sc.list_first + i, o, d, a, t_min, best, cnt);

    if (EXACT && sc.n_tiny != 0u) {
        // A reference tree of a few nodes (the Game1 scene has 5): walk the nodes in depth-first order with a
        // wave-uniform loop.  A lane takes part in node i when its own cursor points at i; a box miss moves the
        // cursor past the subtree.  Node and primitive data then come from uniform addresses (scalar loads, no
        // per-lane gathers, no stack), and the visit order, box semantics and tie-breaks are BVHNode.hit's.
        const f3 inv = rcp3_rn(d);
        uint32_t cur = 0u;
        q4 n0 = table_word(sc.tiny, 0), n1 = table_word(sc.tiny, 1);
        for (uint32_t i = 0; i < sc.n_tiny; i++) {
            // fetch the next node while this one is tested (the array has one spare entry at the end)
            q4 m0 = table_word(sc.tiny, 2 * i + 2), m1 = table_word(sc.tiny, 2 * i + 3);
            // a leaf's first primitive is fetched while its box is tested (uniform address: a scalar load whose latency the
            // box test hides; node 0's descriptor says nothing about a primitive when the node is an inner one)
            const uint32_t desc = fbits(n0.w);
            const uint32_t first = (desc & DESC_LEAF) ? (desc & 0x07FFFFFFu) : 0u, count = ((desc >> 27) & 15u) + 1u;
            const q4 r0 = table_word(sc.prims, 3 * first), r1 = table_word(sc.prims, 3 * first + 1), r2 = table_word(sc.prims, 3 * first + 2);
            if (cur == i) {
                if (COUNT) cnt.box_tests++;
                if (box_hit_reference(n0, n1, o, inv)) {
                    cur = i + 1u;
                    if (desc & DESC_LEAF) {
                        test_prim_record<COUNT>(r0, r1, r2, first, o, d, a, t_min, best, cnt);
                        for (uint32_t k = 1; k < count; k++) test_prim<COUNT>(sc, first + k, o, d, a, t_min, best, cnt);
                    }
                } else {
                    cur = fbits(n1.w);  // first node after this subtree
                }
            }
            n0 = m0; n1 = m1;
        }
        return best;
    }

    return best;
}

// ------------------------------------------------------------------------------------------
// The same query in resumable form, for kernels that interleave the traversals of many lanes with other work: the
// state of a traversal is (ray direction, best hit so far, current node, stack height) plus the lane's stack column.
//   trav_begin   the linearly scanned primitives and the root box; scenes without a stack traversal (flat lists,
//                tiny reference trees) are answered completely here.
//   trav_run     at most `rounds` times: descend at most `steps` inner nodes, then test the leaf reached (if one was).
// A traversal is finished when tr.cur == TRAV_NONE; any split into rounds gives the result of closest_hit().
// ------------------------------------------------------------------------------------------
enum : uint32_t { TRAV_NONE = 0xFFFFFFFFu, TRAV_NEW = 0xFFFFFFFEu };   // both carry the leaf bit; no real leaf has 16 primitives
struct Trav { f3 d; Hit best; uint32_t cur; int32_t sp; float pad; };

// One child of a wide node: slab test in the node's grid.  With plane = origin + step * q the form (plane - o) * inv
// becomes q * (step * inv) + (origin - o) * inv = q * A + B, one fused multiply-add per plane.  The caller has
// already picked, per axis, which of the child's two planes the ray meets first (by the sign of the direction), so
// the entry distance is a max of three and the exit distance a min of three.  The rounding differs from the unfolded
// form by a few ulp of the distances involved, orders of magnitude below the padding of the boxes (culling is
// conservative, not part of the arithmetic contract); NaNs (0 * inf) drop out of fmaxf / fminf, i.e. never cull.
// Returns the entry distance, +inf for a miss.
SRT_HD float wide_child(uint32_t nx, uint32_t ny, uint32_t nz, uint32_t fx, uint32_t fy, uint32_t fz, int shift, f3 A, f3 Bn, f3 Bf, float t_best)
{
    float n1 = __builtin_fmaf((float)((nx >> shift) & 255u), A.x, Bn.x), f1 = __builtin_fmaf((float)((fx >> shift) & 255u), A.x, Bf.x);
    float n2 = __builtin_fmaf((float)((ny >> shift) & 255u), A.y, Bn.y), f2 = __builtin_fmaf((float)((fy >> shift) & 255u), A.y, Bf.y);
    float n3 = __builtin_fmaf((float)((nz >> shift) & 255u), A.z, Bn.z), f3_ = __builtin_fmaf((float)((fz >> shift) & 255u), A.z, Bf.z);
    // entry distance clamped to the ray's start, exit distance to the closest hit so far: one compare decides
    // (tmax < 0, tmin > tmax and tmin > t_best at once)
    float tmin = max2(max2(max2(n1, n2), n3), 0.0f);
    float tmax = min2(min2(min2(f1, f2), f3_), t_best);
    return tmin > tmax ? INFINITY : tmin;
}
SRT_HD void sort_pair(float& ta, uint32_t& ca, float& tb, uint32_t& cb)
{
    bool swap = tb < ta;
    float t = swap ? ta : tb; uint32_t c = swap ? ca : cb;
    ta = swap ? tb : ta; ca = swap ? cb : ca;
    tb = t; cb = c;
}

template <bool EXACT, bool COUNT, int STRIDE>
SRT_HD void trav_begin(const SceneView& sc, f3 o, float t_min, Trav& tr, uint32_t* stack_mem, Counters& cnt)
{
    tr.sp = 0;
    tr.pad = 0.0f;
    if (sc.has_bvh == 0u || (EXACT && sc.n_tiny != 0u)) {
        tr.best = closest_hit<EXACT, COUNT>(sc, o, tr.d, t_min, cnt);
        tr.cur = TRAV_NONE;
        return;
    }
    Hit best;
    best.t = INFINITY; best.prim = NO_PRIM; best.u = 0.0f; best.v = 0.0f; best.order = 0u;
    const f3 d = tr.d;
    const float a = lensq3(d);
    bool plain = true;
    if (!EXACT) plain = (fabsf(o.x) <= kTameRange) & (fabsf(o.y) <= kTameRange) & (fabsf(o.z) <= kTameRange) & (a <= 4.0f);  // false for NaN
    for (uint32_t i = 0; i < sc.list_count; i++) {
        test_prim<COUNT>(sc, sc.list_first + i, o, d, a, t_min, best, cnt);
        if (!EXACT) plain = plain & (best.t == best.t);
    }
    if (!plain) {
        tr.best = scan_all_surfaces<COUNT>(sc, o, d, a, t_min, cnt);
        tr.cur = TRAV_NONE;
        return;
    }
    tr.best = best;
    if (!EXACT) {
        // The float primitive tests accept rays that miss their surface: Sphere.hit's discriminant
        // half_b^2 - a (|oc|^2 - r^2) carries a rounding error of up to 20 * 2^-24 |oc|^2, so a ray that passes a sphere of
        // radius r at distance m still "hits" it when m^2 <= r^2 + 20 * 2^-24 |oc|^2.  The boxes of the tree are padded
        // for that with |oc| up to far_L0 (any origin inside the ball around the scene); a query that starts farther
        // away widens every box it tests by far_slope times its own distance bound instead (trav_run<.., PAD>).
        const float L = sqrt_rn(distsq3(o, sc.far_center)) + sc.far_radius;
        tr.pad = L > sc.far_L0 ? sc.far_slope * L : 0.0f;
    }
    if (EXACT) {
        const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        q4 r0 = table_word(sc.nodes, 0), r1 = table_word(sc.nodes, 1);
        if (COUNT) cnt.box_tests++;
        tr.cur = box_hit_reference(r0, r1, o, inv) ? fbits(r0.w) : (uint32_t)TRAV_NONE;
    } else {
        tr.cur = 0u;   // the wide root; its children's boxes are tested when it is visited
    }
}

// PAD: widen every box by tr.pad on all sides (queries that start far outside the scene, see trav_begin).
// trav_run_with: the caller keeps a = |d|^2 and inv = 1 / d across calls (a kernel that resumes a traversal many times).
template <bool EXACT, bool COUNT, int STRIDE, bool PAD = false, int NEAR = STACK_DEPTH>
SRT_HD void trav_run_with(const SceneView& sc, f3 o, float a, f3 inv, float t_min, Trav& tr, uint32_t* stack_mem, Counters& cnt, int rounds, int steps,
                          uint32_t* far_stack = nullptr)
{
    const f3 d = tr.d;
    LaneStack<STRIDE, NEAR> st;
    st.base = stack_mem; st.far = far_stack; st.sp = tr.sp;
    Hit best = tr.best;
    uint32_t cur = tr.cur;
    for (int r = 0; r < rounds && cur != TRAV_NONE; r++) {
        for (int s = 0; s < steps && !(cur & DESC_LEAF); s++) {
            if (EXACT) {
                // the reference tree: two adjacent children, left subtree first, no distance culling
                q4 l0 = table_word(sc.nodes, 2 * cur), l1 = table_word(sc.nodes, 2 * cur + 1);
                q4 h0 = table_word(sc.nodes, 2 * cur + 2), h1 = table_word(sc.nodes, 2 * cur + 3);
                if (COUNT) cnt.box_tests += 2;
                uint32_t dl = fbits(l0.w), dr = fbits(h0.w);
                bool hl = box_hit_reference(l0, l1, o, inv);
                bool hr = box_hit_reference(h0, h1, o, inv);
                if (hl) { if (hr) st.push(dr); cur = dl; }
                else if (hr) cur = dr;
                else cur = st.sp == 0 ? (uint32_t)TRAV_NONE : st.pop();
            } else {
                // a wide node: up to four quantised child boxes, nearest first, the others pushed far to near
                q4 w0 = table_word(sc.nodes, 4 * cur), w1 = table_word(sc.nodes, 4 * cur + 1), w2 = table_word(sc.nodes, 4 * cur + 2), w3 = table_word(sc.nodes, 4 * cur + 3);
                uint32_t meta = fbits(w0.w);
                uint32_t nk = meta >> 24;
                if (COUNT) cnt.box_tests += nk;
                f3 A = mk3(bitsf((meta & 255u) << 23) * inv.x, bitsf(((meta >> 8) & 255u) << 23) * inv.y, bitsf(((meta >> 16) & 255u) << 23) * inv.z);
                f3 B = mk3((w0.x - o.x) * inv.x, (w0.y - o.y) * inv.y, (w0.z - o.z) * inv.z);
                f3 Bn = B, Bf = B;
                if (PAD) {
                    // the plane the ray enters through moves towards the origin by pad, the one it leaves through away from
                    // it: pad * |1 / d| in ray distance (an infinite 1 / d makes NaNs, which fmaxf / fminf drop: never culled)
                    const f3 w = mk3(tr.pad * fabsf(inv.x), tr.pad * fabsf(inv.y), tr.pad * fabsf(inv.z));
                    Bn = sub3(B, w); Bf = add3(B, w);
                }
                // per axis: the planes the ray enters through / leaves through (a negative direction enters at hi)
                const bool bx = (fbits(d.x) >> 31) != 0u, by = (fbits(d.y) >> 31) != 0u, bz = (fbits(d.z) >> 31) != 0u;
                uint32_t lx = fbits(w1.x), ly = fbits(w1.y), lz = fbits(w1.z), hx = fbits(w1.w), hy = fbits(w2.x), hz = fbits(w2.y);
                uint32_t nx = bx ? hx : lx, fx = bx ? lx : hx, ny = by ? hy : ly, fy = by ? ly : hy, nz = bz ? hz : lz, fz = bz ? lz : hz;
                uint32_t c0 = fbits(w2.z), c1 = fbits(w2.w), c2 = fbits(w3.x), c3 = fbits(w3.y);
                // all four slots are evaluated (no branch on the child count); unused slots are masked afterwards
                float t0 = wide_child(nx, ny, nz, fx, fy, fz, 0, A, Bn, Bf, best.t);
                float t1 = wide_child(nx, ny, nz, fx, fy, fz, 8, A, Bn, Bf, best.t);
                float t2 = wide_child(nx, ny, nz, fx, fy, fz, 16, A, Bn, Bf, best.t);
                float t3 = wide_child(nx, ny, nz, fx, fy, fz, 24, A, Bn, Bf, best.t);
                t1 = nk > 1u ? t1 : INFINITY; t2 = nk > 2u ? t2 : INFINITY; t3 = nk > 3u ? t3 : INFINITY;
                sort_pair(t0, c0, t1, c1); sort_pair(t2, c2, t3, c3);
                sort_pair(t0, c0, t2, c2); sort_pair(t1, c1, t3, c3);
                sort_pair(t1, c1, t2, c2);
                if (t3 < INFINITY) st.push(c3);
                if (t2 < INFINITY) st.push(c2);
                if (t1 < INFINITY) st.push(c1);
                cur = t0 < INFINITY ? c0 : (st.sp == 0 ? (uint32_t)TRAV_NONE : st.pop());
            }
        }
        if ((cur & DESC_LEAF) && cur != TRAV_NONE) {
            uint32_t first = cur & 0x07FFFFFFu, count = ((cur >> 27) & 15u) + 1u;
            for (uint32_t i = 0; i < count; i++) test_prim<COUNT>(sc, first + i, o, d, a, t_min, best, cnt);
            cur = st.sp == 0 ? (uint32_t)TRAV_NONE : st.pop();
        }
    }
    tr.cur = cur; tr.sp = st.sp; tr.best = best;
}
template <bool EXACT, bool COUNT, int STRIDE, bool PAD = false>
SRT_HD void trav_run(const SceneView& sc, f3 o, float t_min, Trav& tr, uint32_t* stack_mem, Counters& cnt, int rounds, int steps)
{
    const f3 d = tr.d;
    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);  // CustomRay.dirFrac (RaytracingClasses.cs:16-18)
    trav_run_with<EXACT, COUNT, STRIDE, PAD>(sc, o, lensq3(d), inv, t_min, tr, stack_mem, cnt, rounds, steps);
}

// ------------------------------------------------------------------------------------------
// hitRecord reconstruction for the winning primitive
// ------------------------------------------------------------------------------------------
struct HitRecord { f3 point, normal; uint32_t material; };

SRT_HD f3 ray_at(f3 o, f3 d, float t) { return mk3(o.x + (t * d.x), o.y + (t * d.y), o.z + (t * d.z)); }  // CustomRay.getPos

SRT_HD uint32_t hit_material(const SceneView& sc, const Hit& h) { return fbits(sc.hit_prims[3 * h.prim + 2].y) >> 8; }

SRT_HD HitRecord resolve_hit(const SceneView& sc, const Hit& h, f3 o, f3 d)
{
    HitRecord r;
    q4 p2 = sc.hit_prims[3 * h.prim + 2];
    uint32_t info = fbits(p2.y);
    uint32_t type = info & 3u;
    r.material = info >> 8;
    r.point = ray_at(o, d, h.t);
    if (type == PRIM_SPHERE) {
        // RaytracingClasses.cs:482-483: Normalize((point - origin) / radius), flipped to face the ray
        q4 s = sc.hit_prims[3 * h.prim];
        float f = 1.0f / s.w;
        f3 n = normalize3(mk3((r.point.x - s.x) * f, (r.point.y - s.y) * f, (r.point.z - s.z) * f));
        bool front = dot3(d, n) < 0.0f;
        r.normal = front ? n : neg3(n);
    } else if (type == PRIM_TRI) {
        // RaytracingClasses.cs:629-643: never flipped towards the ray; frontFace is always true
        f3 n;
        if (info & INFO_BLEND) {
            uint32_t ni = fbits(p2.w);
            q4 a = table_word(sc.normals, 3 * ni), b = table_word(sc.normals, 3 * ni + 1), c = table_word(sc.normals, 3 * ni + 2);
            // Vector3.Barycentric: a + (b - a) u + (c - a) v per component
            n = mk3(a.x + (b.x - a.x) * h.u + (c.x - a.x) * h.v, a.y + (b.y - a.y) * h.u + (c.y - a.y) * h.v,
                    a.z + (b.z - a.z) * h.u + (c.z - a.z) * h.v);
        } else {
            q4 p0 = sc.hit_prims[3 * h.prim], p1 = sc.hit_prims[3 * h.prim + 1];
            n = cross3(mk3(p0.w, p1.x, p1.y), mk3(p1.z, p1.w, p2.x));
        }
        n = normalize3(n);
        r.normal = (info & INFO_FLIP) ? neg3(n) : n;
    } else {
        // RaytracingClasses.cs:531-532: outward normal (0,0,1), flipped to face the ray
        bool front = dot3(d, mk3(0.0f, 0.0f, 1.0f)) < 0.0f;
        r.normal = front ? mk3(0.0f, 0.0f, 1.0f) : mk3(-0.0f, -0.0f, -1.0f);
    }
    return r;
}

// ------------------------------------------------------------------------------------------
// Shading pieces of RayOperations.GetRayColor / PointLight.calculateLight
// ------------------------------------------------------------------------------------------
// RaytracingOperations.cs:88-116
SRT_HD int depth_heuristic(int depth, float smoothness)
{
    if (smoothness < 0.6f) depth -= (int)((float)depth / 4.0f);
    if (smoothness < 0.4f) depth -= (int)((float)depth / 2.0f);
    if (smoothness < 0.2f) depth -= (int)((float)depth / 2.0f);
    if (smoothness <= 0.05f) depth = 1;
    return depth < 1 ? 1 : (depth > 999999 ? 999999 : depth);
}
// RaytracingOperations.cs:192-193
SRT_HD f3 sky_color(float dir_y)
{
    float t = 0.5f * (dir_y + 1.0f);
    float s = 1.0f - t;
    return mk3(0.2f * t + 0.25f * s, 0.17f * t + 0.22f * s, 0.17f * t + 0.23f * s);
}
// CustomVectorMath.AngleBetween / 180, squared (MiscFunctions.cs:53-60, RaytracingOperations.cs:137-140)
SRT_HD float fresnel_term(f3 dir, f3 reflected)
{
#if defined(SRT_FAST_SHADING) && defined(__HIP_DEVICE_COMPILE__)
    // experiments only (not the arithmetic contract): what the shading would cost with single-precision atan2 and fused arithmetic
    {
        const float dtf = __builtin_fmaf(dir.z, reflected.z, __builtin_fmaf(dir.y, reflected.y, dir.x * reflected.x));
        const f3 cf = cross3(dir, reflected);
        const float lenf = __builtin_amdgcn_sqrtf(__builtin_fmaf(cf.z, cf.z, __builtin_fmaf(cf.y, cf.y, cf.x * cf.x)));
        const float ff = (3.14159274f - atan2f(lenf, dtf)) * (57.2957802f / 180.0f);
        return ff * ff;
    }
#endif
    float dt = dot3(dir, reflected);
    f3 c = cross3(dir, reflected);
    float len = sqrt_rn((c.x * c.x) + (c.y * c.y) + (c.z * c.z));
    float angle = (float)(3.14159265358979323846 - atan2((double)len, (double)dt));
    float deg = (float)((double)angle * 57.295779513082320876798154814105);
    float f = deg / 180.0f;
    return f * f;
}

struct Light { f3 position; float brightness; f3 color; };
SRT_HD Light load_light(const SceneView& sc, uint32_t i)
{
    q4 a = table_word(sc.lights, 2 * i), b = table_word(sc.lights, 2 * i + 1);
    Light l;
    l.position = mk3(a.x, a.y, a.z); l.brightness = a.w; l.color = mk3(b.x, b.y, b.z);
    return l;
}

// Direction of a light's shadow ray from hit point P (RaytracingClasses.cs:105,174):
// Normalize(position - P) + ranged(0.95), normalised again by the CustomRay constructor.
SRT_HD f3 shadow_direction(const Light& L, f3 P, uint32_t word, const float* unit)
{
    f3 to_light = normalize3(sub3(L.position, P));
    f3 j = ranged_vector_from_word(word, 0.95f, unit);
    return normalize3(add3(to_light, j));
}

// PointLight.calculateLight, GGX branch (RaytracingClasses.cs:96-188), evaluated after its
// shadow ray came back.  `former` is formerLightLevel; shadow_hit/t_s/P_s/metal_s describe
// what the shadow ray's closest-hit query left in `record` (the occluder), metal_s being the
// primary material's metalness when the shadow ray hit nothing.
#if defined(SRT_FAST_SHADING) && defined(__HIP_DEVICE_COMPILE__)
// experiments only (not the arithmetic contract): the same formulas with contraction allowed, v_rcp_f32 / v_rsq_f32 / v_sqrt_f32
// instead of the correctly rounded operations and the fifth power in single precision.  The occlusion test keeps its exact form.
SRT_HD f3 fast_normalize3(f3 a) { const float f = __builtin_amdgcn_rsqf(__builtin_fmaf(a.z, a.z, __builtin_fmaf(a.y, a.y, a.x * a.x))); return mk3(a.x * f, a.y * f, a.z * f); }
SRT_HD f3 point_light_fast(const Light& L, f3 P, f3 N, f3 dir, const Material& m, f3 former, bool shadow_hit, float t_s, f3 P_s, float metal_s)
{
#pragma clang fp contract(fast)
    const float roughness = 1.0f - m.smoothness, r2 = roughness * roughness;
    const float att = clampf(__builtin_amdgcn_rcpf(distsq3(P, L.position)), 0.0f, 1.0f);
    const f3 to_light = fast_normalize3(sub3(L.position, P)), view = fast_normalize3(neg3(dir)), half = fast_normalize3(add3(to_light, view));
    const float NdotL = clampf(dot3(N, to_light), 0.0f, 1.0f), NdotV = clampf(dot3(N, view), 0.0f, 1.0f);
    const float LdotH = clampf(dot3(to_light, half), 0.0f, 1.0f), NdotH = clampf(dot3(N, half), 0.0f, 1.0f);
    const float NdotL2 = NdotL * NdotL, NdotV2 = NdotV * NdotV;
    const float smithL = 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_sqrtf(1.0f + r2 * (1.0f - NdotL2) * __builtin_amdgcn_rcpf(NdotL2)));
    const float smithV = 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_sqrtf(1.0f + r2 * (1.0f - NdotV2) * __builtin_amdgcn_rcpf(NdotV2)));
    const float geo = smithL * smithV;
    const float pi = 3.14159274f;
    const float denom = NdotH * NdotH * (r2 - 1.0f) + 1.0f;
    const float D = r2 * __builtin_amdgcn_rcpf(pi * denom * denom);
    const float x = 1.0f - LdotH, x2 = x * x;
    const float F = 0.5f + 0.5f * (x2 * x2 * x);
    const float k = r2 * 0.5f;
    const float vis = __builtin_amdgcn_rcpf(NdotL * (1.0f - k) + k) * __builtin_amdgcn_rcpf(NdotV * (1.0f - k) + k);
    float ggx = NdotL * D * F * vis * (L.brightness * 0.5f);
    const f3 light_albedo = add3(m.albedo, L.color);
    f3 level = add3(former, scale3(light_albedo, (1.0f - m.metalness) * (NdotL * NdotV) * geo * L.brightness * att));
    if (shadow_hit) {
#pragma clang fp contract(off)
        f3 d2 = normalize3(normalize3(sub3(L.position, P)));
        f3 along = sub3(ray_at(P_s, d2, t_s), P_s);
        if (lensq3(along) < lensq3(sub3(L.position, P_s))) { level = former; ggx = 0.0f; }
    }
    const f3 one = mk3(1.0f + L.color.x, 1.0f + L.color.y, 1.0f + L.color.z);
    const f3 lerp = mk3(one.x + (light_albedo.x - one.x) * metal_s, one.y + (light_albedo.y - one.y) * metal_s, one.z + (light_albedo.z - one.z) * metal_s);
    return add3(level, scale3(lerp, (1.0f / pi) * ggx * att));
}
#endif
SRT_HD f3 point_light(const Light& L, f3 P, f3 N, f3 dir, const Material& m, f3 former, bool shadow_hit, float t_s, f3 P_s, float metal_s)
{
#if defined(SRT_FAST_SHADING) && defined(__HIP_DEVICE_COMPILE__)
    return point_light_fast(L, P, N, dir, m, former, shadow_hit, t_s, P_s, metal_s);
#endif
    float roughness = 1.0f - m.smoothness;
    float r2 = roughness * roughness;
    float att = clampf(1.0f / distsq3(P, L.position), 0.0f, 1.0f);
    f3 to_light = normalize3(sub3(L.position, P));
    f3 view = normalize3(neg3(dir));
    f3 half = normalize3(add3(to_light, view));
    float NdotL = clampf(dot3(N, to_light), 0.0f, 1.0f);
    float NdotV = clampf(dot3(N, view), 0.0f, 1.0f);
    float LdotH = clampf(dot3(to_light, half), 0.0f, 1.0f);
    float NdotH = clampf(dot3(N, half), 0.0f, 1.0f);

    float NdotL2 = NdotL * NdotL, NdotV2 = NdotV * NdotV;
    float smithL = smith_term(r2 * (1.0f - NdotL2) / NdotL2);
    float smithV = smith_term(r2 * (1.0f - NdotV2) / NdotV2);
    float geo = smithL * smithV;
    const float pi = 3.14159274f;  // (float)Math.PI
    float denom = NdotH * NdotH * (r2 - 1.0f) + 1.0f;
    float D = r2 / (pi * denom * denom);
    double x = (double)(1.0f - LdotH);
    double x2 = x * x;
    float p5 = (float)(x2 * x2 * x);  // Math.Pow(1 - LdotH, 5)
    float F = 0.5f + (1.0f - 0.5f) * p5;
    float k = r2 / 2.0f;
    float vis = (1.0f / (NdotL * (1.0f - k) + k)) * (1.0f / (NdotV * (1.0f - k) + k));
    float ggx = NdotL * D * F * vis;
    ggx *= (L.brightness / 2.0f);

    f3 light_albedo = add3(m.albedo, L.color);
    f3 diffuse = scale3(scale3(scale3(scale3(scale3(light_albedo, 1.0f - m.metalness), NdotL * NdotV), geo), L.brightness), att);
    f3 level = add3(former, diffuse);

    if (shadow_hit) {
        // :175 -- compares squared lengths: |(P_s + t_s * Normalize(to_light)) - P_s|^2 < |position - P_s|^2
        f3 d2 = normalize3(to_light);
        f3 along = sub3(ray_at(P_s, d2, t_s), P_s);
        if (lensq3(along) < lensq3(sub3(L.position, P_s))) {
            level = former;
            ggx = 0.0f;
        }
    }
    // :185 -- Lerp((1,1,1) + lightColor, lightAlbedo, record.hitMat.metalness) / pi * ggx * attenuation
    f3 one = mk3(1.0f + L.color.x, 1.0f + L.color.y, 1.0f + L.color.z);
    f3 lerp = mk3(one.x + (light_albedo.x - one.x) * metal_s, one.y + (light_albedo.y - one.y) * metal_s, one.z + (light_albedo.z - one.z) * metal_s);
    const float inv_pi = 1.0f / pi;  // Vector3 / float multiplies by the reciprocal
    f3 spec = scale3(scale3(scale3(lerp, inv_pi), ggx), att);
    return add3(level, spec);
}

// Whether the shadow ray of light L at the hit (P, N) of a ray with direction `dir` can influence what
// PointLight.calculateLight returns.  It cannot when the light is behind the surface and everything involved is of
// ordinary size: with NdotL = 0 (RaytracingClasses.cs:113) SmithL = 2 / (1 + sqrt(1 + r2 / 0)) = 0, so the diffuse term
// (:160-161) is lightAlbedo (1 - metalness) * (0 * NdotV) * 0 * brightness * attenuation = 0 and lightLevel is
// formerLightLevel whether or not line :178 resets it; and ggx = 0 * D * F * vis (:155) is 0 whether or not line :179
// resets it, so the specular term (:185) is Lerp(..., metalness of whatever the shadow ray hit) / pi * 0 * attenuation
// = 0 for any occluder.  That needs every other factor to be finite, which the conditions below guarantee:
//   r2 in [2^-20, 2]        D = r2 / (pi denom^2) with denom = NdotH^2 (r2 - 1) + 1 >= 2^-24 (r2 - 1 > -1 in float), and
//                           vis = 1 / k * 1 / (NdotV (1 - k) + k) with k = r2 / 2 in (0, 1], both factors <= 2^21;
//   |to_light + view|^2 in [2^-60, 8]   the half vector is finite (and so are to_light and view), so with a finite N the
//                           clamped dot products NdotV, NdotH, LdotH are numbers in [0, 1] and F is in [0.5, 1];
//   P and N of ordinary size (the attenuation is a number in [0, 1]); albedo, metalness, light colour, brightness and
//   position of ordinary size for every material and light of the scene (sc.inert_lights, checked by the host).
// formerLightLevel is a sum that started at +0, so it is never -0 and adding a zero of either sign returns it unchanged:
// the call returns formerLightLevel bit for bit, with or without the shadow ray.  NdotL is computed exactly as
// point_light() computes it.
SRT_HD bool shadow_ray_matters(const SceneView& sc, const Light& L, f3 P, f3 N, f3 dir, float smoothness)
{
    if (sc.inert_lights == 0u) return true;
    const float big = 0x1p40f;
    const float roughness = 1.0f - smoothness;
    const float r2 = roughness * roughness;
    const f3 to_light = normalize3(sub3(L.position, P));
    const f3 view = normalize3(neg3(dir));
    const float NdotL = clampf(dot3(N, to_light), 0.0f, 1.0f);
    const float s2 = lensq3(add3(to_light, view));
    const bool tame = (fabsf(P.x) <= big) & (fabsf(P.y) <= big) & (fabsf(P.z) <= big) & (fabsf(N.x) <= big) & (fabsf(N.y) <= big) & (fabsf(N.z) <= big);
    const bool inert = (NdotL == 0.0f) & (r2 >= 0x1p-20f) & (r2 <= 2.0f) & (s2 >= 0x1p-60f) & (s2 <= 8.0f) & tame;   // all false for NaN
    return !inert;
}

// ------------------------------------------------------------------------------------------
// Render arguments and the work-item <-> pixel mapping
// ------------------------------------------------------------------------------------------
struct Camera { f3 position, lower_left, horizontal, vertical; };

struct RenderArgs {
    Camera cam;
    int32_t width, height, samples, max_depth;
    float color_clamp;
    uint32_t seed;
    int32_t mode, layout;
    int32_t tile_w, tile_h, rank, world_size;
    int32_t tiles_x, tiles_y;       // tile grid of the full image
    uint32_t n_work;                // pixels (incl. padding of edge tiles) owned by this rank; a multiple of 64
    // one pass of the trace kernel covers samples [sample_first, sample_first + sample_count) of every pixel
    uint32_t sample_first, sample_count;
    uint32_t n_items;               // n_work * sample_count work items (one per pixel and sample)
    // The first `coherent` samples of the pass are enumerated block by block: 64 consecutive items are one 8x8 pixel
    // block at one sample index (neighbouring rays take the same branches and share nodes and cache lines).  The
    // remaining samples come one after the other, and inside a sample item i is position i % 64 of block
    // (i / 64 + (i % 64) * scatter) mod (n_work / 64): 64 consecutive items come from 64 blocks spread over the
    // picture, so every draw of items costs about the same and no wavefront ends the launch with a pile of deep trees.
    uint32_t coherent, scatter;
};

// Item order of one pass (after sample_count is set): the last `scatter_samples` samples in scattered order.
inline void set_item_order(RenderArgs& ra, uint32_t scatter_samples)
{
    ra.coherent = ra.sample_count;
    ra.scatter = 0u;
    if (scatter_samples != 0u && ra.n_work >= 128u) {
        ra.coherent = ra.sample_count > scatter_samples ? ra.sample_count - scatter_samples : 0u;
        ra.scatter = ((ra.n_work >> 6) / 64u) | 1u;
    }
}

// A parent waiting for its children (one per pending level of the recursion).
struct Frame {
    f3 acc;        // emission + weighted children so far
    f3 light;      // lightColor, added last
    f3 point;      // record.point
    f3 refl;       // Normalize(reflectedDirection): direction of the metal and reflect children
    float w_refl;  // (1 - metalness) * (fresnel * smoothness^2)
    uint32_t meta; // material | child depth << 24
};

enum : uint32_t { MODE_COLOR = 0u, MODE_NORMAL = 1u, MODE_DEPTH = 2u, MODE_LIGHTS = 3u, MODE_DIFFUSE = 4u, MODE_REFLECTION = 5u };

// The mix-buffer technique's step for one pixel (Game1.cs:215-218): new Color(a.ToVector3() / 2 + b.ToVector3() / 2), with
// Color.ToVector3() = (R / 255f, G / 255f, B / 255f) and Vector3 / 2 a multiplication by 0.5.
SRT_HD uint32_t mix_colors(uint32_t a, uint32_t b)
{
    const f3 ta = mk3((float)(a & 255u) / 255.0f, (float)((a >> 8) & 255u) / 255.0f, (float)((a >> 16) & 255u) / 255.0f);
    const f3 tb = mk3((float)(b & 255u) / 255.0f, (float)((b >> 8) & 255u) / 255.0f, (float)((b >> 16) & 255u) / 255.0f);
    const f3 m = add3(scale3(ta, 0.5f), scale3(tb, 0.5f));
    return pack_color(m.x, m.y, m.z);
}

// Tiles are numbered row by row, and inside row ty the numbering starts ty columns in: tile t sits in row
// ty = t / tiles_x at column (t % tiles_x + ty) % tiles_x.  With tile t going to rank t % world_size the ranks then lie
// on diagonals of the tile grid even when tiles_x is a multiple of world_size (3840 and 7680 pixels are 120 and 240
// tiles of 32: 2, 4 and 8 ranks would otherwise own the same columns in every row).
SRT_HD void tile_position(const RenderArgs& ra, uint32_t t, uint32_t& tx, uint32_t& ty)
{
    ty = t / (uint32_t)ra.tiles_x;
    tx = (t - ty * (uint32_t)ra.tiles_x + ty) % (uint32_t)ra.tiles_x;
}
SRT_HD uint32_t tile_number(const RenderArgs& ra, uint32_t tx, uint32_t ty)
{
    const uint32_t n = (uint32_t)ra.tiles_x;
    return ty * n + (tx + n - ty % n) % n;
}

// Maps a pixel slot of this rank to its pixel.  Slots enumerate this rank's tiles (tile t belongs to
// rank t % world_size), and inside a tile 8x8 blocks, so that 64 consecutive slots are an 8x8
// pixel block.  Returns false for padding pixels of edge tiles.
SRT_HD bool work_to_pixel(const RenderArgs& ra, uint32_t w, int32_t& x, int32_t& y, uint32_t& out_index)
{
    uint32_t tile_px = (uint32_t)(ra.tile_w * ra.tile_h);
    uint32_t j = w / tile_px, rem = w - j * tile_px;
    uint32_t t = j * (uint32_t)ra.world_size + (uint32_t)ra.rank;
    uint32_t tx, ty;
    tile_position(ra, t, tx, ty);
    uint32_t sbw = (uint32_t)ra.tile_w >> 3;
    uint32_t sb = rem >> 6, k = rem & 63u;
    uint32_t sby = sb / sbw, sbx = sb - sby * sbw;
    uint32_t lx = sbx * 8u + (k & 7u), ly = sby * 8u + (k >> 3);
    x = (int32_t)(tx * (uint32_t)ra.tile_w + lx);
    y = (int32_t)(ty * (uint32_t)ra.tile_h + ly);
    out_index = ra.layout == 0 ? (uint32_t)x + (uint32_t)y * (uint32_t)ra.width : j * tile_px + ly * (uint32_t)ra.tile_w + lx;
    return x < ra.width && y < ra.height && ty < (uint32_t)ra.tiles_y;
}

// Where pixel i = x + y * width of the frame sits in the gathered tile buffers of a partitioned render (rank r's buffer
// starts at r * stride): its tile, number t, belongs to rank t % world_size, which holds it as its (t / world_size)-th
// tile, row-major inside the tile.  The inverse of work_to_pixel's SRT_LAYOUT_TILES output index.
SRT_HD size_t gathered_index(const RenderArgs& ra, uint32_t i, size_t stride)
{
    uint32_t y = i / (uint32_t)ra.width, x = i - y * (uint32_t)ra.width;
    uint32_t tx = x / (uint32_t)ra.tile_w, ty = y / (uint32_t)ra.tile_h;
    uint32_t lx = x - tx * (uint32_t)ra.tile_w, ly = y - ty * (uint32_t)ra.tile_h;
    uint32_t t = tile_number(ra, tx, ty);
    uint32_t r = t % (uint32_t)ra.world_size, j = t / (uint32_t)ra.world_size;
    return (size_t)r * stride + (size_t)j * (uint32_t)(ra.tile_w * ra.tile_h) + ly * (uint32_t)ra.tile_w + lx;
}

}  // namespace srt
#endif  // SRT_CORE_H
