/*
 * srt_oracle.cpp -- CPU restatement of helpvisa/simple_raytracer's per-pixel trace loop.
 *
 * TEST INFRASTRUCTURE ONLY: parity checker and CPU baseline.  The product library never
 * includes, links or calls this file.
 *
 * PARITY UNPINNED: the reference is C# on MonoGame with no tests or golden vectors and no
 * .NET toolchain is available, so this restatement was written by reading the C# and is
 * pinned only by hand-derived known-answer tests.  MonoGame / .NET arithmetic that is not
 * part of the reference tree (Vector3 helpers, MathHelper, Color, Math.Min/Max/Clamp,
 * float->int casts) is restated from the published behaviour of MonoGame 3.8.0.1641 and
 * .NET Core 3.1 (simpleRaytracer.csproj:4,29-31).
 *
 * It keeps the reference's structure on purpose -- heap objects, virtual dispatch, a
 * recursive pointer-tree BVH built by the reference's own builder, per-light
 * re-intersection, recursion for the ray tree -- because it doubles as the CPU baseline.
 *
 * The one deliberate change: System.Random (entropy seeded per thread, Game1.cs:44,378, so
 * the reference is not reproducible) is replaced by a counter-based generator keyed by
 * (seed, pixel, sample, ray-tree node, draw slot).  Next(lo,hi) keeps its integer-quantised
 * semantics.  The GPU kernels use the same generator, which makes images comparable bit
 * for bit.
 *
 * Build: g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared -pthread (see Makefile).
 * All float arithmetic is plain IEEE binary32 in the operation order of the C#.
 */
#include "srt_oracle.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_error;

// ---------------------------------------------------------------------------------------
// MonoGame Vector3 (Microsoft.Xna.Framework.Vector3, 3.8.0) -- operation order as published.
// ---------------------------------------------------------------------------------------
struct Vec { float X, Y, Z; };

inline Vec V(float x, float y, float z) { Vec r; r.X = x; r.Y = y; r.Z = z; return r; }
inline Vec operator+(Vec a, Vec b) { return V(a.X + b.X, a.Y + b.Y, a.Z + b.Z); }
inline Vec operator-(Vec a, Vec b) { return V(a.X - b.X, a.Y - b.Y, a.Z - b.Z); }
inline Vec operator-(Vec a) { return V(-a.X, -a.Y, -a.Z); }
inline Vec operator*(Vec a, float s) { return V(a.X * s, a.Y * s, a.Z * s); }
inline Vec operator*(float s, Vec a) { return V(a.X * s, a.Y * s, a.Z * s); }
inline Vec operator*(Vec a, Vec b) { return V(a.X * b.X, a.Y * b.Y, a.Z * b.Z); }
// Vector3 / float multiplies by the reciprocal.
inline Vec operator/(Vec a, float s) { float factor = 1.0f / s; return V(a.X * factor, a.Y * factor, a.Z * factor); }

// (float)Math.Sqrt(x): double sqrt of a float, rounded back -- equals correctly rounded sqrtf.
inline float SqrtF(float v) { return (float)std::sqrt((double)v); }

inline float Dot(Vec a, Vec b) { return a.X * b.X + a.Y * b.Y + a.Z * b.Z; }
inline Vec Cross(Vec a, Vec b)
{
    float x = a.Y * b.Z - b.Y * a.Z;
    float y = -(a.X * b.Z - b.X * a.Z);
    float z = a.X * b.Y - b.X * a.Y;
    return V(x, y, z);
}
inline float LengthOf(Vec a) { return SqrtF((a.X * a.X) + (a.Y * a.Y) + (a.Z * a.Z)); }
inline Vec Normalize(Vec a)
{
    float factor = SqrtF((a.X * a.X) + (a.Y * a.Y) + (a.Z * a.Z));
    factor = 1.0f / factor;
    return V(a.X * factor, a.Y * factor, a.Z * factor);
}
inline float DistanceSquared(Vec a, Vec b)
{
    return (a.X - b.X) * (a.X - b.X) + (a.Y - b.Y) * (a.Y - b.Y) + (a.Z - b.Z) * (a.Z - b.Z);
}
inline float Distance(Vec a, Vec b) { return SqrtF(DistanceSquared(a, b)); }
inline Vec Reflect(Vec v, Vec n)
{
    float d = ((v.X * n.X) + (v.Y * n.Y)) + (v.Z * n.Z);
    return V(v.X - (2.0f * n.X) * d, v.Y - (2.0f * n.Y) * d, v.Z - (2.0f * n.Z) * d);
}
inline float LerpF(float a, float b, float t) { return a + (b - a) * t; }
inline Vec Lerp(Vec a, Vec b, float t) { return V(LerpF(a.X, b.X, t), LerpF(a.Y, b.Y, t), LerpF(a.Z, b.Z, t)); }
inline float BaryF(float v1, float v2, float v3, float a1, float a2) { return v1 + (v2 - v1) * a1 + (v3 - v1) * a2; }
inline Vec Barycentric(Vec a, Vec b, Vec c, float u, float v)
{
    return V(BaryF(a.X, b.X, c.X, u, v), BaryF(a.Y, b.Y, c.Y, u, v), BaryF(a.Z, b.Z, c.Z, u, v));
}
inline float ToRadians(float deg) { return (float)(deg * 0.017453292519943295769236907684886); }
inline float ToDegrees(float rad) { return (float)(rad * 57.295779513082320876798154814105); }

// .NET Core 3.1 System.Math for float: Min/Max propagate NaN, Clamp passes NaN through.
inline bool IsNeg(float v) { return std::signbit(v); }
inline float MaxF(float a, float b)
{
    if ((a > b) || (a != a)) return a;
    if (a == b) return IsNeg(a) ? b : a;
    return b;
}
inline float MinF(float a, float b)
{
    if ((a < b) || (a != a)) return a;
    if (a == b) return IsNeg(a) ? a : b;
    return b;
}
inline float ClampF(float v, float lo, float hi)
{
    if (v < lo) return lo;
    if (v > hi) return hi;
    return v;
}
inline int ClampI(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// (int)floatValue on x64 RyuJIT (cvttss2si): truncation; NaN / out of range -> int.MinValue.
inline int32_t FloatToInt(float v)
{
    if (!(v >= -2147483648.0f && v < 2147483648.0f)) return std::numeric_limits<int32_t>::min();
    return (int32_t)v;
}
// new Color(float r, float g, float b): (int)(c * 255) per channel, clamped to a byte, alpha 255.
inline uint32_t PackColor(float r, float g, float b)
{
    int R = ClampI(FloatToInt(r * 255), 0, 255);
    int G = ClampI(FloatToInt(g * 255), 0, 255);
    int B = ClampI(FloatToInt(b * 255), 0, 255);
    return (uint32_t)R | ((uint32_t)G << 8) | ((uint32_t)B << 16) | 0xFF000000u;
}

// ---------------------------------------------------------------------------------------
// Counter-based RNG shared with the GPU kernels (replaces System.Random).
//   call(key, c):  four words = pcg4d(pixel, sample, node, c + 0x9E3779B9 * seed)
//   digit(w, B):   d = floor(w * B / 2^32), w <- (w * B) mod 2^32 -- successive base-B digits of the
//                  fraction w / 2^32, independent and uniform on [0, B).  Random.Next(lo, lo + B)
//                  (= (int)(Sample() * B) + lo with Sample() uniform in [0,1)) becomes lo + digit.
// Words inside one ray-tree node: call 0 word 0 = pixel jitter (node 0: two base-100 digits, u
// first) or the reflection roughness vector (three base-200 digits, X first); word (1+l)&3 of
// call (1+l)>>2 = shadow jitter of light l; word i&3 of call 0x10000 + (i>>2) = i-th try of the
// unit-sphere rejection loop.
// Node ids: primary ray = 1; child k (0 diffuse, 1 metal, 2 reflect) of node n = 4n + k + 1.
// ---------------------------------------------------------------------------------------
struct RngKey { uint32_t seed, pixel, sample, node; };

enum : uint32_t { CALL_AUX = 0, CALL_SPHERE0 = 0x10000u };
const int kMaxRejectionTries = 64;  // probability of reaching it: (1 - pi/6)^64 ~ 2e-21

inline void RngWords(const RngKey& k, uint32_t slot, uint32_t w[4])
{
    uint32_t x = k.pixel, y = k.sample, z = k.node, t = slot + 0x9E3779B9u * k.seed;
    x = x * 1664525u + 1013904223u;
    y = y * 1664525u + 1013904223u;
    z = z * 1664525u + 1013904223u;
    t = t * 1664525u + 1013904223u;
    x += y * t; y += z * x; z += x * y; t += y * z;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; t ^= t >> 16;
    x += y * t; y += z * x; z += x * y; t += y * z;
    w[0] = x; w[1] = y; w[2] = z; w[3] = t;
}
// random.Next(lo, lo + range): the next base-`range` digit of the word
inline int RngNext(uint32_t& word, int lo, int range)
{
    uint64_t p = (uint64_t)word * (uint64_t)(uint32_t)range;
    word = (uint32_t)p;
    return lo + (int)(p >> 32);
}
// word j of the node's auxiliary calls (j = 0: reflection roughness / jitter, j = 1 + l: light l)
inline uint32_t AuxWord(const RngKey& k, uint32_t j)
{
    uint32_t w[4];
    RngWords(k, CALL_AUX + (j >> 2), w);
    return w[j & 3u];
}

// MiscFunctions.cs:12-15  ReturnRandomVector: three Next(-100,100)/100 draws, X then Y then Z.
inline Vec ReturnRandomVector(uint32_t word)
{
    float x = (float)RngNext(word, -100, 200) / 100;
    float y = (float)RngNext(word, -100, 200) / 100;
    float z = (float)RngNext(word, -100, 200) / 100;
    return V(x, y, z);
}
// MiscFunctions.cs:18-23
inline Vec ReturnRandomRangedVector(uint32_t word, float multiplier)
{
    float factor = 1 - ClampF(multiplier, 0, 1);
    float x = (float)RngNext(word, -100, 200) / 100;
    float y = (float)RngNext(word, -100, 200) / 100;
    float z = (float)RngNext(word, -100, 200) / 100;
    return V(x * factor, y * factor, z * factor);
}
// MiscFunctions.cs:26-34 (rejection loop; bounded here, see kMaxRejectionTries)
inline Vec ReturnRandomUnitSphereVector(const RngKey& k)
{
    Vec p = V(0, 0, 0);
    for (int i = 0; i < kMaxRejectionTries; i++) {
        uint32_t w[4];
        RngWords(k, CALL_SPHERE0 + (uint32_t)(i >> 2), w);
        p = ReturnRandomVector(w[i & 3]);
        if (DistanceSquared(p, V(0, 0, 0)) >= 1) continue;
        return p;
    }
    return p;
}
// MiscFunctions.cs:36-39
inline Vec ReturnRandomNormalizedUnitSphereVector(const RngKey& k) { return Normalize(ReturnRandomUnitSphereVector(k)); }
// MiscFunctions.cs:41-48
inline Vec ReturnHemisphereRandomVector(const RngKey& k, Vec normal)
{
    Vec unitSphere = ReturnRandomUnitSphereVector(k);
    if (Dot(unitSphere, normal) > 0) return unitSphere;
    return -unitSphere;
}
// MiscFunctions.cs:53-60: 180 degrees minus the angle, atan2 in double.
inline float AngleBetween(Vec a, Vec b)
{
    float dot = Dot(a, b);
    Vec cross = Cross(a, b);
    float angle = (float)(3.14159265358979323846 - std::atan2((double)LengthOf(cross), (double)dot));
    return ToDegrees(angle);
}
// MiscFunctions.cs:68-71: despite its name, the squared length.
inline float Magnitude(Vec a) { return (a.X * a.X + a.Y * a.Y + a.Z * a.Z); }

// ---------------------------------------------------------------------------------------
// Scene classes (RaytracingClasses.cs)
// ---------------------------------------------------------------------------------------
struct Counters {
    uint64_t rays_path = 0, rays_shadow = 0, rays_redundant = 0, box_tests = 0;
    uint64_t sphere_tests = 0, tri_tests = 0, rect_tests = 0, hit_nodes = 0;
    // the part of the four *_tests above that repeats a test whose answer is already known: everything inside a
    // per-light re-intersection (RaytracingClasses.cs:211) and the second test of a single-surface leaf (:362-365)
    uint64_t box_repeats = 0, sphere_repeats = 0, tri_repeats = 0, rect_repeats = 0;
    // shadow rays that cannot change what PointLight.calculateLight returns (see there), and the tests of their
    // traversals that are not repeats already
    uint64_t shadow_inert = 0, box_inert = 0, sphere_inert = 0, tri_inert = 0, rect_inert = 0;
    void add(const Counters& o)
    {
        rays_path += o.rays_path; rays_shadow += o.rays_shadow; rays_redundant += o.rays_redundant;
        box_tests += o.box_tests; sphere_tests += o.sphere_tests; tri_tests += o.tri_tests;
        rect_tests += o.rect_tests; hit_nodes += o.hit_nodes;
        box_repeats += o.box_repeats; sphere_repeats += o.sphere_repeats; tri_repeats += o.tri_repeats; rect_repeats += o.rect_repeats;
        shadow_inert += o.shadow_inert; box_inert += o.box_inert; sphere_inert += o.sphere_inert; tri_inert += o.tri_inert; rect_inert += o.rect_inert;
    }
};
thread_local Counters* t_cnt = nullptr;
thread_local int t_repeat = 0;   // > 0 while the traversal only repeats tests (see Counters)
thread_local int t_inert = 0;    // > 0 during the traversal of a shadow ray that cannot matter
thread_local bool t_ordinary = false;   // every material and light value of the scene is of ordinary size (see PointLight)
Counters g_sink;
inline Counters& C() { return t_cnt ? *t_cnt : g_sink; }

// RaytracingClasses.cs:9-29
struct CustomRay {
    Vec origin, direction, dirFrac;
    CustomRay(Vec o, Vec d)
    {
        origin = o;
        direction = Normalize(d);
        dirFrac.X = 1.0f / direction.X;
        dirFrac.Y = 1.0f / direction.Y;
        dirFrac.Z = 1.0f / direction.Z;
    }
    Vec getPos(float t) const { return origin + (t * direction); }
};

// RaytracingClasses.cs:221-262
struct Material {
    Vec albedo; float metalness; float smoothness; Vec emission;
};

// RaytracingClasses.cs:797-810
struct hitRecord {
    Vec point = V(0, 0, 0), normal = V(0, 0, 0);
    float t = 0;
    bool frontFace = false;
    const Material* hitMat = nullptr;
    void setFaceNormal(const CustomRay& ray, Vec outwardNormal)
    {
        frontFace = Dot(ray.direction, outwardNormal) < 0;
        normal = frontFace ? outwardNormal : -outwardNormal;
    }
};

// RaytracingClasses.cs:694-794 (constructor argument order is (max, min); "approach 3" slab test)
struct AABB {
    Vec max, min;
    AABB(Vec inMax, Vec inMin) : max(inMax), min(inMin) {}
    bool hit(const CustomRay& ray) const
    {
        C().box_tests++;
        if (t_repeat) C().box_repeats++;
        else if (t_inert) C().box_inert++;
        float t1 = (min.X - ray.origin.X) * ray.dirFrac.X;
        float t2 = (max.X - ray.origin.X) * ray.dirFrac.X;
        float t3 = (min.Y - ray.origin.Y) * ray.dirFrac.Y;
        float t4 = (max.Y - ray.origin.Y) * ray.dirFrac.Y;
        float t5 = (min.Z - ray.origin.Z) * ray.dirFrac.Z;
        float t6 = (max.Z - ray.origin.Z) * ray.dirFrac.Z;
        // the caller's t_min / t_max are overwritten, i.e. ignored (RaytracingClasses.cs:780-781)
        float t_min = MaxF(MaxF(MinF(t1, t2), MinF(t3, t4)), MinF(t5, t6));
        float t_max = MinF(MinF(MaxF(t1, t2), MaxF(t3, t4)), MaxF(t5, t6));
        if (t_max < 0) return false;
        if (t_min > t_max) return false;
        return true;
    }
};

// MiscFunctions.cs:76-86
inline AABB surroundingBox(const AABB& box0, const AABB& box1)
{
    Vec small_ = V(MinF(box0.min.X, box1.min.X), MinF(box0.min.Y, box1.min.Y), MinF(box0.min.Z, box1.min.Z));
    Vec big = V(MaxF(box0.max.X, box1.max.X), MaxF(box0.max.Y, box1.max.Y), MaxF(box0.max.Z, box1.max.Z));
    return AABB(big, small_);
}

// RaytracingClasses.cs:265-279
struct Surface {
    virtual ~Surface() {}
    virtual bool hit(const CustomRay&, float, float, hitRecord&) const { return false; }
    virtual bool boundingBox() { return false; }
    std::unique_ptr<AABB> bounds;  // null until boundingBox() succeeds
    int primId = -1;               // bookkeeping for tests only (not in the reference)
};

struct NullBounds {};  // stands for the NullReferenceException the reference would throw

// MiscFunctions.cs:88-105: the "centroid" is min + (min - max/2)
inline bool boxCompare(const Surface* a, const Surface* b, int axis)
{
    if (!a->bounds || !b->bounds) throw NullBounds();
    Vec aCent = a->bounds->min + (a->bounds->min - a->bounds->max / 2);
    Vec bCent = b->bounds->min + (b->bounds->min - b->bounds->max / 2);
    switch (axis) {
        case 0: return aCent.X < bCent.X;
        case 1: return aCent.Y < bCent.Y;
        case 2: return aCent.Z < bCent.Z;
        default: return false;
    }
}

// RaytracingClasses.cs:449-500
struct Sphere : Surface {
    Vec origin; float radius; const Material* material;
    Sphere(Vec o, float r, const Material* m) : origin(o), radius(r), material(m) {}
    bool hit(const CustomRay& ray, float t_min, float t_max, hitRecord& record) const override
    {
        C().sphere_tests++;
        if (t_repeat) C().sphere_repeats++;
        else if (t_inert) C().sphere_inert++;
        Vec distance = ray.origin - origin;
        float a = DistanceSquared(ray.direction, V(0, 0, 0));
        float half_b = Dot(distance, ray.direction);
        float c = DistanceSquared(distance, V(0, 0, 0)) - radius * radius;
        float discriminant = half_b * half_b - a * c;
        if (discriminant < 0) return false;
        float sqrtDisc = SqrtF(discriminant);
        float root = (-half_b - sqrtDisc) / a;
        if (root < t_min || t_max < root) {
            root = (-half_b + sqrtDisc) / a;
            if (root < t_min || t_max < root) return false;
        }
        record.t = root;
        record.point = ray.getPos(record.t);
        Vec outwardNormal = Normalize((record.point - origin) / radius);
        record.setFaceNormal(ray, outwardNormal);
        record.hitMat = material;
        return true;
    }
    bool boundingBox() override
    {
        bounds.reset(new AABB(origin + V(radius, radius, radius), origin - V(radius, radius, radius)));
        return true;
    }
};

// RaytracingClasses.cs:503-544.  t is computed as written: (-k - origin.Z / direction.Z).
struct RectAxis : Surface {
    float x1, x2, y1, y2, k; const Material* material;
    RectAxis(float a, float b, float c, float d, float kk, const Material* m) : x1(a), x2(b), y1(c), y2(d), k(kk), material(m) {}
    bool hit(const CustomRay& ray, float t_min, float t_max, hitRecord& record) const override
    {
        C().rect_tests++;
        if (t_repeat) C().rect_repeats++;
        else if (t_inert) C().rect_inert++;
        float t = (-k - ray.origin.Z / ray.direction.Z);
        if (t < t_min || t > t_max) return false;
        float x = ray.origin.X + (t * ray.direction.X);
        float y = ray.origin.Y + (t * ray.direction.Y);
        if (x < x1 || x > x2 || y < y1 || y > y2) return false;
        if (t < 0) return false;
        record.t = t;
        Vec outwardNormal = V(0, 0, 1);
        record.setFaceNormal(ray, outwardNormal);
        record.hitMat = material;
        record.point = ray.getPos(t);
        return true;
    }
    // no boundingBox override: Surface.boundingBox() returns false and bounds stays null
};

// RaytracingClasses.cs:547-680 (Moeller-Trumbore, culling by the sign of the determinant)
struct Tri : Surface {
    Vec vertA, vertB, vertC, vANorm, vBNorm, vCNorm, AB, AC;
    bool flipNormal, blend; const Material* material;
    Tri(const float* p, const float* n, const Material* m, bool inFlip, bool blendNormals)
    {
        vertA = V(p[0], p[1], p[2]); vertB = V(p[3], p[4], p[5]); vertC = V(p[6], p[7], p[8]);
        vANorm = V(n[0], n[1], n[2]); vBNorm = V(n[3], n[4], n[5]); vCNorm = V(n[6], n[7], n[8]);
        flipNormal = inFlip; blend = blendNormals; material = m;
        AB = vertB - vertA;
        AC = vertC - vertA;
    }
    bool hit(const CustomRay& ray, float t_min, float t_max, hitRecord& record) const override
    {
        C().tri_tests++;
        if (t_repeat) C().tri_repeats++;
        else if (t_inert) C().tri_inert++;
        Vec pVec = Cross(ray.direction, AC);  // a racy instance field in the reference (:599,679); local here
        float discriminant = Dot(AB, pVec);
        if ((double)discriminant < 1e-8 && !flipNormal) return false;
        else if ((double)discriminant > 1e-8 && flipNormal) return false;
        if ((double)std::fabs(discriminant) < 1e-8) return false;
        float invDisc = 1 / discriminant;
        Vec tVec = ray.origin - vertA;
        float u = Dot(tVec, pVec) * invDisc;
        if (u < 0 || u > 1) return false;
        Vec qVec = Cross(tVec, AB);
        float v = Dot(ray.direction, qVec) * invDisc;
        if (v < 0 || u + v > 1) return false;
        float t = Dot(AC, qVec) * invDisc;
        if (t < 0) return false;
        if (t < t_min || t > t_max) return false;
        record.t = t;
        if (!blend) {
            if (!flipNormal) record.normal = Normalize(Cross(AB, AC));
            else record.normal = -Normalize(Cross(AB, AC));
        } else {
            if (!flipNormal) record.normal = Normalize(Barycentric(vANorm, vBNorm, vCNorm, u, v));
            else record.normal = -Normalize(Barycentric(vANorm, vBNorm, vCNorm, u, v));
        }
        record.frontFace = true;
        record.hitMat = material;
        record.point = ray.getPos(t);
        return true;
    }
    bool boundingBox() override
    {
        Vec mx, mn;
        mx.X = MaxF(MaxF(vertA.X, vertB.X), vertC.X);
        mx.Y = MaxF(MaxF(vertA.Y, vertB.Y), vertC.Y);
        mx.Z = MaxF(MaxF(vertA.Z, vertB.Z), vertC.Z);
        mn.X = MinF(MinF(vertA.X, vertB.X), vertC.X);
        mn.Y = MinF(MinF(vertA.Y, vertB.Y), vertC.Y);
        mn.Z = MinF(MinF(vertA.Z, vertB.Z), vertC.Z);
        bounds.reset(new AABB(mx, mn));
        return true;
    }
};

// RaytracingClasses.cs:282-342
struct SurfaceList : Surface {
    std::vector<Surface*> surfaces;      // not owning
    std::vector<Surface*> surfaceArray;  // snapshot taken by Init()
    void Init()
    {
        if (!surfaces.empty()) {
            for (size_t i = 0; i < surfaces.size(); i++) surfaces[i]->boundingBox();
            boundingBox();
            surfaceArray = surfaces;
        }
    }
    bool hit(const CustomRay& ray, float t_min, float t_max, hitRecord& record) const override
    {
        bool hitAnything = false;
        float closestSoFar = t_max;
        for (size_t i = 0; i < surfaceArray.size(); i++) {
            if (surfaceArray[i]->hit(ray, t_min, closestSoFar, record)) {
                hitAnything = true;
                closestSoFar = record.t;
            }
        }
        return hitAnything;
    }
    // The un-braced else (:330-335) makes the union run on every iteration, the first one
    // against a zero box, so every list's bounds also contain the origin.
    bool boundingBox() override
    {
        if (surfaces.size() < 1) return false;
        bool firstBox = true;
        AABB tempBounds(V(0, 0, 0), V(0, 0, 0));
        for (size_t i = 0; i < surfaces.size(); i++) {
            if (!surfaces[i]->boundingBox()) return false;
            if (firstBox) bounds.reset(new AABB(*surfaces[i]->bounds));
            else tempBounds = *bounds;
            bounds.reset(new AABB(surroundingBox(*surfaces[i]->bounds, tempBounds)));
            firstBox = false;
        }
        return true;
    }
};

// RaytracingClasses.cs:344-446
struct BVHNode : Surface {
    Surface* left = nullptr;
    Surface* right = nullptr;
    std::vector<std::unique_ptr<Surface>> owned;  // child BVHNodes / lists created here
    int depthBelow = 1;
    int nodeCount = 1;

    explicit BVHNode(SurfaceList& list)
    {
        if (!list.bounds) throw NullBounds();
        int axis;
        float sizeX = std::fabs(list.bounds->max.X - list.bounds->min.X);
        float sizeY = std::fabs(list.bounds->max.Y - list.bounds->min.Y);
        float sizeZ = std::fabs(list.bounds->max.Z - list.bounds->min.Z);
        if (sizeX > MaxF(sizeY, sizeZ)) axis = 0;
        else if (sizeY > MaxF(sizeX, sizeZ)) axis = 1;
        else if (sizeZ > MaxF(sizeX, sizeY)) axis = 2;
        else axis = 0;

        if (list.surfaces.size() == 1) {
            left = right = list.surfaces[0];
            bounds.reset(new AABB(*list.bounds));
        } else if (list.surfaces.size() == 2) {
            if (boxCompare(list.surfaces[0], list.surfaces[1], axis)) {
                left = list.surfaces[1];
                right = list.surfaces[0];
            } else {
                left = list.surfaces[0];
                right = list.surfaces[1];
            }
            if (!left->bounds || !right->bounds) throw NullBounds();
            bounds.reset(new AABB(surroundingBox(*right->bounds, *left->bounds)));
        } else {
            auto leftList = std::make_unique<SurfaceList>();
            auto rightList = std::make_unique<SurfaceList>();
            // a single bubble pass, not a sort (:392-401)
            for (size_t i = 0; i + 1 < list.surfaces.size(); i++) {
                if (boxCompare(list.surfaces[i], list.surfaces[i + 1], axis)) {
                    Surface* temp = list.surfaces[i];
                    list.surfaces[i] = list.surfaces[i + 1];
                    list.surfaces[i + 1] = temp;
                }
            }
            size_t mid = list.surfaces.size() / 2;
            for (size_t l = 0; l < mid; l++) leftList->surfaces.push_back(list.surfaces[l]);
            for (size_t r = mid; r < list.surfaces.size(); r++) rightList->surfaces.push_back(list.surfaces[r]);
            leftList->Init();
            rightList->Init();
            // Count > 2 always holds in this branch, so both children are BVHNodes (:412-416)
            auto l = std::make_unique<BVHNode>(*leftList);
            auto r = std::make_unique<BVHNode>(*rightList);
            left = l.get();
            right = r.get();
            depthBelow = 1 + (l->depthBelow > r->depthBelow ? l->depthBelow : r->depthBelow);
            nodeCount = 1 + l->nodeCount + r->nodeCount;
            owned.push_back(std::move(l));
            owned.push_back(std::move(r));
            bounds.reset(new AABB(surroundingBox(*right->bounds, *left->bounds)));
        }
    }
    bool hit(const CustomRay& ray, float t_min, float t_max, hitRecord& record) const override
    {
        if (!bounds->hit(ray)) return false;
        bool hitLeft = false;
        bool hitRight = false;
        if ((hitLeft = left->hit(ray, t_min, t_max, record))) t_max = record.t;
        if (left == right) t_repeat++;   // a leaf with one surface tests it twice (:362-365)
        if ((hitRight = right->hit(ray, t_min, t_max, record))) t_max = record.t;
        if (left == right) t_repeat--;
        return hitLeft || hitRight;
    }
    void leafOrder(std::vector<int>& out) const
    {
        const BVHNode* l = dynamic_cast<const BVHNode*>(left);
        const BVHNode* r = dynamic_cast<const BVHNode*>(right);
        if (l) l->leafOrder(out); else out.push_back(left->primId);
        if (left == right) return;
        if (r) r->leafOrder(out); else out.push_back(right->primId);
    }
};

// RaytracingClasses.cs:67-219
struct Light {
    virtual ~Light() {}
    virtual Vec calculateLight(Vec, const CustomRay&, const Surface*, hitRecord&, const RngKey&, uint32_t, bool) const { return V(0, 0, 0); }
};

struct PointLight : Light {
    Vec position; float brightness; Vec lightColor;
    PointLight(Vec p, float b, Vec c) : position(p), brightness(b), lightColor(c) {}

    // RaytracingClasses.cs:96-188, blinnPhong == false branch (GetLights hard-codes false,
    // RaytracingOperations.cs:202; the Blinn branch is unreachable from the trace loop).
    Vec calculateLight(Vec formerLightLevel, const CustomRay& ray, const Surface* world, hitRecord& record,
                       const RngKey& key, uint32_t lightIndex, bool /*blinnPhong*/) const override
    {
        float roughness = 1 - record.hitMat->smoothness;
        float roughSqr = roughness * roughness;
        float geoShadow = 1;
        float attenuation = ClampF(1 / DistanceSquared(record.point, position), 0, 1);

        Vec directionToLight = Normalize((position - record.point));
        Vec halfAngle = Normalize(directionToLight + Normalize(-ray.direction));

        float NdotL = ClampF(Dot(record.normal, directionToLight), 0, 1);
        float NdotV = ClampF(Dot(record.normal, Normalize(-ray.direction)), 0, 1);
        float LdotH = ClampF(Dot(directionToLight, halfAngle), 0, 1);
        float NdotH = ClampF(Dot(record.normal, halfAngle), 0, 1);

        float ggx = 0;
        {
            float NdotLSqr = NdotL * NdotL;
            float NdotVSqr = NdotV * NdotV;
            float SmithL = 2 / (1 + SqrtF(1 + roughSqr * (1 - NdotLSqr) / NdotLSqr));
            float SmithV = 2 / (1 + SqrtF(1 + roughSqr * (1 - NdotVSqr) / NdotVSqr));
            geoShadow = SmithL * SmithV;

            float F, D, vis;
            float aSqr = roughSqr;
            float pi = (float)3.14159265358979323846;
            float denom = NdotH * NdotH * (aSqr - 1.0f) + 1.0f;
            D = aSqr / (pi * denom * denom);

            float LdotH5 = (float)std::pow((double)(1.0f - LdotH), 5.0);
            float f0 = 0.5f;
            F = f0 + (1.0f - f0) * LdotH5;

            float k = roughSqr / 2.0f;
            vis = (1.0f / (NdotL * (1.0f - k) + k)) * (1.0f / (NdotV * (1.0f - k) + k));

            ggx = NdotL * D * F * vis;
            ggx *= (brightness / 2);
        }

        Vec lightAlbedo = (record.hitMat->albedo + lightColor);
        Vec lightLevel = formerLightLevel
            + ((lightAlbedo * (1 - record.hitMat->metalness)) * (NdotL * NdotV) * geoShadow * brightness * attenuation);

        // Shadow ray.  It is a closest-hit query that overwrites `record`, so from here on
        // record.point / record.t / record.hitMat belong to the occluder if there is one.
        C().rays_shadow++;
        // Bookkeeping only (the ray is traced all the same): with the light behind the surface (NdotL = 0) SmithL, and
        // with it geoShadow, the diffuse term and ggx are 0 -- provided every other factor is finite --, so lightLevel is
        // formerLightLevel and the specular term is 0 whatever the shadow ray meets.  An implementation may leave such a
        // ray out; its tests are counted apart so that the counts of one that does can be checked.  "Finite" is made
        // certain by: roughSqr in [2^-20, 2] (D and vis stay bounded), |directionToLight + Normalize(-direction)|^2 in
        // [2^-60, 8] (a finite half vector), hit point and normal of ordinary size, and ordinary material / light values
        // throughout the scene (t_ordinary).
        bool inert = false;
        if (t_ordinary) {
            const float big = 0x1p40f;
            const float s2 = Magnitude(directionToLight + Normalize(-ray.direction));   // the squared length
            const Vec P = record.point, N = record.normal;
            const bool tame = std::fabs(P.X) <= big && std::fabs(P.Y) <= big && std::fabs(P.Z) <= big &&
                              std::fabs(N.X) <= big && std::fabs(N.Y) <= big && std::fabs(N.Z) <= big;
            inert = NdotL == 0.0f && roughSqr >= 0x1p-20f && roughSqr <= 2.0f && s2 >= 0x1p-60f && s2 <= 8.0f && tame;
        }
        if (inert) { C().shadow_inert++; t_inert++; }
        const bool shadowRayHit = world->hit(CustomRay(record.point, directionToLight + ReturnRandomRangedVector(AuxWord(key, 1u + lightIndex), 0.95f)),
                                             0.001f, std::numeric_limits<float>::infinity(), record);
        if (inert) t_inert--;
        if (shadowRayHit)
            if (Magnitude(CustomRay(record.point, directionToLight).getPos(record.t) - record.point) < Magnitude(position - record.point)) {
                lightLevel = formerLightLevel;
                ggx = 0;
            }

        return lightLevel
            + ((Lerp(V(1, 1, 1) + lightColor, lightAlbedo, record.hitMat->metalness) / (float)3.14159265358979323846) * ggx * attenuation);
    }
};

// Negative controls of the tests that compare renders with the reference's own picture: what the picture would look like
// if the restatement had read one line differently.  Never set outside those tests.
//   bit 0: LightList sums the lights' contributions plainly (f <- f + c_i) instead of the reference's f <- 2 f + c_i
std::atomic<int> g_misreading{0};

struct LightList : Light {
    std::vector<std::unique_ptr<PointLight>> lightArray;
    // RaytracingClasses.cs:205-218: re-intersects the ray once per light; `+=` on a value that
    // already contains formerLightLevel doubles it: f <- 2f + c_i.
    Vec calculateLight(Vec formerLightLevel, const CustomRay& ray, const Surface* world, hitRecord& record,
                       const RngKey& key, uint32_t, bool blinnPhong) const override
    {
        for (size_t i = 0; i < lightArray.size(); i++) {
            C().rays_redundant++;
            t_repeat++;
            const bool again = world->hit(ray, 0.01f, std::numeric_limits<float>::infinity(), record);
            t_repeat--;
            if (again) {
                const Vec level = lightArray[i]->calculateLight(formerLightLevel, ray, world, record, key, (uint32_t)i, blinnPhong);
                formerLightLevel = (g_misreading.load(std::memory_order_relaxed) & 1) ? level : formerLightLevel + level;
            }
        }
        return formerLightLevel;
    }
};

// ---------------------------------------------------------------------------------------
// RayOperations (RaytracingOperations.cs)
// ---------------------------------------------------------------------------------------
const float kInf = std::numeric_limits<float>::infinity();

// :197-205
Vec GetLights(const CustomRay& ray, const Surface* world, const RngKey& key, const Light* light)
{
    hitRecord record;
    Vec formerLightLevel = V(0, 0, 0);
    formerLightLevel = light->calculateLight(formerLightLevel, ray, world, record, key, 0, false);
    return formerLightLevel;
}

// :88-116 -- only the depth edits matter; the `samples` edits feed nothing (the value is
// passed down and never read).
int DepthHeuristic(int depth, float smoothness)
{
    if (smoothness < 0.6f) depth -= (int)((float)depth / 4);
    if (smoothness < 0.4f) depth -= (int)((float)depth / 2);
    if (smoothness < 0.2f) depth -= (int)((float)depth / 2);
    if (smoothness <= 0.05f) depth = 1;
    return ClampI(depth, 1, 999999);
}

Vec Sky(Vec direction)
{
    float t = 0.5f * (direction.Y + 1);
    return t * V(0.2f, 0.17f, 0.17f) + (1 - t) * V(0.25f, 0.22f, 0.23f);
}

// :75-195
Vec GetRayColor(const CustomRay& ray, const Surface* world, RngKey key, int depth, const Light* light, float clamp,
                uint32_t* nodeCounter)
{
    hitRecord record;
    Vec direction = ray.direction;

    if (depth < 1) return V(0, 0, 0);

    C().rays_path++;
    if (nodeCounter) (*nodeCounter)++;
    if (world->hit(ray, 0.01f, kInf, record)) {
        C().hit_nodes++;
        const Material& m = *record.hitMat;
        depth = DepthHeuristic(depth, m.smoothness);

        Vec diffuseDirection = record.normal + ReturnHemisphereRandomVector(key, record.normal);
        Vec reflectedDirection = Reflect(direction, record.normal + ReturnRandomRangedVector(AuxWord(key, 0), SqrtF(m.smoothness)));

        float fresnel = (AngleBetween(direction, reflectedDirection) / 180);
        fresnel *= fresnel;

        Vec finalColor = V(0, 0, 0);
        Vec lightColor = GetLights(ray, world, key, light);
        bool doMetalPass = true;
        bool doReflectPass = true;
        if (m.metalness == 0) doMetalPass = false;
        if (m.metalness == 1) doReflectPass = false;

        RngKey kd = key, km = key, kr = key;
        kd.node = key.node * 4u + 1u;
        km.node = key.node * 4u + 2u;
        kr.node = key.node * 4u + 3u;

        if (doMetalPass && doReflectPass) {
            finalColor = (m.emission + (m.albedo * (1 - m.metalness))
                          * 0.7f * (GetRayColor(CustomRay(record.point, diffuseDirection), world, kd, depth - 1, light, clamp, nodeCounter)))
                + (m.albedo * m.metalness)
                * GetRayColor(CustomRay(record.point, reflectedDirection), world, km, depth - 1, light, clamp, nodeCounter)
                + (1 - m.metalness) * (fresnel * (m.smoothness * m.smoothness))
                * GetRayColor(CustomRay(record.point, reflectedDirection), world, kr, depth - 1, light, clamp, nodeCounter)
                + lightColor;
        } else if (doMetalPass && !doReflectPass) {
            finalColor = (m.emission + (m.albedo * (1 - m.metalness))
                          * 0.7f * (GetRayColor(CustomRay(record.point, diffuseDirection), world, kd, depth - 1, light, clamp, nodeCounter)))
                + (m.albedo * m.metalness)
                * GetRayColor(CustomRay(record.point, reflectedDirection), world, km, depth - 1, light, clamp, nodeCounter)
                + lightColor;
        } else if (!doMetalPass && doReflectPass) {
            finalColor = (m.emission + (m.albedo * (1 - m.metalness))
                          * 0.7f * (GetRayColor(CustomRay(record.point, diffuseDirection), world, kd, depth - 1, light, clamp, nodeCounter)))
                + (1 - m.metalness) * (fresnel * (m.smoothness * m.smoothness))
                * GetRayColor(CustomRay(record.point, reflectedDirection), world, kr, depth - 1, light, clamp, nodeCounter)
                + lightColor;
        }

        finalColor.X = ClampF(finalColor.X, 0, clamp);
        finalColor.Y = ClampF(finalColor.Y, 0, clamp);
        finalColor.Z = ClampF(finalColor.Z, 0, clamp);
        return finalColor;
    }
    return Sky(direction);
}

// :11-21
Vec GetRayNormalColor(const CustomRay& ray, const Surface* world)
{
    hitRecord record;
    if (world->hit(ray, 0, kInf, record)) return 0.5f * (record.normal + V(1, 1, 1));
    Vec direction = Normalize(ray.direction);
    float t = 0.5f * (direction.Y + 1);
    return t * V(0.2f, 0.2f, 0.2f) + (1 - t) * V(0.5f, 0.7f, 1);
}

// :24-35
Vec GetRayDepthColor(const CustomRay& ray, const Surface* world)
{
    hitRecord record;
    if (world->hit(ray, 0, kInf, record)) {
        float value = Distance(ray.origin, record.point);
        return V(1, 1, 1) - (0.07f * V(value, value, value));
    }
    Vec direction = Normalize(ray.direction);
    float t = 0.5f * (direction.Y + 1);
    return t * V(0, 0, 0);
}

// The child of a node in the single-child recursions below: the same id rule as GetRayColor's diffuse child.
inline RngKey FirstChild(RngKey key) { key.node = key.node * 4u + 1u; return key; }

// :38-54
Vec GetRayColorDiffuse(const CustomRay& ray, const Surface* world, RngKey key, int depth)
{
    hitRecord record;
    if (depth < 1) return V(0, 0, 0);
    C().rays_path++;
    if (world->hit(ray, 0.001f, kInf, record)) {
        C().hit_nodes++;
        Vec target = record.point + record.normal + ReturnRandomNormalizedUnitSphereVector(key);
        return 0.5f * (GetRayColorDiffuse(CustomRay(record.point, target - record.point), world, FirstChild(key), depth - 1));
    }
    Vec direction = Normalize(ray.direction);
    float t = 0.5f * (direction.Y + 1);
    return t * V(0.85f, 0.75f, 1.f) + (1 - t) * V(1, 1, 1);
}

// :57-73
Vec GetRayColorReflection(const CustomRay& ray, const Surface* world, RngKey key, int depth)
{
    hitRecord record;
    if (depth < 1) return V(0, 0, 0);
    C().rays_path++;
    if (world->hit(ray, 0.001f, kInf, record)) {
        C().hit_nodes++;
        Vec target = record.point + Reflect(ray.direction, record.normal);
        return 0.5f * (GetRayColorReflection(CustomRay(record.point, target - record.point), world, FirstChild(key), depth - 1));
    }
    Vec direction = Normalize(ray.direction);
    float t = 0.5f * (direction.Y + 1);
    return t * V(0.85f, 0.75f, 1.f) + (1 - t) * V(1, 1, 1);
}

}  // namespace

// ---------------------------------------------------------------------------------------
// Scene container + render driver (Game1.cs)
// ---------------------------------------------------------------------------------------
struct orc_scene {
    std::vector<Material> materials;
    std::vector<std::unique_ptr<Surface>> prims;
    SurfaceList world;
    std::unique_ptr<BVHNode> nodeWorld;
    bool bvhWouldThrow = false;
    LightList lights;
    bool empty = true;
};

namespace {

struct EmptyWorld : Surface {};  // hit() is always false

const Surface* pickWorld(const orc_scene* s, int kind)
{
    static EmptyWorld nothing;
    if (s->empty) return &nothing;
    if (kind == ORC_WORLD_SURFACELIST) return &s->world;
    if (s->bvhWouldThrow || !s->nodeWorld) return nullptr;
    return s->nodeWorld.get();
}

Vec A3(const float* p) { return V(p[0], p[1], p[2]); }

// Albedo and metalness of every material, position, colour and brightness of every light at most 2^40 in size (and
// not NaN): what PointLight.calculateLight's note on shadow rays that cannot matter asks of the scene.
bool sceneIsOrdinary(const orc_scene* s)
{
    const float big = 0x1p40f;
    auto ok = [&](float v) { return std::fabs(v) <= big; };
    for (const Material& m : s->materials)
        if (!(ok(m.albedo.X) && ok(m.albedo.Y) && ok(m.albedo.Z) && ok(m.metalness))) return false;
    for (const auto& l : s->lights.lightArray)
        if (!(ok(l->position.X) && ok(l->position.Y) && ok(l->position.Z) && ok(l->lightColor.X) && ok(l->lightColor.Y) && ok(l->lightColor.Z) && ok(l->brightness))) return false;
    return true;
}

// Game1.cs:264-317, one band of scanlines.
void renderBlock(const orc_scene* s, const Surface* world, const orc_camera* cam, const orc_params* p, int yStart, int yEnd,
                 int xStart, int xEnd, uint32_t* out, uint32_t* nodesPerPixel, Counters* counters, int sampleFirst, int sampleCount)
{
    // sampleFirst / sampleCount: renderBlock runs all samples (0, samples); renderBlockFull (Game1.cs:320-374) is the
    // same loop body run for ONE sample -- still divided by `samples` (:352) -- into a buffer of its own.
    t_cnt = counters;
    t_ordinary = sceneIsOrdinary(s);
    const Vec camPos = A3(cam->position), llc = A3(cam->lower_left), horiz = A3(cam->horizontal), vert = A3(cam->vertical);
    const int samples = p->samples;
    for (int y = yStart; y < yEnd; y++) {
        for (int x = xStart; x < xEnd; x++) {
            Vec vectorRayColor = V(0, 0, 0);
            uint32_t nodes = 0;
            RngKey key;
            key.seed = p->seed;
            key.pixel = (uint32_t)x + (uint32_t)y * (uint32_t)p->width;
            for (int i = sampleFirst; i < sampleFirst + sampleCount; i++) {
                key.sample = (uint32_t)i;
                key.node = 0;
                uint32_t jw = AuxWord(key, 0);
                float u = (float)(x + (float)RngNext(jw, -50, 100) / 100) / p->width;
                float v = (float)(y + (float)RngNext(jw, -50, 100) / 100) / p->height;
                CustomRay ray(camPos, llc + u * horiz + v * vert - camPos);
                key.node = 1;
                Vec c;
                switch (p->mode) {
                    case ORC_MODE_NORMAL: c = GetRayNormalColor(ray, world); break;
                    case ORC_MODE_DEPTH: c = GetRayDepthColor(ray, world); break;
                    case ORC_MODE_LIGHTS: c = GetLights(ray, world, key, &s->lights); break;
                    case ORC_MODE_DIFFUSE: c = GetRayColorDiffuse(ray, world, key, p->max_depth); break;
                    case ORC_MODE_REFLECTION: c = GetRayColorReflection(ray, world, key, p->max_depth); break;
                    default: c = GetRayColor(ray, world, key, p->max_depth, &s->lights, p->color_clamp, &nodes); break;
                }
                vectorRayColor = vectorRayColor + c / (float)samples;
            }
            vectorRayColor.X = SqrtF(vectorRayColor.X);
            vectorRayColor.Y = SqrtF(vectorRayColor.Y);
            vectorRayColor.Z = SqrtF(vectorRayColor.Z);
            // PixelOperations.WritePixel (PixelManagement.cs:8-23): re-clamps bytes, stores at x + y*width
            out[x + (y * p->width)] = PackColor(vectorRayColor.X, vectorRayColor.Y, vectorRayColor.Z);
            if (nodesPerPixel) nodesPerPixel[x + (y * p->width)] = nodes;
        }
    }
    t_cnt = nullptr;
}

}  // namespace

// ---------------------------------------------------------------------------------------
// Sorting.BlurArray / ContrastBlurArray / Despeckle (MiscFunctions.cs:122-416): the post-process filters that
// Game1.Update runs on the finished Color[] (Game1.cs:226-232 calls Despeckle blurPasses times).  They work IN PLACE
// in row-major order (`Color[] originalArray = array` is an alias, not a copy), so a pixel sees the new values of its
// left neighbour and of the row above and the old values of the rest.  Only interior pixels are touched.
// ---------------------------------------------------------------------------------------
inline Vec ColorToBytes(uint32_t c) { return V((float)(c & 255u), (float)((c >> 8) & 255u), (float)((c >> 16) & 255u)); }   // .R .G .B as float
inline bool VecDiffers(Vec a, Vec b) { return a.X != b.X || a.Y != b.Y || a.Z != b.Z; }                                      // Vector3 operator !=
inline float Mixdown(Vec c) { return c.X * 0.3f + c.Y * 0.6f + c.Z * 0.11f; }                                                // :231, :364

struct Neighbours { Vec left, right, up, down, leftUp, rightUp, leftDown, rightDown; };
inline Neighbours ReadNeighbours(const uint32_t* array, int x, int y, int width)
{
    Neighbours n;
    n.left = ColorToBytes(array[x - 1 + (y * width)]);
    n.right = ColorToBytes(array[x + 1 + (y * width)]);
    n.up = ColorToBytes(array[x + ((y - 1) * width)]);
    n.down = ColorToBytes(array[x + ((y + 1) * width)]);
    n.leftUp = ColorToBytes(array[x - 1 + ((y - 1) * width)]);
    n.leftDown = ColorToBytes(array[x - 1 + ((y + 1) * width)]);
    n.rightUp = ColorToBytes(array[x + 1 + ((y - 1) * width)]);
    n.rightDown = ColorToBytes(array[x + 1 + ((y + 1) * width)]);
    return n;
}
// (tempColorLeft + tempColorRight + tempColorUp + tempColorDown + tempColorLeftUp + tempColorLeftDown + tempColorRightUp + tempColorRightDown) / 8
inline Vec MeanOfEight(const Neighbours& n) { return (n.left + n.right + n.up + n.down + n.leftUp + n.leftDown + n.rightUp + n.rightDown) / 8; }
// new Color(v.X / 255, v.Y / 255, v.Z / 255): scalar float / int is a true division
inline uint32_t ColorFromBytes(Vec v) { return PackColor(v.X / 255, v.Y / 255, v.Z / 255); }

void BlurArray(uint32_t* array, int width, int height, float weight)   // MiscFunctions.cs:124-191
{
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
            if (y > 0 && y < height - 1 && x > 0 && x < width - 1) {
                weight = ClampF(weight, 0, 1);
                Vec originalColor = ColorToBytes(array[x + (y * width)]);
                Neighbours n = ReadNeighbours(array, x, y, width);
                Vec newVectorColor = (originalColor * (1 - weight)) + (MeanOfEight(n) * weight);
                array[x + (y * width)] = ColorFromBytes(newVectorColor);
            }
}

void ContrastBlurArray(uint32_t* array, int width, int height, float weight, float contrast)   // MiscFunctions.cs:193-335
{
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
            if (y > 0 && y < height - 1 && x > 0 && x < width - 1) {
                weight = ClampF(weight, 0, 1);
                Vec originalColor = ColorToBytes(array[x + (y * width)]);
                float grayscaleOriginal = Mixdown(originalColor);
                Neighbours n = ReadNeighbours(array, x, y, width);
                // identities in the order they are added (:239-288)
                const Vec identities[8] = {n.left, n.right, n.up, n.down, n.leftUp, n.leftDown, n.rightUp, n.rightDown};
                Vec colorsToBlur[8];
                int count = 0;
                for (const Vec& original : identities) {
                    float grayscale = Mixdown(original);
                    if (grayscaleOriginal >= grayscale) {
                        if (std::fabs(grayscaleOriginal - grayscale) > contrast) colorsToBlur[count++] = original;
                    } else {
                        if (std::fabs(grayscale - grayscaleOriginal) > contrast) colorsToBlur[count++] = original;
                    }
                }
                Vec blurColor;
                if (count != 0) {
                    blurColor = colorsToBlur[0];
                    for (int i = 0; i < count; i++)
                        if (VecDiffers(colorsToBlur[i], colorsToBlur[0])) {
                            blurColor = blurColor / 2;
                            blurColor = blurColor + colorsToBlur[i] / 2;
                        }
                } else
                    blurColor = originalColor;
                Vec newVectorColor = (originalColor * (1 - weight)) + (blurColor * weight);
                array[x + (y * width)] = ColorFromBytes(newVectorColor);
            }
}

void Despeckle(uint32_t* array, int width, int height, float weight, float threshold)   // MiscFunctions.cs:337-416
{
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
            if (y > 0 && y < height - 1 && x > 0 && x < width - 1) {
                weight = ClampF(weight, 0, 1);
                Vec originalColor = ColorToBytes(array[x + (y * width)]);
                float grayscaleOriginal = Mixdown(originalColor);
                Neighbours n = ReadNeighbours(array, x, y, width);
                if (grayscaleOriginal <= 0 || grayscaleOriginal > threshold) {
                    Vec blurColor = (originalColor * (1 - weight)) + (MeanOfEight(n) * weight);
                    Vec newVectorColor = (originalColor * (1 - weight)) + (blurColor * weight);
                    array[x + (y * width)] = ColorFromBytes(newVectorColor);
                }
            }
}


extern "C" {

const char* orc_last_error(void) { return g_error.c_str(); }
void orc_set_misreading(int bits) { g_misreading.store(bits); }

void orc_blur_array(uint32_t* array, int32_t width, int32_t height, float weight) { BlurArray(array, width, height, weight); }
void orc_contrast_blur_array(uint32_t* array, int32_t width, int32_t height, float weight, float contrast) { ContrastBlurArray(array, width, height, weight, contrast); }
void orc_despeckle(uint32_t* array, int32_t width, int32_t height, float weight, float threshold) { Despeckle(array, width, height, weight, threshold); }

orc_scene* orc_scene_create(const orc_material* mats, int32_t n_mats, const orc_tri* tris, int32_t n_tris,
                            const orc_sphere* spheres, int32_t n_spheres, const orc_rect* rects, int32_t n_rects,
                            const orc_light* lights, int32_t n_lights)
{
    return orc_scene_create_ordered(mats, n_mats, tris, n_tris, spheres, n_spheres, rects, n_rects, lights, n_lights, nullptr, nullptr, 0);
}

orc_scene* orc_scene_create_ordered(const orc_material* mats, int32_t n_mats, const orc_tri* tris, int32_t n_tris,
                                    const orc_sphere* spheres, int32_t n_spheres, const orc_rect* rects, int32_t n_rects,
                                    const orc_light* lights, int32_t n_lights,
                                    const uint8_t* order_type, const uint32_t* order_index, int32_t n_order)
{
    auto s = std::make_unique<orc_scene>();
    s->materials.resize(n_mats > 0 ? n_mats : 0);
    for (int i = 0; i < n_mats; i++) {
        s->materials[i].albedo = A3(mats[i].albedo);
        s->materials[i].metalness = mats[i].metalness;
        s->materials[i].smoothness = mats[i].smoothness;
        s->materials[i].emission = A3(mats[i].emission);
    }
    auto matOf = [&](int idx) -> const Material* {
        if (idx < 0 || idx >= n_mats) return nullptr;
        return &s->materials[idx];
    };
    int id = 0;
    for (int i = 0; i < n_tris; i++) {
        const Material* m = matOf(tris[i].material);
        if (!m) { g_error = "triangle material index out of range"; return nullptr; }
        auto t = std::make_unique<Tri>(tris[i].pos, tris[i].nrm, m, tris[i].flip != 0, tris[i].blend != 0);
        t->primId = id++;
        s->world.surfaces.push_back(t.get());
        s->prims.push_back(std::move(t));
    }
    for (int i = 0; i < n_spheres; i++) {
        const Material* m = matOf(spheres[i].material);
        if (!m) { g_error = "sphere material index out of range"; return nullptr; }
        auto sp = std::make_unique<Sphere>(A3(spheres[i].origin), spheres[i].radius, m);
        sp->primId = id++;
        s->world.surfaces.push_back(sp.get());
        s->prims.push_back(std::move(sp));
    }
    for (int i = 0; i < n_rects; i++) {
        const Material* m = matOf(rects[i].material);
        if (!m) { g_error = "rect material index out of range"; return nullptr; }
        auto r = std::make_unique<RectAxis>(rects[i].x1, rects[i].x2, rects[i].y1, rects[i].y2, rects[i].k, m);
        r->primId = id++;
        s->world.surfaces.push_back(r.get());
        s->prims.push_back(std::move(r));
    }
    for (int i = 0; i < n_lights; i++)
        s->lights.lightArray.push_back(std::make_unique<PointLight>(A3(lights[i].position), lights[i].brightness, A3(lights[i].color)));

    if (n_order > 0) {
        // world.surfaces.Add in the caller's order (Game1.cs:117-134, ModelLoader.cs:60); primId stays the index by type
        if ((size_t)n_order != s->prims.size()) { g_error = "the surface order does not name every surface once"; return nullptr; }
        std::vector<Surface*> ordered;
        std::vector<char> seen(s->prims.size(), 0);
        for (int32_t r = 0; r < n_order; r++) {
            const uint32_t t = order_type[r], i = order_index[r];
            const int64_t count = t == 0 ? n_tris : (t == 1 ? n_spheres : (t == 2 ? n_rects : 0));
            if ((int64_t)i >= count) { g_error = "the surface order names a surface that does not exist"; return nullptr; }
            const size_t g = t == 0 ? i : (t == 1 ? (size_t)n_tris + i : (size_t)n_tris + (size_t)n_spheres + i);
            if (seen[g]) { g_error = "the surface order names a surface twice"; return nullptr; }
            seen[g] = 1;
            ordered.push_back(s->prims[g].get());
        }
        s->world.surfaces.assign(ordered.begin(), ordered.end());
    }
    s->empty = s->world.surfaces.empty();
    if (!s->empty) {
        // Game1.cs:136-137.  Init() snapshots the insertion order before the BVH builder's
        // bubble passes permute world.surfaces.
        s->world.Init();
        try {
            SurfaceList builderView;
            builderView.surfaces = s->world.surfaces;
            if (s->world.bounds) builderView.bounds.reset(new AABB(*s->world.bounds));
            s->nodeWorld.reset(new BVHNode(builderView));
        } catch (const NullBounds&) {
            s->bvhWouldThrow = true;  // a RectAxis has no bounds: NullReferenceException in the reference
            s->nodeWorld.reset();
        }
    }
    return s.release();
}

void orc_scene_destroy(orc_scene* s) { delete s; }

int32_t orc_scene_bvh_nodes(const orc_scene* s) { return s && s->nodeWorld ? s->nodeWorld->nodeCount : 0; }
int32_t orc_scene_bvh_depth(const orc_scene* s) { return s && s->nodeWorld ? s->nodeWorld->depthBelow : 0; }
int32_t orc_scene_bvh_leaf_order(const orc_scene* s, int32_t* out, int32_t cap)
{
    if (!s || !s->nodeWorld) return 0;
    std::vector<int> order;
    s->nodeWorld->leafOrder(order);
    int n = (int)order.size() < cap ? (int)order.size() : cap;
    for (int i = 0; i < n; i++) out[i] = order[i];
    return n;
}

static int RenderSamples(const orc_scene* s, const orc_camera* cam, const orc_params* p, uint32_t* out, uint32_t* nodesPerPixel,
                         orc_stats* stats, int sampleFirst, int sampleCount)
{
    if (!s || !cam || !p || !out) { g_error = "null argument"; return -1; }
    if (p->width <= 0 || p->height <= 0 || p->samples <= 0) { g_error = "width, height and samples must be positive"; return -1; }
    if (!(p->color_clamp >= 0)) { g_error = "Math.Clamp(x, 0, clamp) throws when clamp < 0"; return -2; }
    const Surface* world = pickWorld(s, p->world_kind);
    if (!world) { g_error = "new BVHNode(world) dereferences null bounds when the world holds a RectAxis"; return -2; }

    int x0 = p->x0, y0 = p->y0, x1 = p->x1, y1 = p->y1;
    if (x0 == 0 && y0 == 0 && x1 == 0 && y1 == 0) { x1 = p->width; y1 = p->height; }
    if (x0 < 0 || y0 < 0 || x1 > p->width || y1 > p->height || x0 > x1 || y0 > y1) { g_error = "bad window"; return -1; }

    int n = p->threads > 0 ? p->threads : (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    const int rows = y1 - y0;
    std::vector<Counters> counters(n);
    std::vector<std::thread> pool;
    auto t0 = std::chrono::steady_clock::now();
    // Game1.cs:376-380: thread i renders rows [(H/N)*i, (H/N)*(i+1)).  The reference leaves the
    // last H mod N rows unrendered; unless asked to reproduce that, the last band takes them.
    const int band = rows / n;
    for (int i = 0; i < n; i++) {
        int ys = y0 + band * i;
        int ye = y0 + band * (i + 1);
        if (i == n - 1 && !p->drop_remainder_rows) ye = y1;
        if (ye <= ys) continue;
        pool.emplace_back(renderBlock, s, world, cam, p, ys, ye, x0, x1, out, nodesPerPixel, &counters[i], sampleFirst, sampleCount);
    }
    for (auto& t : pool) t.join();
    auto t1 = std::chrono::steady_clock::now();
    if (p->drop_remainder_rows) {
        for (int y = y0 + band * n; y < y1; y++)
            for (int x = x0; x < x1; x++) out[x + y * p->width] = 0;  // default(Color)
    }
    if (stats) {
        Counters total;
        for (auto& c : counters) total.add(c);
        stats->rays_path = total.rays_path; stats->rays_shadow = total.rays_shadow; stats->rays_redundant = total.rays_redundant;
        stats->box_tests = total.box_tests; stats->sphere_tests = total.sphere_tests; stats->tri_tests = total.tri_tests;
        stats->rect_tests = total.rect_tests; stats->hit_nodes = total.hit_nodes;
        stats->box_repeats = total.box_repeats; stats->sphere_repeats = total.sphere_repeats;
        stats->tri_repeats = total.tri_repeats; stats->rect_repeats = total.rect_repeats;
        stats->shadow_inert = total.shadow_inert; stats->box_inert = total.box_inert; stats->sphere_inert = total.sphere_inert;
        stats->tri_inert = total.tri_inert; stats->rect_inert = total.rect_inert;
        stats->seconds = std::chrono::duration<double>(t1 - t0).count();
        stats->threads_used = n;
    }
    return 0;
}

int orc_render(const orc_scene* s, const orc_camera* cam, const orc_params* p, uint32_t* out, uint32_t* nodesPerPixel,
               orc_stats* stats)
{
    return RenderSamples(s, cam, p, out, nodesPerPixel, stats, 0, p ? p->samples : 0);
}

// The "mix buffer" technique (Game1.cs:174-175, :382-388, :320-374, :209-220): every thread renders the WHOLE picture
// with one sample into a buffer of its own -- the colour still divided by `samples` (:352), then gamma and packing --
// and when all are done buffer1Data (default Colors: all zero) takes them in one after the other:
// new Color(buffer1Data.ToVector3() / 2 + buffer.ToVector3() / 2).  Thread b's System.Random becomes sample index b of
// the shared counter RNG; the reference's List.Add from Parallel.For leaves the order of the buffers to chance, here it
// is the thread order.
int orc_render_mix_buffers(const orc_scene* s, const orc_camera* cam, const orc_params* p, int32_t n_buffers, uint32_t* out, orc_stats* stats)
{
    if (!p || n_buffers < 1) { g_error = "n_buffers must be positive"; return -1; }
    const size_t n = (size_t)p->width * (size_t)p->height;
    std::vector<uint32_t> buffer(n);
    for (size_t i = 0; i < n; i++) out[i] = 0u;   // new Color[width * height]
    orc_stats total = {}, one = {};
    for (int32_t b = 0; b < n_buffers; b++) {
        int rc = RenderSamples(s, cam, p, buffer.data(), nullptr, &one, b, 1);
        if (rc != 0) return rc;
        total.rays_path += one.rays_path; total.rays_shadow += one.rays_shadow; total.rays_redundant += one.rays_redundant;
        total.box_tests += one.box_tests; total.sphere_tests += one.sphere_tests; total.tri_tests += one.tri_tests;
        total.rect_tests += one.rect_tests; total.hit_nodes += one.hit_nodes; total.seconds += one.seconds; total.threads_used = one.threads_used;
        total.box_repeats += one.box_repeats; total.sphere_repeats += one.sphere_repeats; total.tri_repeats += one.tri_repeats; total.rect_repeats += one.rect_repeats;
        total.shadow_inert += one.shadow_inert; total.box_inert += one.box_inert; total.sphere_inert += one.sphere_inert; total.tri_inert += one.tri_inert; total.rect_inert += one.rect_inert;
        for (size_t i = 0; i < n; i++) {
            // Color.ToVector3(): (R / 255f, G / 255f, B / 255f)
            Vec tempBuffCol = V((float)(out[i] & 255u) / 255.0f, (float)((out[i] >> 8) & 255u) / 255.0f, (float)((out[i] >> 16) & 255u) / 255.0f);
            Vec currBuffCol = V((float)(buffer[i] & 255u) / 255.0f, (float)((buffer[i] >> 8) & 255u) / 255.0f, (float)((buffer[i] >> 16) & 255u) / 255.0f);
            Vec newBuffCol = (tempBuffCol / 2) + (currBuffCol / 2);
            out[i] = PackColor(newBuffCol.X, newBuffCol.Y, newBuffCol.Z);
        }
    }
    if (stats) *stats = total;
    return 0;
}

// RaytracingClasses.cs:34-50
void orc_camera_ctor(const float origin[3], const float look_at[3], float fov_deg, float ratio, orc_camera* out)
{
    Vec position = A3(origin), lookAt = A3(look_at);
    float theta = ToRadians(fov_deg);
    float h = (float)std::tan((double)(theta / 2));
    float viewportHeight = 2.0f * h;
    float viewportWidth = viewportHeight * ratio;
    Vec w = Normalize(position - lookAt);
    Vec u = Normalize(Cross(V(0, 1, 0), w));
    Vec v = Cross(w, u);
    Vec horizontal = viewportWidth * u;
    Vec vertical = viewportHeight * v;
    Vec lowerLeftCorner = position - horizontal / 2 - vertical / 2 - w;
    auto put = [](float* d, Vec a) { d[0] = a.X; d[1] = a.Y; d[2] = a.Z; };
    put(out->position, position); put(out->lower_left, lowerLeftCorner); put(out->horizontal, horizontal); put(out->vertical, vertical);
}

// RaytracingClasses.cs:223-255
void orc_material_ctor(const float albedo[3], float metal, float smooth, const float emission[3], orc_material* out)
{
    out->albedo[0] = ClampF(albedo[0], 0, 1);
    out->albedo[1] = ClampF(albedo[1], 0, 1);
    out->albedo[2] = ClampF(albedo[2], 0, 1);
    out->metalness = ClampF(metal, 0, 1);
    out->smoothness = ClampF(smooth, 0, 1);
    out->emission[0] = emission[0]; out->emission[1] = emission[1]; out->emission[2] = emission[2];
}

// RaytracingClasses.cs:77-94
void orc_pointlight_ctor(const float origin[3], float brightness, const float* color, orc_light* out)
{
    out->position[0] = origin[0]; out->position[1] = origin[1]; out->position[2] = origin[2];
    out->brightness = brightness;
    out->color[0] = out->color[1] = out->color[2] = 0;
    if (color) {
        out->color[0] = color[0]; out->color[1] = color[1]; out->color[2] = color[2];
        float biggestColor1 = MaxF(color[0], color[1]);
        float biggestColor2 = MaxF(color[1], color[2]);
        float biggestColor3 = MaxF(biggestColor1, biggestColor2);
        if (biggestColor3 != 0 && biggestColor3 >= 1) out->brightness /= biggestColor3;
    }
}

static void putRec(const hitRecord& r, const std::vector<Material>* mats, float rec[9], int matIndex)
{
    rec[0] = r.t; rec[1] = r.point.X; rec[2] = r.point.Y; rec[3] = r.point.Z;
    rec[4] = r.normal.X; rec[5] = r.normal.Y; rec[6] = r.normal.Z; rec[7] = r.frontFace ? 1.0f : 0.0f; rec[8] = (float)matIndex;
    (void)mats;
}

int orc_sphere_hit(const orc_sphere* sp, const float o[3], const float d[3], float t_min, float t_max, float rec[9])
{
    Material m{};
    Sphere s(A3(sp->origin), sp->radius, &m);
    hitRecord r;
    bool h = s.hit(CustomRay(A3(o), A3(d)), t_min, t_max, r);
    if (h) putRec(r, nullptr, rec, sp->material);
    return h ? 1 : 0;
}
int orc_rect_hit(const orc_rect* rc, const float o[3], const float d[3], float t_min, float t_max, float rec[9])
{
    Material m{};
    RectAxis s(rc->x1, rc->x2, rc->y1, rc->y2, rc->k, &m);
    hitRecord r;
    bool h = s.hit(CustomRay(A3(o), A3(d)), t_min, t_max, r);
    if (h) putRec(r, nullptr, rec, rc->material);
    return h ? 1 : 0;
}
int orc_tri_hit(const orc_tri* tr, const float o[3], const float d[3], float t_min, float t_max, float rec[9])
{
    Material m{};
    Tri s(tr->pos, tr->nrm, &m, tr->flip != 0, tr->blend != 0);
    hitRecord r;
    bool h = s.hit(CustomRay(A3(o), A3(d)), t_min, t_max, r);
    if (h) putRec(r, nullptr, rec, tr->material);
    return h ? 1 : 0;
}
int orc_aabb_hit(const float bmin[3], const float bmax[3], const float o[3], const float d[3])
{
    AABB b(A3(bmax), A3(bmin));
    return b.hit(CustomRay(A3(o), A3(d))) ? 1 : 0;
}
void orc_sky(const float d[3], float rgb[3])
{
    Vec c = Sky(A3(d));
    rgb[0] = c.X; rgb[1] = c.Y; rgb[2] = c.Z;
}
int orc_depth_heuristic(int depth, float smoothness) { return DepthHeuristic(depth, smoothness); }
uint32_t orc_pack_color(float r, float g, float b) { return PackColor(r, g, b); }
void orc_rng_words(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t node, uint32_t slot, uint32_t out[4])
{
    RngKey k{seed, pixel, sample, node};
    RngWords(k, slot, out);
}
int orc_rng_next(uint32_t* word, int lo, int range) { return RngNext(*word, lo, range); }
void orc_random_vector(uint32_t word, float out[3])
{
    Vec v = ReturnRandomVector(word);
    out[0] = v.X; out[1] = v.Y; out[2] = v.Z;
}
void orc_hemisphere_vector(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t node, const float n[3], float out[3])
{
    RngKey k{seed, pixel, sample, node};
    Vec v = ReturnHemisphereRandomVector(k, A3(n));
    out[0] = v.X; out[1] = v.Y; out[2] = v.Z;
}
float orc_angle_between(const float a[3], const float b[3]) { return AngleBetween(A3(a), A3(b)); }

int orc_get_lights(const orc_scene* s, int world_kind, const float o[3], const float d[3], uint32_t seed, uint32_t pixel,
                   uint32_t sample, uint32_t node, float rgb[3])
{
    const Surface* world = pickWorld(s, world_kind);
    if (!world) return -2;
    CustomRay ray(A3(o), A3(d));
    hitRecord probe;
    int hit = world->hit(ray, 0.01f, kInf, probe) ? 1 : 0;
    RngKey k{seed, pixel, sample, node};
    Vec c = GetLights(ray, world, k, &s->lights);
    rgb[0] = c.X; rgb[1] = c.Y; rgb[2] = c.Z;
    return hit;
}

void orc_get_ray_color(const orc_scene* s, int world_kind, const float o[3], const float d[3], int depth, float clamp,
                       uint32_t seed, uint32_t pixel, uint32_t sample, float rgb[3])
{
    const Surface* world = pickWorld(s, world_kind);
    rgb[0] = rgb[1] = rgb[2] = 0;
    if (!world) return;
    RngKey k{seed, pixel, sample, 1};
    Vec c = GetRayColor(CustomRay(A3(o), A3(d)), world, k, depth, &s->lights, clamp, nullptr);
    rgb[0] = c.X; rgb[1] = c.Y; rgb[2] = c.Z;
}

// mode: ORC_MODE_DIFFUSE or ORC_MODE_REFLECTION
void orc_get_debug_color(const orc_scene* s, int world_kind, int mode, const float o[3], const float d[3], int depth, uint32_t seed,
                         uint32_t pixel, uint32_t sample, float rgb[3])
{
    const Surface* world = pickWorld(s, world_kind);
    rgb[0] = rgb[1] = rgb[2] = 0;
    if (!world) return;
    RngKey k{seed, pixel, sample, 1};
    CustomRay ray(A3(o), A3(d));
    Vec c = mode == ORC_MODE_DIFFUSE ? GetRayColorDiffuse(ray, world, k, depth) : GetRayColorReflection(ray, world, k, depth);
    rgb[0] = c.X; rgb[1] = c.Y; rgb[2] = c.Z;
}

}  // extern "C"
