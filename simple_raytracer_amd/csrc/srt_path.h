// srt_path.h -- the ray tree of one sample as a per-lane state machine.
//
// A work item is one (pixel, sample): the lane walks the sample's ray tree depth first
// (RayOperations.GetRayColor, RaytracingOperations.cs:75-195) and leaves the sample's colour in
// a sample buffer; a second kernel adds the samples of each pixel in index order, which is the
// order Game1.cs:281-296 adds them in, so the float sum is the same.
//
// A lane is in one of three phases:
//   TRACE   it holds the closest-hit query of a tree node (world.hit(ray, 0.01, inf, record));
//   LIGHT   the node was hit: it still owes one shadow query per light (LightList.calculateLight,
//           RaytracingClasses.cs:205-218) and then the node's shading, which either starts the
//           first child (back to TRACE) or finds that the node has no children to trace (RETURN);
//   RETURN  a GetRayColor call has its value (a miss, or a node without traced children): it is handed
//           up the tree until some parent has another child to start (TRACE) or the sample is done;
//   WAIT    (mesh kernel only) the next child of a pending node was handed to another lane and is not back yet.
// The kernel runs the phases as separate, wave-wide code sections, so that lanes which execute a
// section all execute the same code.
#ifndef SRT_PATH_H
#define SRT_PATH_H

#include "srt_core.h"

namespace srt {

enum { LANE_IDLE = 0, LANE_TRACE = 1, LANE_LIGHT = 2, LANE_RETURN = 3, LANE_WAIT = 4 };

struct Lane {
    uint32_t slot;      // where the sample's colour goes in the sample buffer
    RngKey key;         // key.node = id of the tree node being evaluated
    int32_t depth;      // `depth` argument of the current GetRayColor call
    int32_t level;      // frames in use
    f3 qo, qd;          // pending closest-hit query (TRACE) / the node's ray (LIGHT: qd is its direction)
    float qtmin;
    f3 P, N;            // hit being lit (LIGHT)
    uint32_t mat;
    f3 former;          // formerLightLevel (LIGHT) / the value being returned (RETURN)
    uint32_t rays_path, rays_shadow, hit_nodes;
};

// Work item -> (pixel, sample).  Items enumerate 64-pixel blocks (8x8 pixels of one tile), and inside
// a block all samples of the pass, 64 pixels at a time, so that 64 consecutive items are one 8x8
// pixel block at one sample index; the last samples of a pass may instead come one after the other, and inside a
// sample 64 pixels from 64 different blocks at a time (RenderArgs::coherent).  Returns false for padding pixels of edge tiles.
SRT_HD bool item_start(Lane& ln, const RenderArgs& ra, uint32_t item, const float* unit)
{
    uint32_t s_local, wp;
    const uint32_t n_coherent = ra.n_work * ra.coherent;
    if (item < n_coherent) {
        uint32_t per_block = 64u * ra.coherent;
        uint32_t blk = item / per_block, r = item - blk * per_block;
        s_local = r >> 6;
        wp = blk * 64u + (r & 63u);
    } else {
        uint32_t rest = item - n_coherent, s = rest / ra.n_work;
        uint32_t j = rest - s * ra.n_work, k = j & 63u;
        s_local = ra.coherent + s;
        wp = (((j >> 6) + k * ra.scatter) % (ra.n_work >> 6)) * 64u + k;
    }
    int32_t x, y;
    uint32_t unused;
    if (!work_to_pixel(ra, wp, x, y, unused)) return false;
    ln.slot = s_local * ra.n_work + wp;
    ln.key.seed = ra.seed;
    ln.key.pixel = (uint32_t)x + (uint32_t)y * (uint32_t)ra.width;
    ln.key.sample = ra.sample_first + s_local;
    // Game1.cs:283-287: jittered viewport coordinates and the primary ray
    // (Next(-50,50) / 100f: two base-100 digits of word 0, and (d - 50) / 100 is entry d + 50 of the unit table)
    ln.key.node = 0u;
    uint32_t w[4];
    rng_call(ln.key, CALL_AUX, w);
    uint32_t ju = rng_digit(w[0], 100u), jv = rng_digit(w[0], 100u);
    float u = ((float)x + unit[ju + 50u]) / (float)ra.width;
    float v = ((float)y + unit[jv + 50u]) / (float)ra.height;
    const Camera& cm = ra.cam;
    f3 dir = mk3(((cm.lower_left.x + u * cm.horizontal.x) + v * cm.vertical.x) - cm.position.x,
                 ((cm.lower_left.y + u * cm.horizontal.y) + v * cm.vertical.y) - cm.position.y,
                 ((cm.lower_left.z + u * cm.horizontal.z) + v * cm.vertical.z) - cm.position.z);
    ln.qo = cm.position;
    ln.qd = normalize3(dir);
    const uint32_t mode = (uint32_t)ra.mode;   // t_min of world.hit: 0 (normal, depth), 0.001 (diffuse, reflection), 0.01
    ln.qtmin = (mode == MODE_NORMAL || mode == MODE_DEPTH) ? 0.0f : ((mode == MODE_DIFFUSE || mode == MODE_REFLECTION) ? 0.001f : 0.01f);
    ln.key.node = 1u;
    ln.depth = ra.max_depth;
    ln.level = 0;
    return true;
}

// Where the frames of pending parents live.  All stores hold the same Frame per level; they differ in the memory
// behind a level.
//   PrivateFrames   an array in private memory (the CPU harness);
//   SlabFrames      a slab of device memory, [wavefront][level][lane][64 bytes]: a frame is one whole 64-byte segment
//                   of its own, written with four 16-byte stores.  Private memory interleaves the lanes word by word,
//                   so a push by a few lanes of a wavefront dirties 14 different segments of which it fills 4 bytes each,
//                   and what the caches write back is several times the frames' size; here a push writes the bytes of
//                   the frames pushed, whatever lanes take part.
//   HybridFrames    the first LDS_LEVELS levels of the recursion in LDS (the depth heuristic ends most branches of the
//                   ray tree after a few levels, so more than half of all pushes and pops are there), laid out
//                   [level][field][lane] so that the 64 lanes of a wavefront hit 64 different banks; deeper levels in
//                   the slab.
struct PrivateFrames {
    Frame* a;
    SRT_HD Frame load(int32_t level) const { return a[level]; }
    SRT_HD void store(int32_t level, const Frame& f) const { a[level] = f; }
    SRT_HD void store_acc(int32_t level, f3 acc) const { a[level].acc = acc; }
};

enum { FRAME_SLAB_LEVEL_BYTES = 64 * 64 };   // between two levels of a lane: 64 lanes x 64 bytes
struct SlabFrames {
    char* slab;      // the whole slab (the same pointer on every lane, so it can live in scalar registers)
    uint32_t mine;   // byte offset of this lane's frame of level 0
    SRT_HD q4* at(int32_t level) const { return reinterpret_cast<q4*>(slab + (mine + (uint32_t)level * (uint32_t)FRAME_SLAB_LEVEL_BYTES)); }
    SRT_HD Frame load(int32_t level) const
    {
        const q4* p = at(level);
        const q4 a = p[0], b = p[1], c = p[2], d = p[3];
        Frame f;
        f.acc = mk3(a.x, a.y, a.z);
        f.light = mk3(a.w, b.x, b.y);
        f.point = mk3(b.z, b.w, c.x);
        f.refl = mk3(c.y, c.z, c.w);
        f.w_refl = d.x;
        f.meta = fbits(d.y);
        return f;
    }
    SRT_HD void store(int32_t level, const Frame& f) const
    {
        q4* p = at(level);
        q4 a, b, c, d;
        a.x = f.acc.x; a.y = f.acc.y; a.z = f.acc.z; a.w = f.light.x;
        b.x = f.light.y; b.y = f.light.z; b.z = f.point.x; b.w = f.point.y;
        c.x = f.point.z; c.y = f.refl.x; c.z = f.refl.y; c.w = f.refl.z;
        d.x = f.w_refl; d.y = bitsf(f.meta); d.z = 0.0f; d.w = 0.0f;
        p[0] = a; p[1] = b; p[2] = c; p[3] = d;
    }
    SRT_HD void store_acc(int32_t level, f3 acc) const
    {
        float* p = reinterpret_cast<float*>(at(level));
        p[0] = acc.x; p[1] = acc.y; p[2] = acc.z;
    }
};

template <int LDS_LEVELS>
struct HybridFrames {
    SlabFrames deep;   // levels LDS_LEVELS and up, at slab level (level - LDS_LEVELS)
    float* lds;        // this lane's column of the wavefront's LDS slab
    SRT_HD Frame load(int32_t level) const
    {
        if (level >= LDS_LEVELS) return deep.load(level - LDS_LEVELS);
        const float* p = lds + level * (14 * 64);
        Frame f;
        f.acc = mk3(p[0 * 64], p[1 * 64], p[2 * 64]);
        f.light = mk3(p[3 * 64], p[4 * 64], p[5 * 64]);
        f.point = mk3(p[6 * 64], p[7 * 64], p[8 * 64]);
        f.refl = mk3(p[9 * 64], p[10 * 64], p[11 * 64]);
        f.w_refl = p[12 * 64];
        f.meta = fbits(p[13 * 64]);
        return f;
    }
    SRT_HD void store(int32_t level, const Frame& f) const
    {
        if (level >= LDS_LEVELS) { deep.store(level - LDS_LEVELS, f); return; }
        float* p = lds + level * (14 * 64);
        p[0 * 64] = f.acc.x; p[1 * 64] = f.acc.y; p[2 * 64] = f.acc.z;
        p[3 * 64] = f.light.x; p[4 * 64] = f.light.y; p[5 * 64] = f.light.z;
        p[6 * 64] = f.point.x; p[7 * 64] = f.point.y; p[8 * 64] = f.point.z;
        p[9 * 64] = f.refl.x; p[10 * 64] = f.refl.y; p[11 * 64] = f.refl.z;
        p[12 * 64] = f.w_refl;
        p[13 * 64] = bitsf(f.meta);
    }
    SRT_HD void store_acc(int32_t level, f3 acc) const
    {
        if (level >= LDS_LEVELS) { deep.store_acc(level - LDS_LEVELS, acc); return; }
        float* p = lds + level * (14 * 64);
        p[0 * 64] = acc.x; p[1 * 64] = acc.y; p[2 * 64] = acc.z;
    }
};

// Where the time goes: counting builds pass a clock-reading profiler to the per-lane functions, which mark
// the end of each of their parts; everything else passes NoProf, whose marks compile to nothing.
struct NoProf { SRT_HD void mark(int) const {} };
enum { PROF_DEAL = 0, PROF_RETURN, PROF_CLOSEST, PROF_ON_CLOSEST, PROF_AUX, PROF_SHADOW_SETUP, PROF_SHADOW_QUERY, PROF_LIGHT_EVAL,
       PROF_CHILD_DIRS, PROF_FRESNEL, PROF_CHILD_START,
       PROF_WAIT_RETURN = 11, PROF_WAIT_DEAL, PROF_WAIT_TRACE, PROF_WAIT_LIGHT, PROF_SLOTS = 16 };

SRT_HD void lane_begin_child(Lane& ln, uint32_t child, f3 origin, f3 unit_dir, int32_t depth)
{
    ln.key.node = ln.key.node * 4u + child + 1u;
    ln.qo = origin; ln.qd = unit_dir; ln.qtmin = 0.01f;
    ln.depth = depth;
}

SRT_HD f3 clamp_color(f3 c, float hi) { return mk3(clampf(c.x, 0.0f, hi), clampf(c.y, 0.0f, hi), clampf(c.z, 0.0f, hi)); }

// RETURN phase: the GetRayColor call of node ln.key.node returned ln.former: hand it to the parent, start
// the next child or complete parents up the tree.  Returns LANE_TRACE with a new query, or LANE_IDLE
// with `ret` holding the colour of the whole sample.
// A kernel may hand children 1 and 2 of one of a lane's pending nodes (one frame at a time) to other lanes: the RNG is
// keyed by the node id, so a subtree evaluates to the same colour wherever it runs.  `fork.donated(level, child)` says
// whether that child of the frame at `level` was given away; lane_return then stops with LANE_WAIT after noting the
// child in fork.wait_for(child), with the sum so far stored in the frame, the frame still on the stack and
// ln.key.node = the frame's node id.  The kernel resumes the lane in the RETURN phase with ln.key.node = the child's
// id and ln.former = the child's colour, exactly as if the child had just returned.
struct NoFork {
    SRT_HD bool donated(int32_t, uint32_t) const { return false; }
    SRT_HD void wait_for(uint32_t) const {}
};

template <class FRAMES, class FORK = NoFork>
SRT_HD int lane_return(Lane& ln, f3& ret, const SceneView& sc, const RenderArgs& ra, const FRAMES& frames, FORK&& fork = FORK())
{
    ret = ln.former;
    while (ln.level > 0) {
        const Frame f = frames.load(ln.level - 1);
        uint32_t mat = f.meta & 0x00FFFFFFu;
        int32_t child_depth = (int32_t)(f.meta >> 24);
        Material m = load_material(sc, mat);
        uint32_t child = (ln.key.node & 3u) - 1u;  // which child just returned
        ln.key.node >>= 2;                          // back to the parent's id
        f3 acc = f.acc;
        uint32_t next;
        // RaytracingOperations.cs:156-183: emission + albedo(1-m) 0.7 C_diffuse + albedo m C_metal
        //                                  + (1-m) fresnel s^2 C_reflect + lightColor, left to right
        if (child == 0u) {
            f3 wd = scale3(scale3(m.albedo, 1.0f - m.metalness), 0.7f);
            acc = add3(acc, mul3(wd, ret));
            next = (m.metalness != 0.0f) ? 1u : 2u;
        } else if (child == 1u) {
            acc = add3(acc, mul3(scale3(m.albedo, m.metalness), ret));
            next = (m.metalness != 1.0f) ? 2u : 3u;
        } else {
            acc = add3(acc, scale3(ret, f.w_refl));
            next = 3u;
        }
        if (next < 3u) {
            frames.store_acc(ln.level - 1, acc);
            if (fork.donated(ln.level - 1, next)) { fork.wait_for(next); return LANE_WAIT; }
            lane_begin_child(ln, next, f.point, f.refl, child_depth);
            return LANE_TRACE;
        }
        ret = clamp_color(add3(acc, f.light), ra.color_clamp);  // :186-188, at every level
        ln.level--;
    }
    return LANE_IDLE;
}

// All lights of the node are in (light_color = lightColor of RaytracingOperations.cs:144): directions of the
// children, fresnel, and either the first child (LANE_TRACE) or -- when the children would run at depth < 1 -- the
// node's value in ln.former (LANE_RETURN).  LANE_IDLE with `ret` set ends the sample (lights-only mode).
template <class FRAMES, class PROF = NoProf>
SRT_HD int lane_shade(Lane& ln, f3 light_color, f3& ret, const SceneView& sc, const RenderArgs& ra, const FRAMES& frames, const uint32_t aux[4],
                      const float* unit, PROF&& prof = PROF())
{
    if ((uint32_t)ra.mode == MODE_LIGHTS) { ret = light_color; return LANE_IDLE; }
    Material m = load_material(sc, ln.mat);
    f3 hemi = random_hemisphere_vector(ln.key, ln.N, unit);
    f3 diffuse_dir = add3(ln.N, hemi);                                                  // :119
    prof.mark(PROF_CHILD_DIRS);
    f3 rough = ranged_vector_from_word(aux[0], sqrt_rn(m.smoothness), unit);
    f3 refl = reflect3(ln.qd, add3(ln.N, rough));                                       // :128
    float fres = fresnel_term(ln.qd, refl);                                             // :137-140
    float w_refl = (1.0f - m.metalness) * (fres * (m.smoothness * m.smoothness));
    prof.mark(PROF_FRESNEL);
    int32_t child_depth = ln.depth - 1;
    if (child_depth < 1) {
        // every child returns Vector3.Zero without tracing (:82-83)
        f3 zero = mk3(0.0f, 0.0f, 0.0f);
        f3 wd = scale3(scale3(m.albedo, 1.0f - m.metalness), 0.7f);
        f3 acc = add3(m.emission, mul3(wd, zero));
        if (m.metalness != 0.0f) acc = add3(acc, mul3(scale3(m.albedo, m.metalness), zero));
        if (m.metalness != 1.0f) acc = add3(acc, scale3(zero, w_refl));
        ln.former = clamp_color(add3(acc, light_color), ra.color_clamp);
        prof.mark(PROF_CHILD_START);
        return LANE_RETURN;
    }
    Frame f;
    f.acc = m.emission;
    f.light = light_color;
    f.point = ln.P;
    f.refl = normalize3(refl);  // CustomRay normalises; the metal and the reflect child share the direction
    f.w_refl = w_refl;
    f.meta = ln.mat | ((uint32_t)child_depth << 24);
    frames.store(ln.level, f);
    ln.level++;
    lane_begin_child(ln, 0u, ln.P, normalize3(diffuse_dir), child_depth);
    prof.mark(PROF_CHILD_START);
    return LANE_TRACE;
}

// TRACE phase: the closest-hit query came back.  Returns the lane's next phase; LANE_IDLE means the
// sample is complete and `ret` is its colour, LANE_RETURN that ln.former holds the node's value.
template <class FRAMES>
SRT_HD int lane_on_closest(Lane& ln, const Hit& h, f3& ret, const SceneView& sc, const RenderArgs& ra, const FRAMES& frames, const float* unit)
{
    const uint32_t mode = (uint32_t)ra.mode;
    if (mode == MODE_NORMAL || mode == MODE_DEPTH) {
        // RaytracingOperations.cs:11-35; the miss branch normalises the direction once more
        if (h.prim != NO_PRIM) {
            HitRecord r = resolve_hit(sc, h, ln.qo, ln.qd);
            if (mode == MODE_NORMAL) ret = mk3(0.5f * (r.normal.x + 1.0f), 0.5f * (r.normal.y + 1.0f), 0.5f * (r.normal.z + 1.0f));
            else {
                float value = sqrt_rn(distsq3(ln.qo, r.point));
                float c = 1.0f - (0.07f * value);
                ret = mk3(c, c, c);
            }
        } else {
            f3 dn = normalize3(ln.qd);
            float t = 0.5f * (dn.y + 1.0f);
            if (mode == MODE_NORMAL) ret = mk3(t * 0.2f + (1.0f - t) * 0.5f, t * 0.2f + (1.0f - t) * 0.7f, t * 0.2f + (1.0f - t) * 1.0f);
            else ret = mk3(t * 0.0f, t * 0.0f, t * 0.0f);
        }
        return LANE_IDLE;
    }
    if (mode == MODE_DIFFUSE || mode == MODE_REFLECTION) {
        // GetRayColorDiffuse / GetRayColorReflection (RaytracingOperations.cs:38-73): one child per hit, the miss
        // colour halved once per level on the way back; ln.level counts the levels.
        if (ln.depth < 1) { ret = mk3(0.0f, 0.0f, 0.0f); return LANE_IDLE; }
        ln.rays_path++;
        if (h.prim == NO_PRIM) {
            f3 dn = normalize3(ln.qd);
            float t = 0.5f * (dn.y + 1.0f);
            ret = mk3(t * 0.85f + (1.0f - t) * 1.0f, t * 0.75f + (1.0f - t) * 1.0f, t * 1.0f + (1.0f - t) * 1.0f);
            for (int32_t i = 0; i < ln.level; i++) ret = mk3(0.5f * ret.x, 0.5f * ret.y, 0.5f * ret.z);
            return LANE_IDLE;
        }
        ln.hit_nodes++;
        HitRecord r = resolve_hit(sc, h, ln.qo, ln.qd);
        f3 target;
        if (mode == MODE_DIFFUSE) target = add3(add3(r.point, r.normal), normalize3(random_unit_sphere_vector(ln.key, unit)));
        else target = add3(r.point, reflect3(ln.qd, r.normal));
        ln.level++;
        lane_begin_child(ln, 0u, r.point, normalize3(sub3(target, r.point)), ln.depth - 1);
        ln.qtmin = 0.001f;
        return LANE_TRACE;
    }
    if (mode == MODE_LIGHTS) {
        // GetLights on the primary ray (RaytracingOperations.cs:197-205)
        if (h.prim == NO_PRIM || sc.n_lights == 0u) { ret = mk3(0.0f, 0.0f, 0.0f); return LANE_IDLE; }
    } else {
        // :82-83: depth < 1 returns black before anything is traced (a primary ray with max_depth 0)
        if (ln.depth < 1) { ln.former = mk3(0.0f, 0.0f, 0.0f); return LANE_RETURN; }
        ln.rays_path++;
        if (h.prim == NO_PRIM) { ln.former = sky_color(ln.qd.y); return LANE_RETURN; }
        ln.hit_nodes++;
    }
    HitRecord r = resolve_hit(sc, h, ln.qo, ln.qd);
    ln.P = r.point; ln.N = r.normal; ln.mat = r.material;
    if (mode != MODE_LIGHTS) ln.depth = depth_heuristic(ln.depth, sc.materials[2 * r.material + 1].x);
    ln.former = mk3(0.0f, 0.0f, 0.0f);
    if (sc.n_lights == 0u) {
        uint32_t aux[4];
        rng_call(ln.key, CALL_AUX, aux);
        return lane_shade(ln, ln.former, ret, sc, ra, frames, aux, unit);
    }
    return LANE_LIGHT;
}

// LIGHT phase, light `li`: its shadow ray ...
SRT_HD void lane_shadow_query(const Lane& ln, const SceneView& sc, uint32_t li, const uint32_t aux[4], const float* unit, f3& o, f3& d)
{
    Light L = load_light(sc, li);
    o = ln.P;
    d = shadow_direction(L, ln.P, light_word(ln.key, aux, li), unit);
}
// Whether that query has to be made at all (see shadow_ray_matters); when not, the light is evaluated as if the ray had
// hit nothing.
SRT_HD bool lane_shadow_matters(const Lane& ln, const SceneView& sc, uint32_t li)
{
    return shadow_ray_matters(sc, load_light(sc, li), ln.P, ln.N, ln.qd, sc.materials[2 * ln.mat + 1].x);
}
// The contribution of a light whose shadow ray cannot matter: calculateLight returns formerLightLevel bit for bit
// (shadow_ray_matters), so formerLightLevel += calculateLight(...) doubles it.  The ray still counts as a shadow ray.
SRT_HD void lane_skip_light(Lane& ln)
{
    ln.rays_shadow++;
    ln.former = add3(ln.former, ln.former);
}
// ... and its contribution once the shadow ray (o, d) came back: formerLightLevel += calculateLight(formerLightLevel, ...)
SRT_HD void lane_on_shadow(Lane& ln, const Hit& h, f3 o, f3 d, const SceneView& sc, uint32_t li)
{
    ln.rays_shadow++;
    Material m = load_material(sc, ln.mat);
    Light L = load_light(sc, li);
    bool sh = h.prim != NO_PRIM;
    f3 P_s = ln.P;
    float metal_s = m.metalness;
    if (sh) {
        P_s = ray_at(o, d, h.t);
        metal_s = sc.materials[2 * hit_material(sc, h)].w;
    }
    f3 c = point_light(L, ln.P, ln.N, ln.qd, m, ln.former, sh, h.t, P_s, metal_s);
    ln.former = add3(ln.former, c);
}

// Adds the samples of one pass to a pixel in sample order (Game1.cs:295: vectorRayColor += c / samples,
// Vector3 / float being a multiplication by 1/samples).
SRT_HD f3 accumulate_samples(f3 acc, const f3* samples, uint32_t stride, uint32_t wp, uint32_t count, float inv)
{
    for (uint32_t s = 0; s < count; s++) {
        f3 c = samples[(size_t)s * stride + wp];
        acc = mk3(acc.x + c.x * inv, acc.y + c.y * inv, acc.z + c.z * inv);
    }
    return acc;
}
// Game1.cs:301-306: gamma (sqrt per channel) and packing.
SRT_HD uint32_t finish_color(f3 acc) { return pack_color(sqrt_rn(acc.x), sqrt_rn(acc.y), sqrt_rn(acc.z)); }

}  // namespace srt
#endif  // SRT_PATH_H
