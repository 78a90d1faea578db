// srt_wavefront.hip -- scenes whose tree is walked with a stack, traced level by level.
//
// mesh_trace_kernel keeps a whole ray tree in one lane: the lane walks its traversals, shades its hits and hands values up
// the tree, and the wavefront runs those parts one after the other, each with the lanes that happen to be there.  On
// scenes whose rays scatter (a cloud of spheres around a mesh) the parts other than the traversal run a third full.
// Here the nodes of the ray trees are records in device memory instead, and every part is a kernel of its own that runs
// over all nodes of one level of all trees of a chunk of work items:
//
//   generate                 the primary rays of the chunk's items: the nodes of level 0
//   per level l:
//     walk<closest>          the closest-hit query of every node of the level (persistent wavefronts, lanes refilled as
//                            their traversals end; nothing but traversal, so few registers: 8 wavefronts per SIMD)
//     hit                    misses get their value (the sky); hits their record, the depth heuristic and one shadow
//                            query per light whose ray can matter (lane_on_closest, lane_shadow_matters)
//     walk<shadow>           those queries
//     shade                  the lights in order (lane_on_shadow / lane_skip_light), the directions of the children,
//                            fresnel (lane_shade); a node without traced children gets its value, the others a frame and
//                            two or three child nodes in level l + 1
//   combine, deepest level first: a node's value from its frame and its children's values (lane_return)
//   finish                   the values of level 0 are the samples' colours
//
// Every per-node function is the one the other kernels call, with the same arguments in the same order, and the RNG is
// keyed by (pixel, sample, node id): the colours are the same bits.  The time of a chunk no longer depends on its
// largest tree (3,280 nodes at depth 8) but on the depth.
//
// The node pool has a fixed capacity.  A chunk whose trees outgrow it is given up: its kernels stop creating nodes, the
// finish kernel writes nothing and adds nothing to the statistics, and mesh_trace_kernel -- launched after every chunk
// and a no-op unless the chunk's overflow flag is set -- traces the chunk's items instead.  Nothing here waits for the
// host, so a render stays a sequence of launches on one stream.
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdio>
#include <cstdlib>

#include "../../include/srt.h"
#include "srt_launch.h"
#include "srt_path.h"

// The walkers want wavefronts: with the whole 36-entry traversal stack in LDS (9 KB per wavefront) a CU holds 4 per SIMD,
// and 3 instead of 4 costs 13 % (10,000-sphere frame: 202 against 178 ms).  So only the near end of a lane's stack lives
// in LDS and the far end, which few traversals reach, in a column of device memory (LaneStack<STRIDE, NEAR>).  Measured,
// same frame: 36 entries / 4 wavefronts 177.2 ms; 28 / 5 170.9; 24 / 6 162.6; 20 / 6 162.8; 20 / 7 158.4; 20 / 8 157.4;
// 16 / 6 161.4; 16 / 8 156.3 (8 is what a SIMD takes; the kernel then has 64 registers and spills a dozen).
#ifndef SRT_WAVEFRONT_WALK_WAVES
#define SRT_WAVEFRONT_WALK_WAVES 8
#endif
#ifndef SRT_WAVEFRONT_NEAR_STACK
#define SRT_WAVEFRONT_NEAR_STACK 16   // entries of a walker's traversal stack that live in LDS; the rest in device memory
#endif

namespace srt {

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr uint32_t WF_NONE = 0xFFFFFFFFu;   // val.w: the node's value is final
constexpr uint32_t WF_LIT = 0xFFFFFFFEu;    // val.w: the node was hit and waits for its shading
constexpr uint32_t kDraw = 1024;            // queries a walking wavefront draws at a time while there are plenty
constexpr int kGridBlocks = 256 * 3;        // grid of wf_shade (the per-node kernels stride over the level): 3 blocks per CU, what its registers
                                            // allow (10,000-sphere frame: 1024 blocks 181.0 ms, 896 186.8, 768 177.8, 640 186.0, 512 189.7)

struct WfControl {
    uint32_t level_base[WAVEFRONT_LEVELS + 1];
    uint32_t level_count[WAVEFRONT_LEVELS + 1];
    uint32_t shadow_count[WAVEFRONT_LEVELS];
    uint32_t cursor[2 * WAVEFRONT_LEVELS];   // work cursors of the walkers: [2 * level + shadow]
    uint32_t overflow;                        // non-zero: the chunk is given up
    uint32_t pad_;
    unsigned long long tally[8];              // rays_path, rays_shadow, hit_nodes, box, sphere, tri, rect tests, unasked
    unsigned long long twins;                 // closest-hit queries answered by the query of the node before them (see wf_shade)
};
static_assert(sizeof(WfControl) <= WAVEFRONT_CONTROL_BYTES, "the control block outgrew its reservation");

// The pool: one array of 16-byte words per field, `capacity` entries each.
//   ray0 (o.xyz d.x)  ray1 (d.y d.z depth twin)  id (slot pixel sample node)  hit (t prim u v)
//   pn0 (P.xyz material | depth after the heuristic << 24)  pn1 (N.xyz asked-lights mask)
//   sh0 / sh1 (t prim) of the shadow queries of lights 0,1 / 2,3
//   fr0 (emission.xyz w_refl)  fr1 (lightColor.xyz material)  val (value.xyz, WF_NONE | WF_LIT | index of the first child)
//   sq0 (P.xyz d.x)  sq1 (d.y d.z node light): the shadow queries of the level being worked on
struct WfPool {
    q4 *ray0, *ray1, *id, *hit, *pn0, *pn1, *sh0, *sh1, *fr0, *fr1, *val, *sq0, *sq1;
    uint32_t capacity;
    WfControl* ctrl;
    uint32_t* far_stacks;    // walkers whose LDS holds only the near end of the traversal stack: the rest, per wavefront of the grid
};
constexpr size_t kArrays = 13;
constexpr size_t kFarStackBytes = (size_t)WAVEFRONT_MAX_WALK_BLOCKS * kWaves * (size_t)(STACK_DEPTH - SRT_WAVEFRONT_NEAR_STACK) * 64u * sizeof(uint32_t);
constexpr size_t kHeaderBytes = (size_t)WAVEFRONT_MAX_CHUNKS * 4u + WAVEFRONT_CONTROL_BYTES;

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ q4 mkq(float x, float y, float z, float w) { q4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
// Loads and stores of the pool.  -DSRT_WAVEFRONT_NONTEMPORAL (experiments) marks them non-temporal; measured slower
// (wf_combine reads what the launch before it wrote), so the default is plain accesses.
typedef float pool_word __attribute__((ext_vector_type(4)));
__device__ __forceinline__ q4 ld(const q4* p)
{
#ifdef SRT_WAVEFRONT_NONTEMPORAL
    const pool_word v = __builtin_nontemporal_load(reinterpret_cast<const pool_word*>(p));
    return mkq(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ void st(q4* p, q4 v)
{
#ifdef SRT_WAVEFRONT_NONTEMPORAL
    pool_word t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<pool_word*>(p));
#else
    *p = v;
#endif
}

// Entries of an array that is filled through one global counter.  An atomic on one address costs the whole device about
// 9 ns, so a wavefront does not take entries one request at a time: it reserves `grab` entries at once and hands them to
// its lanes from there.  What is left of a reservation when the next one is made, or when the kernel ends, is marked as
// holes (the readers of the array skip them).
struct WaveRange { uint32_t next = 0u, end = 0u, used = 0u, calls = 0u; };   // wave-uniform; used: entries handed out, calls: requests served
// How much to reserve next: what the inputs the wavefront still has to go through are expected to need, judged by what
// its inputs needed so far (at least one full request, at most `grab`), so that its last reservation leaves little over.
__device__ __forceinline__ uint32_t grab_for(uint32_t grab, uint32_t inputs_done, uint32_t outputs_done, uint32_t inputs_left)
{
    if (inputs_done < 256u) return grab;
    const uint32_t expect = (uint32_t)(((unsigned long long)outputs_done * inputs_left) / inputs_done);
    const uint32_t want = (expect + (expect >> 3) + 63u) & ~63u;
    return want < grab ? want : grab;
}

// `count` entries (0..4) for every lane; returns the lane's first entry.  All lanes of the wavefront must call.
template <class HOLE>
__device__ __forceinline__ uint32_t wave_reserve(WaveRange& r, uint32_t* counter, uint32_t count, uint32_t grab, uint32_t capacity, HOLE&& mark_hole)
{
    const unsigned long long m1 = __ballot(count >= 1u), m2 = __ballot(count >= 2u), m3 = __ballot(count >= 3u), m4 = __ballot(count >= 4u);
    const uint32_t total = (uint32_t)(__popcll(m1) + __popcll(m2) + __popcll(m3) + __popcll(m4));
    r.calls++;
    r.used += total;
    if (total == 0u) return 0u;
    const uint32_t lane = threadIdx.x & 63u;
    if (r.end - r.next < total) {
        for (uint32_t j = r.next + lane; j < r.end; j += 64u)
            if (j < capacity) mark_hole(j);
        const uint32_t g = grab > total ? grab : total;
        uint32_t b = 0u;
        if (lane == 0u) b = atomicAdd(counter, g);
        b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
        r.next = b; r.end = b + g;
    }
    const uint32_t mine = r.next + lanes_below(m1) + lanes_below(m2) + lanes_below(m3) + lanes_below(m4);
    r.next += total;
    return mine;
}
template <class HOLE>
__device__ __forceinline__ void wave_release(WaveRange& r, uint32_t capacity, HOLE&& mark_hole)
{
    for (uint32_t j = r.next + (threadIdx.x & 63u); j < r.end; j += 64u)
        if (j < capacity) mark_hole(j);
    r.next = r.end;
}
// Entries to reserve at a time: a quarter of what a wavefront of the grid is expected to need in all.
__device__ __forceinline__ uint32_t grab_size(uint32_t n_in, uint32_t out_per_in)
{
    const uint32_t waves = gridDim.x * (uint32_t)kWaves;
    const uint32_t expect = (uint32_t)(((unsigned long long)n_in * out_per_in) / waves);
    const uint32_t g = ((expect / 4u) + 63u) & ~63u;
    return g < 128u ? 128u : (g > 4096u ? 4096u : g);
}

// A frame that lives in a register for the length of one call.
struct LocalFrames {
    Frame* f;
    __device__ __forceinline__ Frame load(int32_t) const { return *f; }
    __device__ __forceinline__ void store(int32_t, const Frame& v) const { *f = v; }
    __device__ __forceinline__ void store_acc(int32_t, f3 acc) const { f->acc = acc; }
};

__device__ __forceinline__ void fill_unit_table(float* unit)
{
    if (threadIdx.x < (uint32_t)UNIT_TABLE) unit[threadIdx.x] = (float)((int)threadIdx.x - 100) / 100.0f;
    __syncthreads();
}

// Item i of the chunk is node i of level 0; the padding pixels of edge tiles become dead nodes (depth 0).
__global__ void __launch_bounds__(kBlock) wf_generate(RenderArgs ra, WfPool pool, uint32_t item0, uint32_t item1)
{
    __shared__ float unit[UNIT_TABLE];
    fill_unit_table(unit);
    if (blockIdx.x == 0u && threadIdx.x == 0u) pool.ctrl->level_count[0] = item1 - item0;
    for (uint32_t item = item0 + blockIdx.x * (uint32_t)kBlock + threadIdx.x; item < item1; item += gridDim.x * (uint32_t)kBlock) {
        const uint32_t at = item - item0;
        Lane ln;
        if (!item_start(ln, ra, item, unit)) { st(&pool.ray1[at], mkq(0.0f, 0.0f, 0.0f, 0.0f)); continue; }
        st(&pool.ray0[at], mkq(ln.qo.x, ln.qo.y, ln.qo.z, ln.qd.x));
        st(&pool.ray1[at], mkq(ln.qd.y, ln.qd.z, bitsf((uint32_t)ln.depth), 0.0f));
        st(&pool.id[at], mkq(bitsf(ln.slot), bitsf(ln.key.pixel), bitsf(ln.key.sample), bitsf(ln.key.node)));
    }
}

// The queries of one level: closest-hit queries of its nodes, or the shadow queries the hit kernel left.
template <bool SHADOW, bool COUNT>
__global__ void __launch_bounds__(kBlock, SRT_WAVEFRONT_WALK_WAVES) wf_walk(SceneView sc, WfPool pool, uint32_t level, int rounds, int steps, int refill)
{
    constexpr int kNear = SRT_WAVEFRONT_NEAR_STACK;
    __shared__ uint32_t lds_stack[kWaves * kNear * 64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* my_stack = &lds_stack[wave * (kNear * 64) + lane];
    // the deep end of the stack, if the LDS does not hold all of it: a column per lane of every wavefront of the grid
    uint32_t* far_stack = kNear < STACK_DEPTH ? pool.far_stacks + ((size_t)(blockIdx.x * (uint32_t)kWaves + wave) * (size_t)(STACK_DEPTH - kNear)) * 64u + lane : nullptr;
    WfControl* const ctrl = pool.ctrl;
    // A chunk that outgrew the pool is given up as soon as a kernel notices: entries that could not be written hold
    // whatever the memory held before, so no later kernel of the chunk may look at the arrays.
    if (ctrl->overflow != 0u) return;
    uint32_t n = SHADOW ? ctrl->shadow_count[level] : ctrl->level_count[level];
    const uint32_t base = SHADOW ? 0u : ctrl->level_base[level];
    if (base >= pool.capacity) return;
    if (n > pool.capacity - base) n = pool.capacity - base;   // stay inside the arrays whatever the counters say
    if ((blockIdx.x * (uint32_t)kWaves + wave) * 64u >= n) return;   // more wavefronts than draws: no need to touch the cursor
    uint32_t* const cursor = &ctrl->cursor[2u * level + (SHADOW ? 1u : 0u)];
    const float t_min = SHADOW ? 0.001f : 0.01f;

    Trav tr;
    tr.d = mk3(0.0f, 0.0f, 0.0f);
    tr.best.t = INFINITY; tr.best.prim = NO_PRIM; tr.best.u = tr.best.v = 0.0f; tr.best.order = 0u;
    tr.cur = TRAV_NONE; tr.sp = 0; tr.pad = 0.0f;
    f3 o = mk3(0.0f, 0.0f, 0.0f), inv = mk3(0.0f, 0.0f, 0.0f);
    float a = 0.0f;
    uint32_t q = WF_NONE;        // the query this lane works on
    uint32_t tag0 = 0u, tag1 = 0u;   // shadow queries: node and light
    Counters cnt;
    cnt.box_tests = cnt.sphere_tests = cnt.tri_tests = cnt.rect_tests = 0u;
    uint32_t next = 0u, end = 0u;
    bool dry = false;
    // Queries per draw from the cursor: kDraw while there is plenty left, then an ever smaller share of what is left (down
    // to one per lane), so that no wavefront ends the launch with a long range of its own.
    const uint32_t guide = gridDim.x * (uint32_t)kWaves * 2u;
    uint32_t draw = n / guide >= kDraw ? kDraw : (n / guide < 64u ? 64u : ((n / guide) & ~63u));

    for (;;) {
        unsigned long long idle = __ballot(q == WF_NONE);
        if (!dry && (int)__popcll(idle) >= refill) {
            // (until fewer than `refill` lanes are idle: entries can be holes, and a pass of this loop costs the same for one lane as for 64)
            while ((int)__popcll(idle) >= refill) {
                if (next >= end) {
                    uint32_t b = 0u;
                    if (lane == 0u) b = atomicAdd(cursor, draw);
                    b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                    if (b >= n) { dry = true; break; }
                    next = b;
                    end = b + draw < n ? b + draw : n;
                    const uint32_t share = ((n - end) / guide) & ~63u;
                    draw = share >= kDraw ? kDraw : (share < 64u ? 64u : share);
                }
                const uint32_t avail = end - next, want = (uint32_t)__popcll(idle), rank = lanes_below(idle);
                if (q == WF_NONE && rank < avail) {
                    const uint32_t mine = next + rank;
                    q4 r0, r1;
                    if (SHADOW) { r0 = ld(&pool.sq0[mine]); r1 = ld(&pool.sq1[mine]); tag0 = fbits(r1.z); tag1 = fbits(r1.w); }
                    else { r0 = ld(&pool.ray0[base + mine]); r1 = ld(&pool.ray1[base + mine]); }
                    // left over from a reservation / a dead node; a query whose node or light is out of range cannot be one of ours
                    // (a twin -- same ray as the node before it -- is answered by that node's query; counting builds walk it all the same, so that
                    // their test counts stay those of the one-kernel tracer)
                    const bool hole = SHADOW ? (tag1 >= WAVEFRONT_LIGHTS || tag0 >= pool.capacity) : (fbits(r1.z) == 0u || (!COUNT && fbits(r1.w) != 0u));
                    if (!hole) {
                        q = mine;
                        o = mk3(r0.x, r0.y, r0.z);
                        tr.d = mk3(r0.w, r1.x, r1.y);
                        a = lensq3(tr.d);
                        inv = mk3(1.0f / tr.d.x, 1.0f / tr.d.y, 1.0f / tr.d.z);   // CustomRay.dirFrac (RaytracingClasses.cs:16-18)
                        trav_begin<false, COUNT, 64>(sc, o, t_min, tr, my_stack, cnt);
                    }
                }
                next += want < avail ? want : avail;
                idle = __ballot(q == WF_NONE);
            }
        }
        const bool active = q != WF_NONE;
        if (__ballot(active) == 0ull) { if (dry) break; else continue; }
        // queries that start far outside the scene test widened boxes; a wavefront without one keeps the plain walk
        const bool padded = __ballot(active && tr.pad != 0.0f) != 0ull;
        if (active && tr.cur != TRAV_NONE) {
            if (padded) trav_run_with<false, COUNT, 64, true, kNear>(sc, o, a, inv, t_min, tr, my_stack, cnt, rounds, steps, far_stack);
            else trav_run_with<false, COUNT, 64, false, kNear>(sc, o, a, inv, t_min, tr, my_stack, cnt, rounds, steps, far_stack);
        }
        if (active && tr.cur == TRAV_NONE) {
            if (SHADOW) {
                float2* dst = reinterpret_cast<float2*>((tag1 < 2u ? pool.sh0 : pool.sh1) + tag0) + (tag1 & 1u);
                typedef float pool_pair __attribute__((ext_vector_type(2)));
                pool_pair pr;
                pr.x = tr.best.t; pr.y = bitsf(tr.best.prim);
                *reinterpret_cast<pool_pair*>(dst) = pr;
            } else {
                st(&pool.hit[base + q], mkq(tr.best.t, bitsf(tr.best.prim), tr.best.u, tr.best.v));
            }
            q = WF_NONE;
        }
    }
    if constexpr (COUNT) {
        unsigned long long c0 = wave_sum(cnt.box_tests), c1 = wave_sum(cnt.sphere_tests), c2 = wave_sum(cnt.tri_tests), c3 = wave_sum(cnt.rect_tests);
        if (lane == 0u) {
            atomicAdd(&ctrl->tally[3], c0);
            atomicAdd(&ctrl->tally[4], c1);
            atomicAdd(&ctrl->tally[5], c2);
            atomicAdd(&ctrl->tally[6], c3);
        }
    }
}

// Returns whether the node is a twin: its ray is the ray of node g - 1.
__device__ __forceinline__ bool load_node(const WfPool& pool, uint32_t g, const RenderArgs& ra, Lane& ln)
{
    const q4 a = ld(&pool.ray0[g]), b = ld(&pool.ray1[g]), c = ld(&pool.id[g]);
    ln.qo = mk3(a.x, a.y, a.z);
    ln.qd = mk3(a.w, b.x, b.y);
    ln.qtmin = 0.01f;
    ln.depth = (int32_t)fbits(b.z);
    ln.slot = fbits(c.x);
    ln.key.seed = ra.seed; ln.key.pixel = fbits(c.y); ln.key.sample = fbits(c.z); ln.key.node = fbits(c.w);
    ln.level = 0;
    return fbits(b.w) != 0u;
}

// Closest-hit queries that are answered (lane_on_closest), and the shadow queries of the hits.
#ifndef SRT_WF_HIT_WAVES
#define SRT_WF_HIT_WAVES 4
#endif
__global__ void __launch_bounds__(kBlock, SRT_WF_HIT_WAVES) wf_hit(SceneView sc, RenderArgs ra, WfPool pool, uint32_t level)
{
    __shared__ float unit[UNIT_TABLE];
    fill_unit_table(unit);
    WfControl* const ctrl = pool.ctrl;
    if (ctrl->overflow != 0u) return;      // the chunk is given up (uniform over the launch: set by an earlier kernel)
    const uint32_t base = ctrl->level_base[level];
    if (base >= pool.capacity) return;
    uint32_t n = ctrl->level_count[level];
    if (n > pool.capacity - base) n = pool.capacity - base;
    // where the next level starts (the shade kernel of this level allocates there)
    if (blockIdx.x == 0u && threadIdx.x == 0u) ctrl->level_base[level + 1u] = base + n;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t rays_path = 0u, hit_nodes = 0u, unasked = 0u, twins = 0u;
    WaveRange queue;
    const uint32_t grab = grab_size(n, sc.n_lights);
    auto hole = [&](uint32_t j) { st(&pool.sq1[j], mkq(0.0f, 0.0f, 0.0f, bitsf(WF_NONE))); };
    for (uint32_t first = blockIdx.x * (uint32_t)kBlock; first < n; first += gridDim.x * (uint32_t)kBlock) {
        const uint32_t i = first + threadIdx.x, g = base + i;
        const bool valid = i < n;
        Lane ln;
        ln.rays_path = ln.rays_shadow = ln.hit_nodes = 0u;
        int p = LANE_IDLE;
        if (valid) {
            const bool twin = load_node(pool, g, ra, ln);
            twins += twin ? 1u : 0u;
            const q4 hw = ld(&pool.hit[twin ? g - 1u : g]);   // a twin is never the first child of its parent
            Hit h;
            h.t = hw.x; h.prim = fbits(hw.y); h.u = hw.z; h.v = hw.w; h.order = 0u;
            Frame unused;
            LocalFrames frames{&unused};
            f3 ret;
            // LANE_RETURN (a miss; a dead node: depth 0 returns black and counts nothing) or LANE_LIGHT
            p = lane_on_closest(ln, h, ret, sc, ra, frames, unit);
            rays_path += ln.rays_path; hit_nodes += ln.hit_nodes;
            if (p == LANE_RETURN) st(&pool.val[g], mkq(ln.former.x, ln.former.y, ln.former.z, bitsf(WF_NONE)));
        }
        uint32_t asked = 0u;
        uint32_t aux[4] = {0u, 0u, 0u, 0u};
        if (p == LANE_LIGHT) {
            rng_call(ln.key, CALL_AUX, aux);
            for (uint32_t li = 0; li < sc.n_lights; li++)
                if (lane_shadow_matters(ln, sc, li)) asked |= 1u << li;
            unasked += sc.n_lights - (uint32_t)__popc(asked);
            st(&pool.pn0[g], mkq(ln.P.x, ln.P.y, ln.P.z, bitsf(ln.mat | ((uint32_t)ln.depth << 24))));
            st(&pool.pn1[g], mkq(ln.N.x, ln.N.y, ln.N.z, bitsf(asked)));
            st(&pool.val[g], mkq(0.0f, 0.0f, 0.0f, bitsf(WF_LIT)));
        }
        const uint32_t count = (uint32_t)__popc(asked);
        const uint32_t calls_left = (n - 1u - first) / (gridDim.x * (uint32_t)kBlock) + 1u;
        uint32_t at = wave_reserve(queue, &ctrl->shadow_count[level], count, grab_for(grab, queue.calls * 64u, queue.used, calls_left * 64u), pool.capacity, hole);
        if (count == 0u) continue;
        if (at > pool.capacity || count > pool.capacity - at) { ctrl->overflow = 1u; continue; }
        for (uint32_t li = 0; li < sc.n_lights; li++) {
            if (((asked >> li) & 1u) == 0u) continue;
            f3 o, d;
            lane_shadow_query(ln, sc, li, aux, unit, o, d);
            st(&pool.sq0[at], mkq(o.x, o.y, o.z, d.x));
            st(&pool.sq1[at], mkq(d.y, d.z, bitsf(g), bitsf(li)));
            at++;
        }
    }
    wave_release(queue, pool.capacity, hole);
    const unsigned long long v0 = wave_sum(rays_path), v2 = wave_sum(hit_nodes), v7 = wave_sum(unasked), vt = wave_sum(twins);
    if (lane == 0u && (v0 | v2 | v7) != 0ull) {
        atomicAdd(&ctrl->tally[0], v0);
        atomicAdd(&ctrl->tally[2], v2);
        atomicAdd(&ctrl->tally[7], v7);
        if (vt != 0ull) atomicAdd(&ctrl->twins, vt);
    }
}

// The lights of the hit nodes, their shading, and their children: the nodes of the next level.
#ifndef SRT_WF_SHADE_WAVES
#define SRT_WF_SHADE_WAVES 3
#endif
__global__ void __launch_bounds__(kBlock, SRT_WF_SHADE_WAVES) wf_shade(SceneView sc, RenderArgs ra, WfPool pool, uint32_t level, uint32_t share_queries)
{
    __shared__ float unit[UNIT_TABLE];
    fill_unit_table(unit);
    WfControl* const ctrl = pool.ctrl;
    if (ctrl->overflow != 0u) return;      // the chunk is given up
    const uint32_t base = ctrl->level_base[level];
    if (base >= pool.capacity) return;
    uint32_t n = ctrl->level_count[level];
    if (n > pool.capacity - base) n = pool.capacity - base;
    const uint32_t next_base = ctrl->level_base[level + 1u];
    if (next_base > pool.capacity) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t rays_shadow = 0u;
    WaveRange nodes;
    const uint32_t grab = grab_size(n, 2u);
    const uint32_t room = pool.capacity - next_base;      // entries of the pool the next level can take
    auto hole = [&](uint32_t j) { st(&pool.ray1[next_base + j], mkq(0.0f, 0.0f, 0.0f, 0.0f)); };   // a dead node: depth 0
    for (uint32_t first = blockIdx.x * (uint32_t)kBlock; first < n; first += gridDim.x * (uint32_t)kBlock) {
        const uint32_t i = first + threadIdx.x, g = base + i;
        bool lit = false;
        if (i < n) lit = fbits(ld(&pool.val[g]).w) == WF_LIT;
        Lane ln;
        ln.rays_path = ln.rays_shadow = ln.hit_nodes = 0u;
        Frame f;
        int p = LANE_IDLE;
        uint32_t parent = 0u, children = 0u;
        float metal = 0.0f;
        if (lit) {
            load_node(pool, g, ra, ln);
            const q4 a = ld(&pool.pn0[g]), b = ld(&pool.pn1[g]);
            ln.P = mk3(a.x, a.y, a.z); ln.N = mk3(b.x, b.y, b.z);
            ln.mat = fbits(a.w) & 0x00FFFFFFu;
            ln.depth = (int32_t)(fbits(a.w) >> 24);
            ln.former = mk3(0.0f, 0.0f, 0.0f);
            const uint32_t asked = fbits(b.w);
            parent = ln.key.node;
            uint32_t aux[4];
            rng_call(ln.key, CALL_AUX, aux);
            q4 s01 = mkq(0.0f, 0.0f, 0.0f, 0.0f), s23 = s01;
            if (asked & 3u) s01 = ld(&pool.sh0[g]);
            if (asked & 12u) s23 = ld(&pool.sh1[g]);
            for (uint32_t li = 0; li < sc.n_lights; li++) {
                if ((asked >> li) & 1u) {
                    f3 o, d;
                    lane_shadow_query(ln, sc, li, aux, unit, o, d);
                    const q4 s = li < 2u ? s01 : s23;
                    Hit h;
                    h.t = (li & 1u) ? s.z : s.x; h.prim = fbits((li & 1u) ? s.w : s.y); h.u = h.v = 0.0f; h.order = 0u;
                    lane_on_shadow(ln, h, o, d, sc, li);
                } else {
                    lane_skip_light(ln);
                }
            }
            rays_shadow += ln.rays_shadow;
            LocalFrames frames{&f};
            f3 ret;
            p = lane_shade(ln, ln.former, ret, sc, ra, frames, aux, unit);   // LANE_RETURN or LANE_TRACE (MODE_COLOR)
            if (p == LANE_RETURN) st(&pool.val[g], mkq(ln.former.x, ln.former.y, ln.former.z, bitsf(WF_NONE)));
            else {
                metal = sc.materials[2u * ln.mat].w;
                children = 1u + (metal != 0.0f ? 1u : 0u) + (metal != 1.0f ? 1u : 0u);
            }
        }
        const uint32_t calls_left = (n - 1u - first) / (gridDim.x * (uint32_t)kBlock) + 1u;
        uint32_t at = wave_reserve(nodes, &ctrl->level_count[level + 1u], children, grab_for(grab, nodes.calls * 64u, nodes.used, calls_left * 64u), room, hole);
        if (children == 0u) continue;
        if (at > room || children > room - at) {
            ctrl->overflow = 1u;
            st(&pool.val[g], mkq(0.0f, 0.0f, 0.0f, bitsf(WF_NONE)));
            continue;
        }
        at += next_base;
        st(&pool.fr0[g], mkq(f.acc.x, f.acc.y, f.acc.z, f.w_refl));
        st(&pool.fr1[g], mkq(f.light.x, f.light.y, f.light.z, bitsf(ln.mat)));   // the material rides along: wf_combine need not read pn0
        st(&pool.val[g], mkq(0.0f, 0.0f, 0.0f, bitsf(at)));
        const float depth_word = bitsf((uint32_t)ln.depth);   // lane_shade left the children's depth here
        // child 0: the diffuse ray, which lane_shade set up as the lane's next query
        st(&pool.ray0[at], mkq(ln.qo.x, ln.qo.y, ln.qo.z, ln.qd.x));
        st(&pool.ray1[at], mkq(ln.qd.y, ln.qd.z, depth_word, 0.0f));
        st(&pool.id[at], mkq(0.0f, bitsf(ln.key.pixel), bitsf(ln.key.sample), bitsf(parent * 4u + 1u)));
        at++;
        // children 1 (metal) and 2 (reflection) are traced along the same ray (RaytracingOperations.cs:160,163: both calls get
        // CustomRay(record.point, reflectedDirection)), so world.hit answers both alike: where a node has both, the second is
        // marked as the twin of the first and takes over its answer instead of walking the tree again.  Everything after the
        // query differs between them (the random numbers are keyed by the node), so both stay nodes of their own.
        for (uint32_t k = 1u; k < 3u; k++) {
            if (k == 1u ? metal == 0.0f : metal == 1.0f) continue;
            const bool twin = k == 2u && metal != 0.0f && share_queries != 0u;
            st(&pool.ray0[at], mkq(f.point.x, f.point.y, f.point.z, f.refl.x));
            st(&pool.ray1[at], mkq(f.refl.y, f.refl.z, depth_word, bitsf(twin ? 1u : 0u)));
            st(&pool.id[at], mkq(0.0f, bitsf(ln.key.pixel), bitsf(ln.key.sample), bitsf(parent * 4u + k + 1u)));
            at++;
        }
    }
    wave_release(nodes, room, hole);
    const unsigned long long v1 = wave_sum(rays_shadow);
    if (lane == 0u && v1 != 0ull) atomicAdd(&ctrl->tally[1], v1);
}

// The values of the nodes of `level` that have children, from the children's values (lane_return adds them in the
// order of the recursion and clamps).
__global__ void __launch_bounds__(kBlock) wf_combine(SceneView sc, RenderArgs ra, WfPool pool, uint32_t level)
{
    WfControl* const ctrl = pool.ctrl;
    if (ctrl->overflow != 0u) return;      // the chunk is given up
    const uint32_t base = ctrl->level_base[level];
    if (base >= pool.capacity) return;
    uint32_t n = ctrl->level_count[level];
    if (n > pool.capacity - base) n = pool.capacity - base;
    for (uint32_t i = blockIdx.x * (uint32_t)kBlock + threadIdx.x; i < n; i += gridDim.x * (uint32_t)kBlock) {
        const uint32_t g = base + i;
        uint32_t child = fbits(ld(&pool.val[g]).w);
        if (child == WF_NONE || child == WF_LIT) continue;
        const q4 a = ld(&pool.fr0[g]), b = ld(&pool.fr1[g]);
        Frame f;
        f.acc = mk3(a.x, a.y, a.z); f.w_refl = a.w;
        f.light = mk3(b.x, b.y, b.z);
        f.point = f.refl = mk3(0.0f, 0.0f, 0.0f);
        f.meta = fbits(b.w) & 0x00FFFFFFu;
        LocalFrames frames{&f};
        Lane ln;
        ln.level = 1;
        ln.key.node = 1u;      // "child 0 of the frame just returned"; lane_return moves it on to the next child
        f3 ret = mk3(0.0f, 0.0f, 0.0f);
        for (int k = 0; k < 3; k++) {
            if (child >= pool.capacity) break;   // an overflowing chunk
            const q4 v = ld(&pool.val[child++]);
            ln.former = mk3(v.x, v.y, v.z);
            if (lane_return(ln, ret, sc, ra, frames) != LANE_TRACE) break;
            ln.level = 1;
        }
        st(&pool.val[g], mkq(ret.x, ret.y, ret.z, bitsf(WF_NONE)));
    }
}

__global__ void __launch_bounds__(kBlock) wf_finish(WfPool pool, f3* __restrict__ samples, uint32_t n_slots, unsigned long long* __restrict__ stats, uint32_t* __restrict__ given_up)
{
    WfControl* const ctrl = pool.ctrl;
    if (ctrl->overflow != 0u) {
        if (blockIdx.x == 0u && threadIdx.x == 0u) { *given_up = 1u; atomicAdd(&stats[kStatsWavefrontGivenUp], 1ull); }
        return;
    }
    uint32_t n = ctrl->level_count[0];
    if (n > pool.capacity) n = pool.capacity;
    for (uint32_t i = blockIdx.x * (uint32_t)kBlock + threadIdx.x; i < n; i += gridDim.x * (uint32_t)kBlock) {
        if (fbits(ld(&pool.ray1[i]).z) == 0u) continue;   // a padding pixel
        const q4 v = ld(&pool.val[i]);
        const uint32_t slot = fbits(ld(&pool.id[i]).x);
        if (slot < n_slots) samples[slot] = mk3(v.x, v.y, v.z);
    }
    if (blockIdx.x == 0u && threadIdx.x == 0u) {
        for (int k = 0; k < 7; k++) atomicAdd(&stats[k], ctrl->tally[k]);
        atomicAdd(&stats[kStatsUnaskedWord], ctrl->tally[7]);
        unsigned long long nodes = 0ull;
        for (uint32_t l = 0; l <= WAVEFRONT_LEVELS; l++) nodes += ctrl->level_count[l];
        atomicAdd(&stats[kStatsWavefrontNodes], nodes);
        // the shadow queries of a level share the pool's capacity too: the largest queue of the chunk
        uint32_t queue = 0u;
        for (uint32_t l = 0; l < WAVEFRONT_LEVELS; l++) queue = ctrl->shadow_count[l] > queue ? ctrl->shadow_count[l] : queue;
        atomicAdd(&stats[kStatsWavefrontQueue], (unsigned long long)queue);
        atomicAdd(&stats[kStatsWavefrontTwins], ctrl->twins);
        atomicAdd(&stats[kStatsWavefrontItems], (unsigned long long)n);
    }
}

WfPool carve(void* mem, size_t bytes)
{
    WfPool p;
    char* at = reinterpret_cast<char*>(mem);
    p.ctrl = reinterpret_cast<WfControl*>(at + WAVEFRONT_MAX_CHUNKS * 4u);
    p.far_stacks = nullptr;
    at += kHeaderBytes;
    const size_t cap = (bytes - kHeaderBytes) / (kArrays * sizeof(q4));
    p.capacity = cap > 0x7FFFFFF0ull ? 0x7FFFFFF0u : (uint32_t)cap;
    q4** field[kArrays] = {&p.ray0, &p.ray1, &p.id, &p.hit, &p.pn0, &p.pn1, &p.sh0, &p.sh1, &p.fr0, &p.fr1, &p.val, &p.sq0, &p.sq1};
    for (size_t k = 0; k < kArrays; k++) {
        *field[k] = reinterpret_cast<q4*>(at);
        at += (size_t)p.capacity * sizeof(q4);
    }
    return p;
}

}  // namespace

bool wavefront_serves(const SceneView& sc, const RenderArgs& ra, bool exact)
{
    const bool off = std::getenv("SRT_NO_WAVEFRONT") != nullptr;   // read per launch: a process can switch between the tracers
    return !off && !exact && sc.has_bvh != 0u && (uint32_t)ra.mode == MODE_COLOR && sc.n_lights >= 1u && sc.n_lights <= WAVEFRONT_LIGHTS &&
           ra.max_depth >= 1 && ra.max_depth <= (int32_t)WAVEFRONT_LEVELS;
}

size_t wavefront_node_bytes() { return kArrays * sizeof(q4); }
size_t wavefront_header_bytes() { return kHeaderBytes; }
size_t wavefront_far_stack_bytes() { return kFarStackBytes; }

uint32_t wavefront_capacity(size_t bytes)
{
    if (bytes < kHeaderBytes + kArrays * sizeof(q4) * 4096u) return 0u;
    return carve(nullptr, bytes).capacity;
}

const uint32_t* wavefront_chunk_flag(void* pool_mem, uint32_t chunk)
{
    return reinterpret_cast<const uint32_t*>(pool_mem) + chunk;
}

hipError_t launch_wavefront_chunk(const SceneView& sc, const RenderArgs& ra, bool counted, float* d_samples, unsigned long long* d_stats,
                                  void* pool_mem, size_t pool_bytes, void* far_stacks, uint32_t chunk, uint32_t item0, uint32_t item1, hipStream_t stream)
{
    if (item1 <= item0) return hipSuccess;
    if (chunk >= WAVEFRONT_MAX_CHUNKS || (kFarStackBytes != 0 && far_stacks == nullptr)) return hipErrorInvalidValue;
    WfPool pool = carve(pool_mem, pool_bytes);
    pool.far_stacks = reinterpret_cast<uint32_t*>(far_stacks);
    if (pool.capacity < item1 - item0) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(pool.ctrl, 0, sizeof(WfControl), stream);
    if (e != hipSuccess) return e;
    static const int rounds = std::getenv("SRT_WF_ROUNDS") ? std::atoi(std::getenv("SRT_WF_ROUNDS")) : 4;
    static const int steps = std::getenv("SRT_WF_STEPS") ? std::atoi(std::getenv("SRT_WF_STEPS")) : 4;
    static const int refill_knob = std::getenv("SRT_WF_REFILL") ? std::atoi(std::getenv("SRT_WF_REFILL")) : 8;
    const int refill = refill_knob < 1 ? 1 : (refill_knob > 64 ? 64 : refill_knob);
    static uint32_t walk_blocks = 0u;
    if (walk_blocks == 0u) {
        int per_cu = 0, dev = 0, cus = 0;
        walk_blocks = 1024u;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wf_walk<false, false>, kBlock, 0) == hipSuccess && per_cu > 0 && cus > 0)
            walk_blocks = (uint32_t)per_cu * (uint32_t)cus;
        if (const char* knob = std::getenv("SRT_WF_WALK_BLOCKS")) walk_blocks = std::atoi(knob) > 0 ? (uint32_t)std::atoi(knob) : walk_blocks;   // experiments
        if (walk_blocks > WAVEFRONT_MAX_WALK_BLOCKS) walk_blocks = WAVEFRONT_MAX_WALK_BLOCKS;   // the far stacks are sized for that many
        if (std::getenv("SRT_DEBUG_LAUNCH")) std::fprintf(stderr, "srt: wavefront walkers: %u resident blocks\n", walk_blocks);
    }
    // Grids of the per-node kernels, in blocks per CU what their registers allow: wf_shade 3, wf_hit 4, the small ones 6
    // (10,000-sphere frame: all on 768 blocks 147.8 ms, wf_hit on 1024 146.0, the small ones on 1536 as well 145.7).
    static const int grid_knob = std::getenv("SRT_WF_GRID") ? std::atoi(std::getenv("SRT_WF_GRID")) : 2 * kGridBlocks;       // experiments
    static const int hit_knob = std::getenv("SRT_WF_HIT_GRID") ? std::atoi(std::getenv("SRT_WF_HIT_GRID")) : 256 * SRT_WF_HIT_WAVES;
    static const int shade_knob = std::getenv("SRT_WF_SHADE_GRID") ? std::atoi(std::getenv("SRT_WF_SHADE_GRID")) : kGridBlocks;
    const dim3 block(kBlock), grid(grid_knob > 0 ? grid_knob : 2 * kGridBlocks), walkers(walk_blocks);
    const dim3 hit_grid(hit_knob > 0 ? hit_knob : kGridBlocks), shade_grid(shade_knob > 0 ? shade_knob : kGridBlocks);
    const uint32_t share = std::getenv("SRT_NO_SHARED_QUERIES") ? 0u : 1u;   // measurements: every GetRayColor call walks the tree itself
    const uint32_t n_items = item1 - item0;
    const uint32_t gen_blocks = (n_items + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(wf_generate, dim3(gen_blocks < (uint32_t)kGridBlocks ? gen_blocks : (uint32_t)kGridBlocks), block, 0, stream, ra, pool, item0, item1);
    const uint32_t levels = (uint32_t)ra.max_depth;
    for (uint32_t l = 0; l < levels; l++) {
        if (counted) hipLaunchKernelGGL((wf_walk<false, true>), walkers, block, 0, stream, sc, pool, l, rounds, steps, refill);
        else hipLaunchKernelGGL((wf_walk<false, false>), walkers, block, 0, stream, sc, pool, l, rounds, steps, refill);
        hipLaunchKernelGGL(wf_hit, hit_grid, block, 0, stream, sc, ra, pool, l);
        if (counted) hipLaunchKernelGGL((wf_walk<true, true>), walkers, block, 0, stream, sc, pool, l, rounds, steps, refill);
        else hipLaunchKernelGGL((wf_walk<true, false>), walkers, block, 0, stream, sc, pool, l, rounds, steps, refill);
        hipLaunchKernelGGL(wf_shade, shade_grid, block, 0, stream, sc, ra, pool, l, share);
    }
    for (uint32_t l = levels; l-- > 0u;) {
        if (l + 1u >= levels) continue;   // the nodes of the deepest level have no children
        hipLaunchKernelGGL(wf_combine, grid, block, 0, stream, sc, ra, pool, l);
    }
    hipLaunchKernelGGL(wf_finish, grid, block, 0, stream, pool, reinterpret_cast<f3*>(d_samples), ra.n_work * ra.sample_count, d_stats, reinterpret_cast<uint32_t*>(pool_mem) + chunk);
    return hipGetLastError();
}

}  // namespace srt
