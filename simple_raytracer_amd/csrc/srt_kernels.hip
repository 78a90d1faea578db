// srt_kernels.hip -- gfx950 kernels: the two persistent trace kernels, the sample resolve kernel and the tile
// assembly kernel.
//
// Work items are (pixel, sample) pairs.  Wavefronts are persistent: each one draws chunks of items from a global
// counter (one atomic per chunk) and deals them to its idle lanes with a ballot + prefix count, so lanes whose ray
// tree ended are refilled at once instead of idling until the slowest lane of the wavefront is done.  A lane is a
// small state machine over the sample's ray tree (srt_path.h).  Each trip of the main loop the wavefront votes for
// one code section and every lane that is in the matching phase executes it together; the other lanes wait.  That
// keeps the 64 lanes on the same instructions instead of serialising the sections under divergent masks.
//
//   trace_kernel        worlds that need no traversal stack (flat lists, reference trees of a few nodes): phases
//                       TRACE / LIGHT / RETURN, a query is a code section that every participating lane sits through.
//   mesh_trace_kernel   worlds with a tree that is walked with a stack: a traversal is a resumable object and
//                       shadow queries are lane states of their own (one copy of the traversal code).
//
// The traversal stack is a per-wavefront slab of LDS with one column per lane (entry i of lane l at [i][l]:
// conflict-free); the ray-tree frames (one per pending recursion level) are LDS for the first levels of small
// scenes and private memory otherwise.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "../../include/srt.h"
#include "srt_launch.h"
#include "srt_path.h"

// Workspace of srt_render_device_static: every GPU the library runs on holds its own copy of this array, at its own
// address, until the process ends.  In host code the name stands for the symbol, not for device memory: the address
// for a kernel argument or a memset comes from hipGetSymbolAddress (static_scratch_setup below), once per device.
static __device__ __align__(256) unsigned char scratch[SRT_STATIC_SCRATCH_BYTES];

namespace srt {

namespace {
constexpr int kMaxScratchDevices = 64;
unsigned char* g_scratch_ptr[kMaxScratchDevices] = {};
}

// Device `device` must be current.  Looks up (once per device) where that device holds `scratch`.
hipError_t static_scratch_setup(int device, unsigned char** ptr)
{
    *ptr = nullptr;
    if (device < 0 || device >= kMaxScratchDevices) return hipErrorInvalidDevice;
    if (!g_scratch_ptr[device]) {
        unsigned char* p = nullptr;
        hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = p;
    }
    *ptr = g_scratch_ptr[device];
    return hipSuccess;
}

#ifndef SRT_WAVES_PER_SIMD_MESH
#define SRT_WAVES_PER_SIMD_MESH 4   // mesh_trace_kernel waits on node gathers: 983k triangles 38.6 ms at 3 waves/SIMD, 35.9 at 4 (33 spilled VGPRs)
#endif
#ifndef SRT_WAVES_PER_SIMD_BIG
#define SRT_WAVES_PER_SIMD_BIG 3
#endif
#ifndef SRT_WAVES_PER_SIMD
#define SRT_WAVES_PER_SIMD 3   // occupancy the register allocator is asked to make room for (measured: 2 -> 25.7 ms, 3 -> 21.1 ms, 4 -> 22.7 ms)
#endif

size_t trace_frame_slab_block_bytes();

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr uint32_t kChunk = 256;       // items drawn from the global counter at a time
constexpr uint32_t kScatterSamples = 2; // trace_kernel: samples at the end of a pass that are enumerated in scattered order
constexpr int kTraceVote = 24;         // run the TRACE section while at least this many lanes want it
constexpr int kMaxBlocks = 256 * 5;    // persistent grid: never more blocks than 5 per CU (the timeline records are sized for it)

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Items to draw next from the global counter, given where the last draw ended: `chunk` while there is plenty left,
// then an ever smaller share of what is left (down to one item per lane), so that no wavefront ends the launch with
// a long range of its own.  A range of items that all hit a mesh costs a wavefront milliseconds.
__device__ __forceinline__ uint32_t next_draw(uint32_t left, uint32_t guide, uint32_t chunk)
{
    const uint32_t share = (left / guide) & ~63u;      // guide = 2 x the wavefronts of the grid
    return share >= chunk ? chunk : (share < 64u ? 64u : share);
}

// Profiler of the counting builds: shader-clock cycles between consecutive marks, by the part that ended.
// The records are per lane: a lane that sits out a part does not execute the marks inside it, and the time goes to the
// next mark it does execute.  The kernels therefore mark PROF_WAIT_* where the lanes of a wavefront join again: summed
// over the lanes, slots 0-10 hold the lane-cycles spent working in each part and the wait slots the lane-cycles spent
// waiting for other lanes' parts.
struct ClockProf {
    unsigned long long cyc[PROF_SLOTS] = {};
    unsigned long long t = 0ull;
    __device__ __forceinline__ void start() { t = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void mark(int i)
    {
        unsigned long long now = __builtin_readcyclecounter();
        cyc[i] += now - t;
        t = now;
    }
};

// Sharing one ray tree between lanes.  Once the global counter is exhausted, a wavefront whose lanes are partly idle
// lets busy lanes hand subtrees to the idle ones: a lane gives away child 1 or 2 (the metal / reflect child, whose ray
// is stored in the frame) of its shallowest pending node that still has one, one subtree per trip, so after a few
// trips it keeps only the first-child spine of its tree.  The helper lane traces the subtree as a work item of its
// own -- same pixel, sample and node id, hence the same random numbers and the same colour.  When it is done it keeps
// the colour (LANE_DELIVER) until the giver has reached the point where it would have started that child and waits for
// it (lane_return -> LANE_WAIT); then the colour travels by readlane and the giver resumes as if the child had just
// returned (LANE_RETURN).  No memory is involved.  Helpers give away parts of their subtrees in turn, so one deep tree
// (255 nodes at depth 8 with two children per node, 3,280 with three) spreads over the wavefront instead of keeping
// one lane busy long after everything else has finished: the time of a tree is then set by its depth, not its node
// count.  Results and ray counts do not change.  The children that a lane can still give away are kept in a bit mask
// (`open`, bit 2*level + child - 1) instead of being searched for in the frames.
constexpr int kForkLevels = 8;         // frames at levels below this can give children away
struct TreeFork {
    uint32_t given;    // bit 2*level + child - 1: that child of the frame at `level` is with another lane
    uint32_t wait;     // LANE_WAIT: the child (of the frame at ln.level - 1) whose colour the lane waits for
    __device__ __forceinline__ bool donated(int32_t lvl, uint32_t child) const { return lvl < kForkLevels && ((given >> (2 * lvl + (int32_t)child - 1)) & 1u) != 0u; }
    __device__ __forceinline__ void wait_for(uint32_t child) { wait = child; }
};
constexpr uint32_t kMaxStalledTrips = 4096u;   // trips in a row with only waiting lanes before a wavefront gives up (never reached unless a colour is lost)
constexpr int kStatsAlarm = 11;                // stats word that counts wavefronts that gave up
constexpr int kStatsUnasked = 48;              // stats word that counts the shadow queries answered without a traversal
constexpr int kStatsTune = 12;                 // stats words 12-15: traversals finished / wall-clock ticks under quantum preset 0, then 1
constexpr int kForkMinDepth = 2;       // only subtrees that can still grow (remaining depth >= this)
constexpr int kSteadyForkDepth = 5;    // mesh kernel, while new items are left: only subtrees with at least this much depth left
constexpr size_t kTimelinePerWave = 64;     // -DSRT_TIMELINE builds: float4 entries of scratch per wavefront of the grid

enum { LANE_DELIVER = 5 };

// `open` of a lane that enters the closing loop in the middle of a tree
template <class FRAMES>
__device__ __forceinline__ uint32_t open_children(const Lane& ln, int phase, const FRAMES& frames, const SceneView& sc)
{
    uint32_t open = 0u;
    if (phase == LANE_IDLE) return open;
    for (int32_t j = 0; j < ln.level && j < kForkLevels; j++) {
        const Frame f = frames.load(j);
        if ((int32_t)(f.meta >> 24) < kForkMinDepth) break;      // deeper frames have less depth left still
        const float metal = sc.materials[2u * (f.meta & 0x00FFFFFFu)].w;
        const uint32_t running = ((ln.key.node >> (2 * (ln.level - 1 - j))) & 3u) - 1u;
        if (metal != 0.0f && running < 1u) open |= 1u << (2 * j);
        if (metal != 1.0f && running < 2u) open |= 2u << (2 * j);
    }
    return open;
}

// Helpers whose giver waits for their subtree hand the colour over.
// tree_base (optional): ln.rays_path - *tree_base is the size of the tree (or subtree) the lane is on; a giver's tree grows
// by the subtree it gets back.
__device__ __forceinline__ void rendezvous(Lane& ln, int& phase, TreeFork& fork, int32_t& deliver, uint32_t lane, uint32_t* tree_base = nullptr)
{
    if (__ballot(phase == LANE_DELIVER) == 0ull) return;
    const int g = phase == LANE_DELIVER ? (deliver & 63) : (int)lane;
    const uint32_t my_wait = phase == LANE_WAIT ? 2u * (uint32_t)(ln.level - 1) + fork.wait - 1u : 0xFFFFFFFFu;
    const uint32_t g_wait = (uint32_t)__shfl((int)my_wait, g, 64);
    unsigned long long match = __ballot(phase == LANE_DELIVER && g_wait == (uint32_t)(deliver >> 8));
    while (match != 0ull) {
        const int h = (int)__builtin_ctzll(match);
        match &= match - 1ull;
        const uint32_t to = (uint32_t)__builtin_amdgcn_readlane(g, h);
        const float x = bitsf((uint32_t)__builtin_amdgcn_readlane((int)fbits(ln.former.x), h));
        const float y = bitsf((uint32_t)__builtin_amdgcn_readlane((int)fbits(ln.former.y), h));
        const float z = bitsf((uint32_t)__builtin_amdgcn_readlane((int)fbits(ln.former.z), h));
        if (tree_base != nullptr) {
            const uint32_t grown = (uint32_t)__builtin_amdgcn_readlane((int)(ln.rays_path - *tree_base), h);
            if (lane == to) *tree_base -= grown;
        }
        if (lane == to) {
            ln.former = mk3(x, y, z);
            ln.key.node = ln.key.node * 4u + fork.wait + 1u;
            fork.given &= ~(1u << my_wait);
            phase = LANE_RETURN;
        }
        if (lane == (uint32_t)h) { phase = LANE_IDLE; deliver = -1; }
    }
}

// Pairs lanes that can give a child away with idle lanes; returns true on the lanes that took one (see share_subtrees).
template <class FRAMES>
__device__ __forceinline__ bool share_open_children(Lane& ln, int phase, const FRAMES& frames, TreeFork& fork, uint32_t& open, int32_t& deliver,
                                                    uint32_t* my_pair, const RenderArgs& ra, uint32_t lane, uint32_t& handed)
{
    const unsigned long long idle_mask = __ballot(phase == LANE_IDLE);
    const bool cand = phase != LANE_IDLE && phase != LANE_DELIVER && open != 0u;
    const unsigned long long cand_mask = __ballot(cand);
    const uint32_t n_cand = (uint32_t)__popcll(cand_mask), n_idle = (uint32_t)__popcll(idle_mask);
    const uint32_t n_pairs = n_cand < n_idle ? n_cand : n_idle;
    if (n_pairs == 0u) return false;
    handed += n_pairs;
    const uint32_t rank_c = lanes_below(cand_mask), rank_i = lanes_below(idle_mask);
    const bool gives = cand && rank_c < n_pairs, takes = phase == LANE_IDLE && rank_i < n_pairs;
    uint32_t g_bit = 0u, g_node = 0u;
    int32_t g_depth = 0;
    f3 g_point = mk3(0.0f, 0.0f, 0.0f), g_dir = mk3(0.0f, 0.0f, 0.0f);
    if (gives) {
        g_bit = (uint32_t)__builtin_ctz(open);                 // the shallowest frame: the largest subtree
        const int32_t j = (int32_t)(g_bit >> 1);
        const Frame f = frames.load(j);
        g_point = f.point; g_dir = f.refl; g_depth = (int32_t)(f.meta >> 24);
        const int32_t below = phase == LANE_WAIT ? ln.level - 1 : ln.level;   // a waiting lane holds its deepest frame's node id
        g_node = (ln.key.node >> (2 * (below - j))) * 4u + (g_bit & 1u) + 2u;
        open &= ~(1u << g_bit);
        fork.given |= 1u << g_bit;
        my_pair[rank_c] = lane;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int src = takes ? (int)my_pair[rank_i] : (int)lane;
    const uint32_t s_pixel = (uint32_t)__shfl((int)ln.key.pixel, src, 64), s_sample = (uint32_t)__shfl((int)ln.key.sample, src, 64);
    const uint32_t s_node = (uint32_t)__shfl((int)g_node, src, 64), s_bit = (uint32_t)__shfl((int)g_bit, src, 64);
    const int32_t s_depth = __shfl(g_depth, src, 64);
    const f3 s_point = mk3(__shfl(g_point.x, src, 64), __shfl(g_point.y, src, 64), __shfl(g_point.z, src, 64));
    const f3 s_dir = mk3(__shfl(g_dir.x, src, 64), __shfl(g_dir.y, src, 64), __shfl(g_dir.z, src, 64));
    if (takes) {
        ln.key.seed = ra.seed; ln.key.pixel = s_pixel; ln.key.sample = s_sample;
        ln.key.node = s_node;
        ln.qo = s_point; ln.qd = s_dir; ln.qtmin = 0.01f;
        ln.depth = s_depth; ln.level = 0;
        fork.given = 0u; fork.wait = 0u; open = 0u;
        deliver = src | (int32_t)(s_bit << 8);
    }
    return takes;
}

// SMALL kernels serve scenes that need no traversal stack (a reference tree of a few nodes, walked with the
// wave-uniform loop, or a handful of surfaces scanned linearly) and whose primitive and material tables fit the
// limits below.  Their LDS holds those two tables (they are indexed per lane) and the first kLdsFrameLevels
// levels of every lane's ray-tree frames instead of the stack.
constexpr uint32_t kSmallPrims = 64, kSmallMaterials = 64;
constexpr int kLdsFrameLevels = 3;
constexpr uint32_t kSmallPrimWords = 3 * kSmallPrims, kSmallMatWords = 2 * kSmallMaterials;
constexpr uint32_t kSmallWords = kSmallPrimWords + kSmallMatWords;

// Byte offset, in the frame slab, of the level-0 frame of this lane: every wavefront of the grid owns FRAME_SLOTS levels
// of 64 frames.
__device__ __forceinline__ uint32_t my_frame_offset(uint32_t wave, uint32_t lane)
{
    return (blockIdx.x * (uint32_t)kWaves + wave) * (uint32_t)(FRAME_SLOTS * FRAME_SLAB_LEVEL_BYTES) + lane * 64u;
}

__device__ __forceinline__ void copy_words(q4* dst, const q4* src, uint32_t n)
{
    for (uint32_t i = threadIdx.x; i < n; i += kBlock) dst[i] = src[i];
}

template <bool EXACT, bool COUNT, bool SMALL>
__global__ void __launch_bounds__(kBlock, SMALL ? SRT_WAVES_PER_SIMD : SRT_WAVES_PER_SIMD_BIG) trace_kernel(SceneView gsc, RenderArgs ra, f3* __restrict__ samples,
                                                       unsigned long long* __restrict__ stats, uint32_t* __restrict__ work_counter,
                                                       float4* __restrict__ timeline /* -DSRT_TIMELINE builds only */, q4* __restrict__ frame_slab, int trace_vote, uint32_t chunk, int share, uint32_t guide)
{
    __shared__ float lds_frames[SMALL ? kWaves * kLdsFrameLevels * 14 * 64 : 1];
    __shared__ uint32_t lds_pair[kWaves * 64];   // handing over subtrees: giver lane by rank
    __shared__ float unit[UNIT_TABLE];   // (d - 100) / 100 for the RNG's base-200 digits
    __shared__ q4 small_tables[SMALL ? kSmallWords : 1];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x < (uint32_t)UNIT_TABLE) unit[threadIdx.x] = (float)((int)threadIdx.x - 100) / 100.0f;
    SceneView sc = gsc;
    if (SMALL) {
        // Only the tables that are indexed per lane go to LDS: the material of a hit and the record of the winning
        // primitive.  Nodes, lights and the primitives tested by wave-uniform loops stay behind scalar loads, which
        // keep them in SGPRs instead of spending a VGPR per value.
        q4* prims = small_tables;
        q4* mats = prims + kSmallPrimWords;
        copy_words(prims, gsc.hit_prims, 3u * gsc.n_prims);
        copy_words(mats, gsc.materials, 2u * gsc.n_materials);
        sc.hit_prims = prims; sc.materials = mats;
    }
    __syncthreads();

#ifdef SRT_TIMELINE
    const unsigned long long tl_start = wall_clock64();
    unsigned long long tl_dry = 0ull, tl_trips = 0ull;
#endif
    Lane ln;
    typename std::conditional<SMALL, HybridFrames<kLdsFrameLevels>, SlabFrames>::type frames;
    if constexpr (SMALL) {
        frames.deep.slab = reinterpret_cast<char*>(frame_slab);
        frames.deep.mine = my_frame_offset(wave, lane);
        frames.lds = &lds_frames[wave * (kLdsFrameLevels * 14 * 64) + lane];
    } else {
        frames.slab = reinterpret_cast<char*>(frame_slab);
        frames.mine = my_frame_offset(wave, lane);
    }
    Counters cnt;
    cnt.box_tests = cnt.sphere_tests = cnt.tri_tests = cnt.rect_tests = 0u;
    ln.rays_path = ln.rays_shadow = ln.hit_nodes = 0u;
    int phase = LANE_IDLE;

    uint32_t sect[4] = {0u, 0u, 0u, 0u};  // counting builds: section executions and the lanes that took part
    typename std::conditional<COUNT, ClockProf, NoProf>::type prof;   // counting builds: shader clock per part
#define SRT_MARK(i) prof.mark(i)
    uint32_t next = 0u, end = 0u;  // wave-uniform: the wavefront's private range of items
    bool dry = false;              // wave-uniform: the global counter is exhausted
    uint32_t draw = chunk;         // wave-uniform: items per draw from the global counter
    uint32_t handed = 0u;          // wave-uniform: subtrees that changed lanes in the closing loop
    unsigned long long t_dry = 0ull;   // counting builds: wall clock (100 MHz) when this wavefront found the counter exhausted

    for (;;) {
        if constexpr (COUNT) prof.start();
#ifdef SRT_TIMELINE
        if (dry) tl_trips++; else tl_trips += 1ull << 32;
#endif
        if (dry && share != 0) break;   // no new items: the trees that are still being walked finish in the loop below
        const int n_trace = __popcll(__ballot(phase == LANE_TRACE || phase == LANE_RETURN || (phase == LANE_IDLE && !dry)));
        const int n_light = __popcll(__ballot(phase == LANE_LIGHT));
        if (n_trace + n_light == 0) break;

        if (n_trace >= trace_vote || n_light == 0) {
            // ---- RETURN: values travel up the tree until a parent starts its next child or the sample is complete
            if (phase == LANE_RETURN) {
                f3 ret;
                phase = lane_return(ln, ret, sc, ra, frames);
                if (phase == LANE_IDLE) samples[ln.slot] = ret;
                SRT_MARK(PROF_RETURN);
            }
            SRT_MARK(PROF_WAIT_RETURN);
            // ---- deal items to idle lanes
            unsigned long long idle = __ballot(phase == LANE_IDLE);
            while (idle != 0ull) {
                if (next >= end) {
                    if (dry) break;
                    uint32_t base = 0u;
                    if (lane == 0u) base = atomicAdd(work_counter, draw);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base >= ra.n_items) {
                        dry = true;
                        if constexpr (COUNT) t_dry = wall_clock64();
#ifdef SRT_TIMELINE
                        tl_dry = wall_clock64();
#endif
                        break;
                    }
                    next = base;
                    end = base + draw < ra.n_items ? base + draw : ra.n_items;
                    // towards the end of the block-ordered items, and again towards the end of the scattered ones
                    const uint32_t n_coherent = ra.n_work * ra.coherent;
                    draw = next_draw((end < n_coherent ? n_coherent : ra.n_items) - end, guide, chunk);
                }
                uint32_t avail = end - next, want = (uint32_t)__popcll(idle);
                uint32_t rank = lanes_below(idle);
                if (phase == LANE_IDLE && rank < avail) {
                    if (item_start(ln, ra, next + rank, unit)) phase = LANE_TRACE;  // padding pixels of edge tiles are skipped
                    SRT_MARK(PROF_DEAL);
                }
                next += want < avail ? want : avail;
                idle = __ballot(phase == LANE_IDLE);
            }
            SRT_MARK(PROF_WAIT_DEAL);
            // ---- TRACE: closest-hit query of a tree node
            if (COUNT) { sect[0]++; sect[1] += (uint32_t)__popcll(__ballot(phase == LANE_TRACE)); }
            if (phase == LANE_TRACE) {
                Hit h = closest_hit<EXACT, COUNT>(sc, ln.qo, ln.qd, ln.qtmin, cnt);
                SRT_MARK(PROF_CLOSEST);
                f3 ret;
                phase = lane_on_closest(ln, h, ret, sc, ra, frames, unit);
                if (phase == LANE_IDLE) samples[ln.slot] = ret;
                SRT_MARK(PROF_ON_CLOSEST);
            }
            SRT_MARK(PROF_WAIT_TRACE);
        } else {
            // ---- LIGHT: one shadow query per light, then the node's shading
            if (COUNT) { sect[2]++; sect[3] += (uint32_t)n_light; }
            if (phase == LANE_LIGHT) {
                uint32_t aux[4];
                rng_call(ln.key, CALL_AUX, aux);
                SRT_MARK(PROF_AUX);
                for (uint32_t li = 0; li < sc.n_lights; li++) {
                    f3 o, d;
                    lane_shadow_query(ln, sc, li, aux, unit, o, d);
                    SRT_MARK(PROF_SHADOW_SETUP);
                    Hit h = closest_hit<EXACT, COUNT>(sc, o, d, 0.001f, cnt);
                    SRT_MARK(PROF_SHADOW_QUERY);
                    lane_on_shadow(ln, h, o, d, sc, li);
                    SRT_MARK(PROF_LIGHT_EVAL);
                }
                f3 ret;
                phase = lane_shade(ln, ln.former, ret, sc, ra, frames, aux, unit, prof);
                if (phase == LANE_IDLE) samples[ln.slot] = ret;
            }
            SRT_MARK(PROF_WAIT_LIGHT);
        }
    }

    // ---- the end of the launch: no new items.  Lanes without work take subtrees from the lanes that still have a
    // tree, so that the time of the last deep trees is set by their depth and not by their node count.
    if (dry && share != 0) {
        TreeFork fork;
        fork.given = 0u; fork.wait = 0u;
        int32_t deliver = -1;      // >= 0: this lane walks a subtree for lane (deliver & 63); deliver >> 8 is the giver's bit for it
        uint32_t open = open_children(ln, phase, frames, sc);
        uint32_t* my_pair = &lds_pair[wave * 64];
        uint32_t stalled = 0u;   // wave-uniform: trips in a row in which no lane had anything to run
        // a tree ended: its colour is a sample, or waits for the giver
        auto finish = [&](f3 ret) {
            if (deliver >= 0) { ln.former = ret; phase = LANE_DELIVER; }
            else { samples[ln.slot] = ret; phase = LANE_IDLE; }
        };
        // a frame was pushed for the node at ln.level - 1 and its child 0 started: which other children can be given away
        auto pushed = [&]() {
            const int32_t j = ln.level - 1;
            if (j >= kForkLevels || ln.depth < kForkMinDepth) return;
            const float metal = sc.materials[2u * ln.mat].w;
            if (metal != 0.0f) open |= 1u << (2 * j);
            if (metal != 1.0f) open |= 2u << (2 * j);
        };
        for (;;) {
#ifdef SRT_TIMELINE
            tl_trips++;
#endif
            rendezvous(ln, phase, fork, deliver, lane);
            if (share_open_children(ln, phase, frames, fork, open, deliver, my_pair, ra, lane, handed)) phase = LANE_TRACE;
            const unsigned long long m_trace = __ballot(phase == LANE_TRACE || phase == LANE_RETURN), m_light = __ballot(phase == LANE_LIGHT);
            if ((m_trace | m_light) == 0ull) {
                if (__ballot(phase == LANE_WAIT) == 0ull) break;
                // givers and helpers that meet at the top of the loop: every such trip hands over at least one colour, so
                // a long run of them means a colour got lost -- leave with an alarm instead of spinning on the GPU
                if (++stalled > kMaxStalledTrips) { if (lane == 0u) atomicAdd(&stats[kStatsAlarm], 1ull); break; }
                continue;
            }
            stalled = 0u;
            if (m_trace != 0ull) {
                if (phase == LANE_RETURN) {
                    f3 ret;
                    const int p = lane_return(ln, ret, sc, ra, frames, fork);
                    // frames above the current level are gone; a child that starts can no longer be given away
                    open &= ln.level < 16 ? (1u << (2 * ln.level)) - 1u : 0xFFFFFFFFu;
                    if (p == LANE_TRACE) { const uint32_t c = (ln.key.node & 3u) - 1u; if (ln.level - 1 < kForkLevels) open &= ~(1u << (2 * (ln.level - 1) + (int32_t)c - 1)); }
                    if (p == LANE_IDLE) finish(ret); else phase = p;
                }
                if (COUNT) { sect[0]++; sect[1] += (uint32_t)__popcll(__ballot(phase == LANE_TRACE)); }
                if (phase == LANE_TRACE) {
                    Hit h = closest_hit<EXACT, COUNT>(sc, ln.qo, ln.qd, ln.qtmin, cnt);
                    f3 ret;
                    const int p = lane_on_closest(ln, h, ret, sc, ra, frames, unit);
                    if (p == LANE_TRACE) pushed();      // a scene without lights: the node was shaded at once
                    if (p == LANE_IDLE) finish(ret); else phase = p;
                }
            }
            if (__ballot(phase == LANE_LIGHT) != 0ull) {
                if (COUNT) { sect[2]++; sect[3] += (uint32_t)__popcll(__ballot(phase == LANE_LIGHT)); }
                if (phase == LANE_LIGHT) {
                    uint32_t aux[4];
                    rng_call(ln.key, CALL_AUX, aux);
                    for (uint32_t li = 0; li < sc.n_lights; li++) {
                        f3 o, d;
                        lane_shadow_query(ln, sc, li, aux, unit, o, d);
                        Hit h = closest_hit<EXACT, COUNT>(sc, o, d, 0.001f, cnt);
                        lane_on_shadow(ln, h, o, d, sc, li);
                    }
                    f3 ret;
                    const int p = lane_shade(ln, ln.former, ret, sc, ra, frames, aux, unit);
                    if (p == LANE_TRACE) pushed();
                    if (p == LANE_IDLE) finish(ret); else phase = p;
                }
            }
        }
    }

#ifdef SRT_TIMELINE
    if (lane == 0u && timeline != nullptr) {
        unsigned long long* rec = reinterpret_cast<unsigned long long*>(timeline + (size_t)1100 * kWaves * kTimelinePerWave) + ((size_t)blockIdx.x * kWaves + wave) * 4;
        rec[0] = tl_start; rec[1] = tl_dry; rec[2] = wall_clock64(); rec[3] = tl_trips;
    }
#endif
    const unsigned long long t_exit = COUNT ? wall_clock64() : 0ull;
    unsigned long long v0 = wave_sum(ln.rays_path), v1 = wave_sum(ln.rays_shadow), v2 = wave_sum(ln.hit_nodes);
    if (lane == 0u) {
        atomicAdd(&stats[0], v0);
        atomicAdd(&stats[1], v1);
        atomicAdd(&stats[2], v2);
    }
    if constexpr (COUNT) {
        for (int i = 0; i < PROF_SLOTS; i++) {
            const unsigned long long lanes = wave_sum(prof.cyc[i]);
            if (lane == 0u) atomicAdd(&stats[32 + i], lanes);
        }
        unsigned long long c0 = wave_sum(cnt.box_tests), c1 = wave_sum(cnt.sphere_tests), c2 = wave_sum(cnt.tri_tests), c3 = wave_sum(cnt.rect_tests);
        if (lane == 0u) {
            atomicAdd(&stats[3], c0);
            atomicAdd(&stats[4], c1);
            atomicAdd(&stats[5], c2);
            atomicAdd(&stats[6], c3);
            for (int i = 0; i < 4; i++) atomicAdd(&stats[7 + i], (unsigned long long)sect[i]);
            for (int i = 0; i < PROF_WAIT_RETURN; i++) atomicAdd(&stats[16 + i], prof.cyc[i]);
            atomicAdd(&stats[16 + 15], (unsigned long long)handed);   // a count, not cycles
            // the tail of the launch: how long wavefronts kept running after the counter ran out (100 MHz ticks)
            const unsigned long long tail = t_exit - t_dry;
            atomicMax(&stats[16 + 11], tail);
            atomicAdd(&stats[16 + 12], tail);
            if (tail > 25000ull) atomicAdd(&stats[16 + 13], 1ull);
            if (tail > 100000ull) atomicAdd(&stats[16 + 14], 1ull);
        }
    }
}

// mesh_trace_kernel: the same work items, lane state machine and per-lane functions for scenes with a tree that is
// traversed with a stack (meshes).  There the length of a query varies from one node visit to hundreds between
// the lanes of a wavefront, so a query is not a section that every lane sits through until the slowest one is
// done: a traversal is resumable (trav_begin / trav_run) and the main loop deals out QUANTA of it.  A lane is
//   walking    it has a closest-hit or shadow query whose traversal is not finished;
//   closest    its closest-hit query is answered: hit record, then the first shadow query, a child or a return;
//   shadow     its shadow query is answered: light evaluation, then the next light, or the node's shading;
//   up         it hands a value up the tree (RETURN) or is idle and wants a new item.
// Each trip the wavefront runs a traversal quantum if at least kWalkVote lanes are walking; with fewer it first serves
// the largest of the other groups as long as one of them has kServeFirst lanes (10,000 spheres + mesh: 341 -> 310 ms
// against "the part that the most lanes are waiting for"; the single meshes do not care), and walks otherwise.  A quantum is kWalkRounds rounds of "descend at most kWalkSteps inner nodes,
// then test the leaf if one was reached".  Measured on a 983,040-triangle mesh (1080p, 64 spp; ms per frame): the
// closest-hit / light sections of trace_kernel 56.1; quanta of 2 x 8 steps 58.2, 8 x 8 48.6, 32 x 8 44.7; unbounded
// quanta (every traversal runs to its end once started) with 8 steps 43.3, 4 steps 40.8, 2 steps 43.1; and with
// kWalkVote 16 instead of 32: 39.8 (all of these on the binary SAH tree; on the 4-wide tree that replaced it 2 steps
// are best: 34.2 / 32.6 / 34.8 / 39.5 ms for 1 / 2 / 3 / 6).  So the defaults never split a traversal; what pays is one copy of the
// traversal code for both kinds of query, short descents between leaf tests, and starting traversals in batches.
enum { MESH_IDLE = 0, MESH_CLOSEST = 1, MESH_SHADOW = 2, MESH_RETURN = 3, MESH_WAIT = 4 };
static_assert(MESH_IDLE == LANE_IDLE && MESH_RETURN == LANE_RETURN && MESH_WAIT == LANE_WAIT, "the subtree-sharing helpers take either kernel's phases");
constexpr int kWalkVote = 16, kWalkRounds = 1 << 24, kWalkSteps = 2;
// The other setting: short quanta.  Where the traversals of a wavefront differ a lot in length (rays scattered by a cloud
// of spheres: the 10,000-sphere scene takes 706 ms per 1080p 16-spp frame with the quanta above and 531 ms with these),
// lanes that are done move on instead of waiting for the longest walk; where they are alike (a single mesh), splitting
// them only adds trips (983,040 triangles: 30.8 -> 42.9 ms).  Which one a scene wants is measured, not guessed: the first
// launches after a commit run both for 3 ms each (preset 0, then preset 1, once the launch is 5 ms old), every wavefront adds the traversals
// it finished and the time it took to two pairs of counters, and whoever finds both pairs filled takes the faster preset;
// the host reads the same counters when it collects the statistics and fixes the preset for later launches.
constexpr int kShortVote = 56, kShortRounds = 6, kShortSteps = 4;
constexpr int kServeFirst = 8;   // lanes in another group that make the wavefront serve that group before it walks (see the vote in the kernel)
constexpr uint32_t kShortScatterSamples = 0xFFFFFFFFu;   // a short-quanta scene enumerates all samples of a pass in scattered order (8 of 16: 481 ms, all: 457 ms)
constexpr unsigned long long kTuneWindowTicks = 300000ull;   // 3 ms of the 100 MHz wall clock per preset ...
constexpr unsigned long long kTuneSkipTicks = 500000ull;     // ... after the first 5 ms of the launch: it starts with nothing but coherent primary rays
struct WalkTune { int vote[2], rounds[2], steps[2]; int mode; int serve; };   // mode: WALK_TUNE_*; preset 0 / 1; serve: see the vote   // mode: WALK_TUNE_*; preset 0 / 1

template <bool EXACT, bool COUNT>
__global__ void __launch_bounds__(kBlock, SRT_WAVES_PER_SIMD_MESH) mesh_trace_kernel(SceneView sc, RenderArgs ra, f3* __restrict__ samples,
                                                            unsigned long long* __restrict__ stats, uint32_t* __restrict__ work_counter,
                                                            float4* __restrict__ timeline /* -DSRT_TIMELINE builds only */, q4* __restrict__ frame_slab, WalkTune tune, uint32_t chunk, uint32_t guide, int share,
                                                            int steady_depth, const uint32_t* __restrict__ run_if)
{
    // a launch that stands by for the level-by-level tracer: it runs only if that one gave its chunk of items up
    if (run_if != nullptr && *run_if == 0u) return;
    __shared__ uint32_t lds_stack[kWaves * STACK_DEPTH * 64];
    __shared__ uint32_t lds_pair[kWaves * 64];   // handing over subtrees: giver lane by rank
    __shared__ float unit[UNIT_TABLE];   // (d - 100) / 100 for the RNG's base-200 digits
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* my_stack = &lds_stack[wave * (STACK_DEPTH * 64) + lane];
    if (threadIdx.x < (uint32_t)UNIT_TABLE) unit[threadIdx.x] = (float)((int)threadIdx.x - 100) / 100.0f;
    __syncthreads();

#ifdef SRT_TIMELINE
    const unsigned long long tl_start = wall_clock64();
    unsigned long long tl_dry = 0ull, tl_trips = 0ull;
#endif
    Lane ln;
    SlabFrames frames;
    frames.slab = reinterpret_cast<char*>(frame_slab);
    frames.mine = my_frame_offset(wave, lane);
    Trav tr;
    tr.d = mk3(0.0f, 0.0f, 0.0f);
    tr.best.t = INFINITY; tr.best.prim = NO_PRIM; tr.best.u = tr.best.v = 0.0f; tr.best.order = 0u;
    tr.cur = TRAV_NONE; tr.sp = 0; tr.pad = 0.0f;
    uint32_t light = 0u;   // the light whose shadow query is pending
    Counters cnt;
    cnt.box_tests = cnt.sphere_tests = cnt.tri_tests = cnt.rect_tests = 0u;
    ln.rays_path = ln.rays_shadow = ln.hit_nodes = 0u;
    ln.level = 0; ln.key.node = 0u;
    int phase = MESH_IDLE;
    TreeFork fork;
    fork.given = 0u; fork.wait = 0u;
    int32_t deliver = -1;      // >= 0: this lane walks a subtree for lane (deliver & 63); deliver >> 8 is the giver's bit for it
    uint32_t open = 0u;        // children this lane could give away (bit 2*level + child - 1); kept once `sharing` is on
    bool sharing = false;      // wave-uniform: no new items, lanes share their trees
    uint32_t* my_pair = &lds_pair[wave * 64];
    uint32_t handed = 0u;
    uint32_t stalled = 0u;     // wave-uniform: trips in a row in which no lane had anything to run
    // The largest ray tree this lane finished, in GetRayColor calls (subtrees that went to other lanes are not in it): one
    // deep tree is milliseconds of one lane's time, and the host sends scenes that have them to the level-by-level tracer.
    uint32_t item_rays_base = 0u, largest_tree = 0u;

    uint32_t sect[4] = {0u, 0u, 0u, 0u};
    typename std::conditional<COUNT, ClockProf, NoProf>::type prof;
    uint32_t next = 0u, end = 0u, draw = chunk;
    bool dry = false;
    // wave-uniform: the quantum preset in use, and the state of its calibration (see WalkTune)
    int preset = tune.mode == WALK_TUNE_SHORT ? 1 : 0;
    bool calibrating = tune.mode == WALK_TUNE_CALIBRATE;
    int tune_window = 0;
    uint32_t tune_done = 0u;
    unsigned long long tune_start = 0ull;
    // takes the faster preset once both have been measured enough (`force`: with whatever there is)
    auto settle = [&](bool force) {
        const unsigned long long q0 = __hip_atomic_load(&stats[kStatsTune + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __hip_atomic_load(&stats[kStatsTune + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long q1 = __hip_atomic_load(&stats[kStatsTune + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t1 = __hip_atomic_load(&stats[kStatsTune + 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool enough = t0 >= kWalkTuneEnoughTicks && t1 >= kWalkTuneEnoughTicks;
        if (!enough && !force) return;
        preset = enough && walk_tune_prefers_short((double)q0, (double)t0, (double)q1, (double)t1) ? 1 : 0;
        preset = __builtin_amdgcn_readfirstlane(preset);
        calibrating = false;
    };
    if (calibrating) { settle(false); tune_start = wall_clock64(); }

    // The next shadow query of the node, starting with light `light`: lights whose ray cannot matter are accounted for
    // on the spot.  Returns false when no light is left; the lane then stays in the "shadow answered" group without a
    // query (asked = false), and that part shades the node.
    uint32_t unasked = 0u;     // counting builds: shadow rays that were not traced
    bool asked = false;        // MESH_SHADOW: tr holds the query of light `light`
    auto begin_shadow = [&](const uint32_t aux[4]) {
        while (light < sc.n_lights && !lane_shadow_matters(ln, sc, light)) {
            lane_skip_light(ln);
            light++;
            if (COUNT) unasked++;
        }
        asked = light < sc.n_lights;
        if (asked) {
            f3 o, d;
            lane_shadow_query(ln, sc, light, aux, unit, o, d);
            tr.d = d;
            tr.cur = TRAV_NEW;
        } else {
            tr.cur = TRAV_NONE;
        }
        return asked;
    };
    // the lane's next state after a per-lane function returned LANE_*; `ret` is the sample's colour for LANE_IDLE;
    // up: the function was lane_return (frames may have gone, a later child may have started), else a frame may have
    // been pushed
    auto enter = [&](int lane_phase, f3 ret, bool up) {
        // While there are new items, only subtrees that can still grow large (steady_depth levels left) may change lanes:
        // they are what keeps a single lane busy long after the rest of its wavefront has moved on.
        if (share != 0) {
            if (up) {
                open &= ln.level < 16 ? (1u << (2 * ln.level)) - 1u : 0xFFFFFFFFu;
                if (lane_phase == LANE_TRACE && ln.level - 1 < kForkLevels) open &= ~(1u << (2 * (ln.level - 1) + (int32_t)((ln.key.node & 3u) - 1u) - 1));
            } else if (lane_phase == LANE_TRACE && ln.level >= 1 && ln.level - 1 < kForkLevels && ln.depth >= (sharing ? kForkMinDepth : steady_depth)) {
                const float metal = sc.materials[2u * ln.mat].w;
                if (metal != 0.0f) open |= 1u << (2 * (ln.level - 1));
                if (metal != 1.0f) open |= 2u << (2 * (ln.level - 1));
            }
        }
        if (lane_phase == LANE_IDLE) {
            if (deliver >= 0) { ln.former = ret; phase = LANE_DELIVER; }
            else {
                samples[ln.slot] = ret; phase = MESH_IDLE;
                const uint32_t tree = ln.rays_path - item_rays_base;
                largest_tree = tree > largest_tree ? tree : largest_tree;
            }
        }
        else if (lane_phase == LANE_WAIT) phase = MESH_WAIT;
        else if (lane_phase == LANE_RETURN) phase = MESH_RETURN;
        else if (lane_phase == LANE_TRACE) { tr.d = ln.qd; tr.cur = TRAV_NEW; phase = MESH_CLOSEST; }
        else {   // LANE_LIGHT: the first light's shadow query
            light = 0u;
            uint32_t aux[4];
            rng_call(ln.key, CALL_AUX, aux);
            begin_shadow(aux);
            phase = MESH_SHADOW;
        }
    };

    for (;;) {
        if constexpr (COUNT) prof.start();
        if (calibrating) {
            const unsigned long long now = wall_clock64();
            if (dry) settle(true);      // the end of a launch says nothing about its steady state
            else if (tune_window == 0 && now - tune_start >= kTuneSkipTicks) { tune_window = 1; tune_done = 0u; tune_start = now; }
            else if (tune_window > 0 && now - tune_start >= kTuneWindowTicks) {
                if (lane == 0u) {
                    atomicAdd(&stats[kStatsTune + 2 * preset], (unsigned long long)tune_done);
                    atomicAdd(&stats[kStatsTune + 2 * preset + 1], now - tune_start);
                }
                tune_done = 0u; tune_start = now;
                if (++tune_window < 3) preset = 1;
                else settle(true);
            }
        }
        const int walk_vote = tune.vote[preset], rounds = tune.rounds[preset], steps = tune.steps[preset];
        // ---- subtrees that other lanes traced for this one; subtrees for idle lanes once there are no new items
#ifdef SRT_TIMELINE
        if (dry) tl_trips++; else tl_trips += 1ull << 32;
#endif
        if (share != 0 && (dry || __ballot(open != 0u || phase == LANE_DELIVER || phase == MESH_WAIT) != 0ull)) {
            if (dry && !sharing) {
                // from here on smaller subtrees change lanes too: look the pending frames over once (a lane that waits for a
                // child holds its frame's node id instead of a child's and keeps the mask it has)
                if (phase != MESH_WAIT) open = open_children(ln, phase, frames, sc) & ~fork.given;
                sharing = true;
            }
            rendezvous(ln, phase, fork, deliver, lane, &item_rays_base);
            if (share_open_children(ln, phase, frames, fork, open, deliver, my_pair, ra, lane, handed)) {
                tr.d = ln.qd; tr.cur = TRAV_NEW; phase = MESH_CLOSEST; item_rays_base = ln.rays_path;
            }
        }
        const bool walking = (phase == MESH_CLOSEST || phase == MESH_SHADOW) && tr.cur != TRAV_NONE;
        const int n_walk = __popcll(__ballot(walking));
        const int n_closest = __popcll(__ballot(phase == MESH_CLOSEST && !walking));
        const int n_shadow = __popcll(__ballot(phase == MESH_SHADOW && !walking));
        const int n_up = __popcll(__ballot(phase == MESH_RETURN || (phase == MESH_IDLE && !dry)));
        const int n_wait = __popcll(__ballot(phase == MESH_WAIT || phase == LANE_DELIVER));
        if (n_walk + n_closest + n_shadow + n_up + n_wait == 0) break;
        if (n_walk + n_closest + n_shadow + n_up == 0) {
            // givers and helpers that meet at the top of the loop (see trace_kernel's closing loop for the alarm)
            if (++stalled > kMaxStalledTrips) { if (lane == 0u) atomicAdd(&stats[kStatsAlarm], 1ull); break; }
            continue;
        }
        stalled = 0u;

        // Short of walk_vote walking lanes, the other groups are served first while one of them has at least tune.serve
        // lanes: their parts are short, and every lane they release joins the next quantum instead of sitting it out.
        const bool serve = n_walk < walk_vote && (n_up >= tune.serve || n_closest >= tune.serve || n_shadow >= tune.serve);
        if (!serve && n_walk > 0) {
            // ---- a quantum of traversal
            if (COUNT) { sect[0]++; sect[1] += (uint32_t)n_walk; }
            if (walking) {
                const bool sh = phase == MESH_SHADOW;
                const f3 o = sh ? ln.P : ln.qo;
                const float t_min = sh ? 0.001f : ln.qtmin;
                if (tr.cur == TRAV_NEW) trav_begin<EXACT, COUNT, 64>(sc, o, t_min, tr, my_stack, cnt);
                // queries that start far outside the scene test widened boxes; a wavefront without one keeps the plain walk
                if (!EXACT && __ballot(tr.pad != 0.0f) != 0ull) trav_run<EXACT, COUNT, 64, !EXACT>(sc, o, t_min, tr, my_stack, cnt, rounds, steps);
                else trav_run<EXACT, COUNT, 64>(sc, o, t_min, tr, my_stack, cnt, rounds, steps);
            }
            if (calibrating) tune_done += (uint32_t)__popcll(__ballot(walking && tr.cur == TRAV_NONE));
            prof.mark(PROF_CLOSEST);
        } else if (n_up >= n_closest && n_up >= n_shadow) {
            // ---- values travel up the tree; idle lanes take new items
            if (COUNT) { sect[2]++; sect[3] += (uint32_t)n_up; }
            if (phase == MESH_RETURN) {
                f3 ret;
                int p = lane_return(ln, ret, sc, ra, frames, fork);
                enter(p, ret, true);
            }
            prof.mark(PROF_RETURN);
            unsigned long long idle = __ballot(phase == MESH_IDLE);
            while (idle != 0ull) {
                if (next >= end) {
                    if (dry) break;
                    uint32_t base = 0u;
                    if (lane == 0u) base = atomicAdd(work_counter, draw);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base >= ra.n_items) {
                        dry = true;
#ifdef SRT_TIMELINE
                        tl_dry = wall_clock64();
#endif
                        break;
                    }
                    next = base;
                    end = base + draw < ra.n_items ? base + draw : ra.n_items;
                    draw = next_draw(ra.n_items - end, guide, chunk);
                }
                uint32_t avail = end - next, want = (uint32_t)__popcll(idle);
                uint32_t rank = lanes_below(idle);
                if (phase == MESH_IDLE && rank < avail) {
                    if (item_start(ln, ra, next + rank, unit)) { tr.d = ln.qd; tr.cur = TRAV_NEW; phase = MESH_CLOSEST; item_rays_base = ln.rays_path; }
                }
                next += want < avail ? want : avail;
                idle = __ballot(phase == MESH_IDLE);
            }
            prof.mark(PROF_DEAL);
        } else if (n_closest >= n_shadow) {
            // ---- closest-hit queries that are answered
            if (COUNT) { sect[2]++; sect[3] += (uint32_t)n_closest; }
            if (phase == MESH_CLOSEST && !walking) {
                f3 ret;
                int p = lane_on_closest(ln, tr.best, ret, sc, ra, frames, unit);
                enter(p, ret, false);
            }
            prof.mark(PROF_ON_CLOSEST);
        } else {
            // ---- shadow queries that are answered: the light's contribution, then the next light or the shading
            if (COUNT) { sect[2]++; sect[3] += (uint32_t)n_shadow; }
            if (phase == MESH_SHADOW && !walking) {
                if (asked) {
                    lane_on_shadow(ln, tr.best, ln.P, tr.d, sc, light);
                    light++;
                }
                uint32_t aux[4];
                rng_call(ln.key, CALL_AUX, aux);
                if (begin_shadow(aux)) {
                    // the next light's query is set up
                } else {
                    f3 ret;
                    int p = lane_shade(ln, ln.former, ret, sc, ra, frames, aux, unit);
                    enter(p, ret, false);
                }
            }
            prof.mark(PROF_LIGHT_EVAL);
        }
    }

#ifdef SRT_TIMELINE
    if (lane == 0u && timeline != nullptr) {
        unsigned long long* rec = reinterpret_cast<unsigned long long*>(timeline + (size_t)1100 * kWaves * kTimelinePerWave) + ((size_t)blockIdx.x * kWaves + wave) * 4;
        rec[0] = tl_start; rec[1] = tl_dry; rec[2] = wall_clock64(); rec[3] = tl_trips;
    }
#endif
    unsigned long long v0 = wave_sum(ln.rays_path), v1 = wave_sum(ln.rays_shadow), v2 = wave_sum(ln.hit_nodes);
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)largest_tree, off, 64);
        largest_tree = other > largest_tree ? other : largest_tree;
    }
    if (lane == 0u) {
        atomicAdd(&stats[0], v0);
        atomicAdd(&stats[1], v1);
        atomicAdd(&stats[2], v2);
        if (largest_tree > 1u) atomicMax(&stats[kStatsLargestTree], (unsigned long long)largest_tree);
    }
    if constexpr (COUNT) {
        for (int i = 0; i < PROF_SLOTS; i++) {
            const unsigned long long lanes = wave_sum(prof.cyc[i]);
            if (lane == 0u) atomicAdd(&stats[32 + i], lanes);
        }
        unsigned long long c0 = wave_sum(cnt.box_tests), c1 = wave_sum(cnt.sphere_tests), c2 = wave_sum(cnt.tri_tests), c3 = wave_sum(cnt.rect_tests);
        const unsigned long long c4 = wave_sum(unasked);   // shadow queries answered without a traversal
        if (lane == 0u) {
            atomicAdd(&stats[3], c0);
            atomicAdd(&stats[4], c1);
            atomicAdd(&stats[5], c2);
            atomicAdd(&stats[6], c3);
            for (int i = 0; i < 4; i++) atomicAdd(&stats[7 + i], (unsigned long long)sect[i]);
            for (int i = 0; i < PROF_WAIT_RETURN; i++) atomicAdd(&stats[16 + i], prof.cyc[i]);
            atomicAdd(&stats[16 + 15], (unsigned long long)handed);   // a count, not cycles
            atomicAdd(&stats[kStatsUnasked], c4);
        }
    }
}

// One thread per pixel of this rank: adds the pass's samples in index order to the running sum
// (Game1.cs:295) and, after the last pass, applies gamma and packs (Game1.cs:301-306).  preview_scale > 0 on an
// earlier pass also writes a preview of the picture so far: the running sum scaled by samples / samples done.
__global__ void __launch_bounds__(kBlock) resolve_kernel(RenderArgs ra, const f3* __restrict__ samples, f3* __restrict__ running,
                                                         uint32_t* __restrict__ out, int first_pass, int last_pass, float preview_scale)
{
    uint32_t wp = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
    if (wp >= ra.n_work) return;
    int32_t x, y;
    uint32_t oi;
    if (!work_to_pixel(ra, wp, x, y, oi)) {
        if (ra.layout != 0 && (last_pass || preview_scale > 0.0f)) out[oi] = 0u;  // padding pixel of an edge tile
        return;
    }
    f3 acc = first_pass ? mk3(0.0f, 0.0f, 0.0f) : running[wp];
    acc = accumulate_samples(acc, samples, ra.n_work, wp, ra.sample_count, 1.0f / (float)ra.samples);
    if (last_pass) out[oi] = finish_color(acc);
    else {
        running[wp] = acc;
        if (preview_scale > 0.0f) out[oi] = finish_color(mk3(acc.x * preview_scale, acc.y * preview_scale, acc.z * preview_scale));
    }
}

// The mix-buffer technique's last step (Game1.cs:209-220): buffer1Data[i] = new Color(buffer1Data[i].ToVector3() / 2 +
// buffer[i].ToVector3() / 2), one buffer after the other; buffer1Data starts as default Colors (all zero).
__global__ void __launch_bounds__(kBlock) mix_kernel(uint32_t* __restrict__ frame, const uint32_t* __restrict__ buffer, uint32_t n, int first)
{
    uint32_t i = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
    if (i >= n) return;
    frame[i] = mix_colors(first ? 0u : frame[i], buffer[i]);
}

// Rank 0 of a multi-GPU job: pixel (x, y) of the frame lives in tile t = tx + ty*tiles_x, which
// rank t % world rendered as its (t / world)-th tile.
__global__ void __launch_bounds__(kBlock) assemble_kernel(RenderArgs ra, const uint32_t* __restrict__ gathered, size_t stride, uint32_t* __restrict__ frame)
{
    uint32_t i = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
    uint32_t total = (uint32_t)ra.width * (uint32_t)ra.height;
    if (i >= total) return;
    frame[i] = gathered[gathered_index(ra, i, stride)];
}

// Blocks of a persistent kernel that the device keeps resident at once.  More than that is useless: the surplus only
// starts when the first blocks have run the counter dry, and each still pays its prologue and its statistics atomics.
template <class K>
static uint32_t resident_blocks(K kernel)
{
    int per_cu = 0, dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0) != hipSuccess || per_cu <= 0 || cus <= 0)
        return (uint32_t)kMaxBlocks;
    uint32_t n = (uint32_t)per_cu * (uint32_t)cus;
    return n < (uint32_t)kMaxBlocks ? n : (uint32_t)kMaxBlocks;
}

hipError_t launch_trace(const SceneView& sc, const RenderArgs& ra_in, bool exact, bool counted, float* d_samples,
                        unsigned long long* d_stats, uint32_t* d_work_counter, float* d_timeline, void* d_frame_slab, size_t slab_bytes, hipStream_t stream, int walk_tune,
                        const WavefrontPool* wavefront)
{
    if (ra_in.n_items == 0u) return hipSuccess;
    if (slab_bytes < trace_frame_slab_block_bytes()) return hipErrorInvalidValue;
    if (d_frame_slab == nullptr || d_samples == nullptr || d_stats == nullptr || d_work_counter == nullptr) return hipErrorInvalidValue;
    q4* const slab = reinterpret_cast<q4*>(d_frame_slab);
    RenderArgs ra = ra_in;
    set_item_order(ra, 0u);
    hipError_t e = hipMemsetAsync(d_work_counter, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    // Tuning knobs for experiments; the defaults are the shipped configuration.
    static const int vote = std::getenv("SRT_TRACE_VOTE") ? std::atoi(std::getenv("SRT_TRACE_VOTE")) : kTraceVote;
    static const uint32_t chunk_knob = std::getenv("SRT_CHUNK") ? (uint32_t)std::atoi(std::getenv("SRT_CHUNK")) : 0u;
    static const uint32_t max_blocks = std::getenv("SRT_MAX_BLOCKS") ? (uint32_t)std::atoi(std::getenv("SRT_MAX_BLOCKS")) : (uint32_t)kMaxBlocks;
    static const bool allow_small = !std::getenv("SRT_NO_LDS_SCENE");
    static const bool force_mesh = std::getenv("SRT_FORCE_MESH_KERNEL") != nullptr;   // experiments: the quantum scheduler for every scene
    const bool stackless = !force_mesh && (sc.has_bvh == 0u || (exact && sc.n_tiny != 0u));
    const bool small = allow_small && stackless && sc.n_prims <= kSmallPrims && sc.n_materials <= kSmallMaterials;
    // one entry per kernel instantiation: [mesh][exact][counted][small]
    static uint32_t resident[2][2][2][2] = {};
    uint32_t& fit = resident[stackless ? 0 : 1][exact ? 1 : 0][counted ? 1 : 0][small ? 1 : 0];
    if (fit == 0u) {
#define SRT_FIT(...) fit = resident_blocks(__VA_ARGS__)
        if (!stackless) {
            if (exact) { if (counted) SRT_FIT(mesh_trace_kernel<true, true>); else SRT_FIT(mesh_trace_kernel<true, false>); }
            else { if (counted) SRT_FIT(mesh_trace_kernel<false, true>); else SRT_FIT(mesh_trace_kernel<false, false>); }
        } else if (exact) {
            if (counted) { if (small) SRT_FIT(trace_kernel<true, true, true>); else SRT_FIT(trace_kernel<true, true, false>); }
            else { if (small) SRT_FIT(trace_kernel<true, false, true>); else SRT_FIT(trace_kernel<true, false, false>); }
        } else {
            if (counted) { if (small) SRT_FIT(trace_kernel<false, true, true>); else SRT_FIT(trace_kernel<false, true, false>); }
            else { if (small) SRT_FIT(trace_kernel<false, false, true>); else SRT_FIT(trace_kernel<false, false, false>); }
        }
#undef SRT_FIT
        if (std::getenv("SRT_DEBUG_LAUNCH")) std::fprintf(stderr, "srt: %u resident blocks\n", fit);
    }
    uint32_t blocks = (ra.n_items + kBlock - 1) / kBlock;
    if (blocks > fit) blocks = fit;
    if (blocks > max_blocks) blocks = max_blocks;
    // every block of the grid owns a part of the frame slab: a smaller slab means a smaller grid
    const size_t slab_blocks = slab_bytes / trace_frame_slab_block_bytes();
    if ((size_t)blocks > slab_blocks) blocks = (uint32_t)slab_blocks;
    dim3 grid(blocks), block(kBlock);
    // Items per draw from the global counter: kChunk for big passes; a short pass gets smaller ranges (down to one
    // wavefront's worth), so that every wavefront draws several times and expensive screen regions are spread out.
    uint32_t chunk = chunk_knob;
    if (chunk == 0u) {
        uint32_t share = (ra.n_items / (blocks * (uint32_t)kWaves)) / 8u;
        chunk = share >= kChunk ? kChunk : (share < 64u ? 64u : (share & ~63u));
    }
    static const uint32_t guide_knob = std::getenv("SRT_GUIDE") ? (uint32_t)std::atoi(std::getenv("SRT_GUIDE")) : 2u;
    const uint32_t guide = guide_knob != 0u ? blocks * (uint32_t)kWaves * guide_knob : 1u;   // 1: every draw is `chunk` items
    f3* s = reinterpret_cast<f3*>(d_samples);
    // Scenes whose tree is walked with a stack get the kernel that schedules traversal in quanta.
    // SRT_WALK_* (experiments) replace preset 0 and switch the calibration off
    static const bool walk_knobs = std::getenv("SRT_WALK_VOTE") || std::getenv("SRT_WALK_ROUNDS") || std::getenv("SRT_WALK_STEPS") || std::getenv("SRT_NO_WALK_TUNE");
    static const int walk_vote = std::getenv("SRT_WALK_VOTE") ? std::atoi(std::getenv("SRT_WALK_VOTE")) : kWalkVote;
    static const int walk_rounds = std::getenv("SRT_WALK_ROUNDS") ? std::atoi(std::getenv("SRT_WALK_ROUNDS")) : kWalkRounds;
    static const int walk_steps = std::getenv("SRT_WALK_STEPS") ? std::atoi(std::getenv("SRT_WALK_STEPS")) : kWalkSteps;
    WalkTune tune;
    tune.vote[0] = walk_vote; tune.rounds[0] = walk_rounds; tune.steps[0] = walk_steps;
    tune.vote[1] = kShortVote; tune.rounds[1] = kShortRounds; tune.steps[1] = kShortSteps;
    static const int serve = std::getenv("SRT_MESH_SERVE") ? std::atoi(std::getenv("SRT_MESH_SERVE")) : kServeFirst;
    tune.serve = serve > 0 ? serve : 65;   // 65: never (experiments)
    tune.mode = walk_knobs ? WALK_TUNE_LONG : walk_tune;
    static const bool allow_fork = !std::getenv("SRT_NO_SUBTREE_SHARING");
    float4* timeline = reinterpret_cast<float4*>(d_timeline);
    const int share = allow_fork && (uint32_t)ra.mode == MODE_COLOR ? 1 : 0;   // lanes share the last ray trees of a launch
    if (!stackless) {
        static const int steady_depth = std::getenv("SRT_STEADY_FORK_DEPTH") ? std::atoi(std::getenv("SRT_STEADY_FORK_DEPTH")) : kSteadyForkDepth;
        // Scenes that want short quanta scatter their rays anyway: their samples are enumerated in scattered order, which
        // spreads the deep ray trees of one screen region over many wavefronts (10,000-sphere scene: 518 -> 457 ms); for a
        // single mesh a scattered sample costs 2.6 times a coherent one, so those keep the block order.
        static const bool scatter_knob = std::getenv("SRT_MESH_SCATTER_SAMPLES") != nullptr;
        static const uint32_t scatter_env = scatter_knob ? (uint32_t)std::atoi(std::getenv("SRT_MESH_SCATTER_SAMPLES")) : 0u;
        const uint32_t mesh_scatter = scatter_knob ? scatter_env : (tune.mode == WALK_TUNE_SHORT ? kShortScatterSamples : 0u);
        set_item_order(ra, mesh_scatter);
        // The level-by-level tracer, a chunk of items at a time, and then, chunk by chunk, this kernel standing by for the
        // items of chunks that were given up.
        const uint32_t capacity = wavefront != nullptr && wavefront->mem != nullptr && !exact && wavefront_serves(sc, ra, exact) ? wavefront_capacity(wavefront->bytes) : 0u;
        if (capacity != 0u) {
            const double per_item = wavefront->nodes_per_item > 0.0 ? wavefront->nodes_per_item : 24.0;
            uint64_t room = (uint64_t)((double)capacity / (per_item * 1.3));
            if (room > capacity / 2u) room = capacity / 2u;
            const uint64_t n_chunks = room >= 4096u ? ((uint64_t)ra.n_items + room - 1u) / room : (uint64_t)WAVEFRONT_MAX_CHUNKS + 1u;
            if (n_chunks <= WAVEFRONT_MAX_CHUNKS) {
                const uint32_t per_chunk = (uint32_t)((((uint64_t)ra.n_items + n_chunks - 1u) / n_chunks + 63u) & ~(uint64_t)63u);
                if (tune.mode == WALK_TUNE_CALIBRATE) tune.mode = WALK_TUNE_LONG;
                RenderArgs rw = ra;
                set_item_order(rw, 0u);   // every sample block by block: nothing there needs the deep trees spread out
                if ((e = hipMemsetAsync(wavefront->mem, 0, (size_t)n_chunks * sizeof(uint32_t), stream)) != hipSuccess) return e;
                uint32_t chunk_index = 0u;
                // (64-bit loop variables: a pass may have close to 2^32 items, and item0 + per_chunk must not wrap)
                for (uint64_t at = 0u; at < ra.n_items; at += per_chunk, chunk_index++) {
                    const uint32_t item0 = (uint32_t)at;
                    const uint32_t item1 = ra.n_items - item0 < per_chunk ? ra.n_items : item0 + per_chunk;
                    e = launch_wavefront_chunk(sc, rw, counted, d_samples, d_stats, wavefront->mem, wavefront->bytes, wavefront->far_stacks, chunk_index, item0, item1, stream);
                    if (e != hipSuccess) return e;
                }
                chunk_index = 0u;
                for (uint64_t at = 0u; at < ra.n_items; at += per_chunk, chunk_index++) {
                    const uint32_t item0 = (uint32_t)at;
                    const uint32_t item1 = ra.n_items - item0 < per_chunk ? ra.n_items : item0 + per_chunk;
                    const uint32_t* given_up = wavefront_chunk_flag(wavefront->mem, chunk_index);
                    e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_work_counter), (int)item0, 1, stream);
                    if (e != hipSuccess) return e;
                    RenderArgs rb = rw;     // the same item order: the stand-by kernel must trace exactly the chunk's items
                    rb.n_items = item1;
                    if (counted) hipLaunchKernelGGL((mesh_trace_kernel<false, true>), grid, block, 0, stream, sc, rb, s, d_stats, d_work_counter, timeline, slab, tune, chunk, guide, share, steady_depth, given_up);
                    else hipLaunchKernelGGL((mesh_trace_kernel<false, false>), grid, block, 0, stream, sc, rb, s, d_stats, d_work_counter, timeline, slab, tune, chunk, guide, share, steady_depth, given_up);
                    if ((e = hipGetLastError()) != hipSuccess) return e;
                }
                return hipSuccess;
            }
        }
#define SRT_LAUNCH_MESH(E, C) hipLaunchKernelGGL((mesh_trace_kernel<E, C>), grid, block, 0, stream, sc, ra, s, d_stats, d_work_counter, timeline, slab, tune, chunk, guide, share, steady_depth, (const uint32_t*)nullptr)
        if (exact) { if (counted) SRT_LAUNCH_MESH(true, true); else SRT_LAUNCH_MESH(true, false); }
        else { if (counted) SRT_LAUNCH_MESH(false, true); else SRT_LAUNCH_MESH(false, false); }
#undef SRT_LAUNCH_MESH
        return hipGetLastError();
    }
    // The last samples of the pass in scattered order (with a stride that spreads the 64 lanes of a draw over the whole
    // picture): measured on the default scene at 1080p, a sample costs 0.89 ms in block order and 1.08 ms scattered, but
    // a launch that ends in block order runs 1.8 ms longer than its samples take.
    static const uint32_t scatter_samples = std::getenv("SRT_SCATTER_SAMPLES") ? (uint32_t)std::atoi(std::getenv("SRT_SCATTER_SAMPLES")) : kScatterSamples;
    // short passes are better off with more of them: measured 5.83 -> 5.29 ms at 4 spp (all four), 9.14 -> 8.80 ms at 8 spp (three)
    const bool knob = std::getenv("SRT_SCATTER_SAMPLES") != nullptr;
    set_item_order(ra, knob ? scatter_samples : (ra.sample_count <= 4u ? ra.sample_count : (ra.sample_count <= 11u ? 3u : scatter_samples)));
#define SRT_LAUNCH(E, C, S) hipLaunchKernelGGL((trace_kernel<E, C, S>), grid, block, 0, stream, sc, ra, s, d_stats, d_work_counter, timeline, slab, vote, chunk, share, guide)
    if (exact) {
        if (counted) { if (small) SRT_LAUNCH(true, true, true); else SRT_LAUNCH(true, true, false); }
        else { if (small) SRT_LAUNCH(true, false, true); else SRT_LAUNCH(true, false, false); }
    } else {
        if (counted) { if (small) SRT_LAUNCH(false, true, true); else SRT_LAUNCH(false, true, false); }
        else { if (small) SRT_LAUNCH(false, false, true); else SRT_LAUNCH(false, false, false); }
    }
#undef SRT_LAUNCH
    return hipGetLastError();
}

// Every wavefront of the largest grid owns FRAME_SLOTS levels of 64 frames of 64 bytes.
size_t trace_frame_slab_block_bytes() { return (size_t)kWaves * FRAME_SLOTS * FRAME_SLAB_LEVEL_BYTES; }
size_t trace_frame_slab_bytes() { return (size_t)kMaxBlocks * trace_frame_slab_block_bytes(); }
static_assert((unsigned long long)kMaxBlocks * kWaves * FRAME_SLOTS * FRAME_SLAB_LEVEL_BYTES < (1ull << 32), "frame offsets are 32-bit");
size_t trace_timeline_records() { return (size_t)kMaxBlocks * kWaves; }
size_t trace_timeline_offset() { return (size_t)1100 * kWaves * kTimelinePerWave * sizeof(float4); }
size_t trace_timeline_bytes() { return (size_t)kMaxBlocks * kWaves * kTimelinePerWave * sizeof(float4); }

hipError_t launch_resolve(const RenderArgs& ra, const float* d_samples, float* d_running, uint32_t* d_out, bool first_pass, bool last_pass,
                          float preview_scale, hipStream_t stream)
{
    if (ra.n_work == 0u) return hipSuccess;
    dim3 grid((ra.n_work + kBlock - 1) / kBlock), block(kBlock);
    hipLaunchKernelGGL(resolve_kernel, grid, block, 0, stream, ra, reinterpret_cast<const f3*>(d_samples), reinterpret_cast<f3*>(d_running),
                       d_out, first_pass ? 1 : 0, last_pass ? 1 : 0, preview_scale);
    return hipGetLastError();
}

hipError_t launch_mix(uint32_t* d_frame, const uint32_t* d_buffer, uint32_t n_pixels, bool first, hipStream_t stream)
{
    if (n_pixels == 0u) return hipSuccess;
    hipLaunchKernelGGL(mix_kernel, dim3((n_pixels + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, d_frame, d_buffer, n_pixels, first ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_assemble(const RenderArgs& ra, const uint32_t* d_gathered, size_t stride, uint32_t* d_frame, hipStream_t stream)
{
    uint32_t total = (uint32_t)ra.width * (uint32_t)ra.height;
    dim3 grid((total + kBlock - 1) / kBlock), block(kBlock);
    hipLaunchKernelGGL(assemble_kernel, grid, block, 0, stream, ra, d_gathered, stride, d_frame);
    return hipGetLastError();
}

}  // namespace srt
