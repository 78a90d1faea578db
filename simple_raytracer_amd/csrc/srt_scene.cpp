// srt_scene.cpp -- flattens the host scene into 16-byte device words and builds the BVH.
//
// Two builders:
//  * reference topology (SRT_WORLD_BVHNODE): the tree `new BVHNode(world)` produces
//    (RaytracingClasses.cs:346-426) -- longest axis of the list bounds, ONE bubble pass with
//    BoundingMath.boxCompare (MiscFunctions.cs:88-105), split at Count/2, lists whose bounds
//    also contain the origin (SurfaceList.boundingBox, RaytracingClasses.cs:318-338) --
//    linearised so that a left-to-right walk visits surfaces in the reference's order;
//  * binned SAH (SRT_WORLD_SURFACELIST): the library's own tree with padded boxes; it only
//    has to return every surface a linear scan could hit.
#include "srt_scene.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <system_error>
#include <thread>

namespace srt {
namespace {

struct Box {
    float mn[3], mx[3];
};

// System.Math.Min/Max for float (.NET Core 3.x): NaN propagates, -0 < +0.
inline float net_max(float a, float b)
{
    if (a > b || a != a) return a;
    if (a == b) return std::signbit(a) ? b : a;
    return b;
}
inline float net_min(float a, float b)
{
    if (a < b || a != a) return a;
    if (a == b) return std::signbit(a) ? a : b;
    return b;
}
// BoundingMath.surroundingBox (MiscFunctions.cs:76-86)
inline Box surrounding(const Box& b0, const Box& b1)
{
    Box r;
    for (int k = 0; k < 3; k++) {
        r.mn[k] = net_min(b0.mn[k], b1.mn[k]);
        r.mx[k] = net_max(b0.mx[k], b1.mx[k]);
    }
    return r;
}

// Runs body(first, last) over [0, n) in slices on a few host threads (one thread below 64k items).
template <class F>
void parallel_slices(uint32_t n, F body)
{
    unsigned hw = std::thread::hardware_concurrency();
    uint32_t parts = n < 65536u ? 1u : std::min<uint32_t>(hw ? hw : 1u, 16u);
    if (parts <= 1) { body(0u, n); return; }
    std::vector<std::thread> pool;
    const uint32_t step = (n + parts - 1) / parts;
    uint32_t first = 0;
    try {
        for (; first + step < n; first += step) pool.emplace_back(body, first, first + step);
    } catch (const std::system_error&) {
        // no further thread to be had: this one does the rest
    }
    body(first, n);
    for (auto& t : pool) t.join();
}

inline q4 mkq(float x, float y, float z, float w) { q4 q; q.x = x; q.y = y; q.z = z; q.w = w; return q; }

struct NodeRec {
    Box box;
    uint32_t desc;
};

inline uint32_t leaf_desc(uint32_t first, uint32_t count) { return DESC_LEAF | ((count - 1u) << 27) | first; }

// Surface.boundingBox of each primitive: Tri (RaytracingClasses.cs:649-664), Sphere (:489-495).
void primitive_boxes(const HostScene& in, std::vector<Box>& pb)
{
    const uint32_t nt = in.n_tris(), ns = in.n_spheres();
    pb.resize((size_t)nt + ns);
    parallel_slices(nt, [&](uint32_t first, uint32_t last) {
        for (uint32_t i = first; i < last; i++) {
            const float* p = &in.tri_pos[(size_t)i * 9];
            Box b;
            for (int k = 0; k < 3; k++) {
                b.mx[k] = net_max(net_max(p[k], p[3 + k]), p[6 + k]);
                b.mn[k] = net_min(net_min(p[k], p[3 + k]), p[6 + k]);
            }
            pb[i] = b;
        }
    });
    for (uint32_t i = 0; i < ns; i++) {
        const float* s = &in.sph[(size_t)i * 4];
        Box b;
        for (int k = 0; k < 3; k++) {
            b.mx[k] = s[k] + s[3];
            b.mn[k] = s[k] - s[3];
        }
        pb[nt + i] = b;
    }
}

// ---------------------------------------------------------------------------------------
// Reference-topology builder
// ---------------------------------------------------------------------------------------
struct ReferenceBuilder {
    const std::vector<Box>& pb;
    std::vector<uint32_t> ids;        // world.surfaces, permuted in place by the bubble passes
    std::vector<NodeRec> nodes;
    std::vector<uint32_t> leaf_order; // surfaces in left-to-right leaf order
    uint32_t max_depth = 0;

    explicit ReferenceBuilder(const std::vector<Box>& boxes) : pb(boxes) {}

    // SurfaceList.boundingBox: the union runs on every iteration, the first time against a
    // zero box, so the result always contains the origin.
    Box list_bounds(uint32_t lo, uint32_t hi) const
    {
        Box bounds = pb[ids[lo]];
        Box temp;
        for (int k = 0; k < 3; k++) temp.mn[k] = temp.mx[k] = 0.0f;
        for (uint32_t i = lo; i < hi; i++) {
            if (i != lo) temp = bounds;
            bounds = surrounding(pb[ids[i]], temp);
        }
        return bounds;
    }
    // BoundingMath.boxCompare: "centroid" = min + (min - max / 2)
    bool box_compare(uint32_t a, uint32_t b, int axis) const
    {
        float ac = pb[a].mn[axis] + (pb[a].mn[axis] - pb[a].mx[axis] * (1.0f / 2.0f));
        float bc = pb[b].mn[axis] + (pb[b].mn[axis] - pb[b].mx[axis] * (1.0f / 2.0f));
        return ac < bc;
    }
    static int pick_axis(const Box& lb)
    {
        float sx = std::fabs(lb.mx[0] - lb.mn[0]), sy = std::fabs(lb.mx[1] - lb.mn[1]), sz = std::fabs(lb.mx[2] - lb.mn[2]);
        if (sx > net_max(sy, sz)) return 0;
        if (sy > net_max(sx, sz)) return 1;
        if (sz > net_max(sx, sy)) return 2;
        return 0;
    }
    Box build(uint32_t lo, uint32_t hi, const Box& lb, uint32_t slot, uint32_t depth)
    {
        if (depth > max_depth) max_depth = depth;
        const uint32_t n = hi - lo;
        const int axis = pick_axis(lb);
        Box box;
        if (n == 1) {
            // left = right = the surface; the node keeps the list's (origin-inflated) bounds
            uint32_t first = (uint32_t)leaf_order.size();
            leaf_order.push_back(ids[lo]);
            box = lb;
            nodes[slot].desc = leaf_desc(first, 1);
        } else if (n == 2) {
            uint32_t l = ids[lo], r = ids[lo + 1];
            if (box_compare(ids[lo], ids[lo + 1], axis)) { l = ids[lo + 1]; r = ids[lo]; }
            uint32_t first = (uint32_t)leaf_order.size();
            leaf_order.push_back(l);
            leaf_order.push_back(r);
            box = surrounding(pb[r], pb[l]);
            nodes[slot].desc = leaf_desc(first, 2);
        } else {
            for (uint32_t i = lo; i + 1 < hi; i++)
                if (box_compare(ids[i], ids[i + 1], axis)) std::swap(ids[i], ids[i + 1]);
            uint32_t mid = lo + n / 2;
            Box lbl = list_bounds(lo, mid), lbr = list_bounds(mid, hi);
            uint32_t pair = (uint32_t)nodes.size();
            nodes.resize(nodes.size() + 2);
            Box bl = build(lo, mid, lbl, pair, depth + 1);
            Box br = build(mid, hi, lbr, pair + 1, depth + 1);
            box = surrounding(br, bl);
            nodes[slot].desc = pair;
        }
        nodes[slot].box = box;
        return box;
    }
    // `insertion`: world.surfaces before the constructor runs (empty: surface i at position i)
    void run(const std::vector<uint32_t>& insertion)
    {
        const uint32_t n = (uint32_t)pb.size();
        ids.resize(n);
        for (uint32_t i = 0; i < n; i++) ids[i] = insertion.empty() ? i : insertion[i];
        nodes.assign(2, NodeRec());  // root + one unused slot so that sibling pairs start at even indices
        std::memset(nodes.data(), 0, sizeof(NodeRec) * 2);
        leaf_order.reserve(n);
        Box wb = list_bounds(0, n);  // world.Init() (Game1.cs:136)
        build(0, n, wb, 0, 1);
    }
};

// ---------------------------------------------------------------------------------------
// Binned SAH builder
// ---------------------------------------------------------------------------------------
// The build runs on several host threads and still produces the tree (node order, primitive order) that
// one thread would: a node with at least `fork_min` primitives builds its left half in another thread.  No
// thread allocates: a subtree over positions [lo, hi) of `ids` leaves its primitives, in leaf order, in those
// positions, and the node that splits at position `mid` keeps its two children in entries 2 * mid - 2 and
// 2 * mid - 1 of `scratch` (every split position belongs to exactly one node).  `compact` then writes the
// nodes out in the depth-first order in which a single thread would have allocated them.
// Each thread works on its own range of `ids`; the primitive boxes and centroids are only read.
struct SahBuilder {
    enum { BINS = 16, LEAF_MAX = 4, DEPTH_LIMIT = 30 };
    enum : uint32_t { FORK_FLOOR = 16384, FORK_SHARE = 16 };
    const std::vector<Box>& pb;
    std::vector<uint32_t> ids;                  // after run(): the primitives in leaf order
    std::vector<float> cx[3];
    std::vector<NodeRec> scratch;
    std::vector<NodeRec> nodes;
    std::vector<uint32_t>& leaf_order = ids;
    uint32_t max_depth = 0;
    uint32_t fork_min = 0xFFFFFFFFu;

    explicit SahBuilder(const std::vector<Box>& boxes) : pb(boxes) {}

    // minimum / maximum that keep the accumulator when the other value is a NaN, as fmin / fmax do, without the
    // library call those cost in a build that honours NaNs
    static float lower(float acc, float v) { return v < acc ? v : acc; }
    static float upper(float acc, float v) { return v > acc ? v : acc; }
    static void grow(Box& b, const Box& o)
    {
        for (int k = 0; k < 3; k++) {
            b.mn[k] = lower(b.mn[k], o.mn[k]);
            b.mx[k] = upper(b.mx[k], o.mx[k]);
        }
    }
    static Box empty_box()
    {
        Box b;
        for (int k = 0; k < 3; k++) { b.mn[k] = INFINITY; b.mx[k] = -INFINITY; }
        return b;
    }
    static float half_area(const Box& b)
    {
        float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
        if (!(dx >= 0.0f) || !(dy >= 0.0f) || !(dz >= 0.0f)) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
    static uint32_t ceil_log2(uint32_t n)
    {
        uint32_t l = 0;
        while ((1u << l) < n) l++;
        return l;
    }
    static int bin_of(float c, float lo_c, float scale)
    {
        int b = (int)((c - lo_c) * scale);
        return b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
    }
    void make_leaf(uint32_t lo, uint32_t hi, uint32_t slot)
    {
        // keep insertion order inside a leaf; ties are settled by the order rank anyway
        std::sort(ids.begin() + lo, ids.begin() + hi);
        scratch[slot].desc = leaf_desc(lo, hi - lo);
    }
    void build(uint32_t lo, uint32_t hi, uint32_t slot, uint32_t depth)
    {
        const uint32_t n = hi - lo;
        Box box = empty_box(), cb = empty_box();
        for (uint32_t i = lo; i < hi; i++) {
            grow(box, pb[ids[i]]);
            for (int k = 0; k < 3; k++) {
                float c = cx[k][ids[i]];
                cb.mn[k] = lower(cb.mn[k], c);
                cb.mx[k] = upper(cb.mx[k], c);
            }
        }
        scratch[slot].box = box;
        if (n <= 1) { make_leaf(lo, hi, slot); return; }

        // levels still available below this node; a balanced split needs ceil_log2(n / LEAF_MAX)
        const bool must_balance = depth + ceil_log2((n + LEAF_MAX - 1) / LEAF_MAX) + 1 >= (uint32_t)DEPTH_LIMIT;
        uint32_t mid = 0;
        bool split = false;
        if (!must_balance) {
            // one pass over the primitives fills the bins of all three axes
            Box bb[3][BINS];
            uint32_t bc[3][BINS];
            float lo_c[3], scale[3];
            bool live[3];
            for (int axis = 0; axis < 3; axis++) {
                float ext = cb.mx[axis] - cb.mn[axis];
                live[axis] = ext > 0.0f;
                lo_c[axis] = cb.mn[axis];
                scale[axis] = live[axis] ? (float)BINS / ext : 0.0f;
                for (int b = 0; b < BINS; b++) { bb[axis][b] = empty_box(); bc[axis][b] = 0; }
            }
            for (uint32_t i = lo; i < hi; i++) {
                const uint32_t id = ids[i];
                const Box& pbox = pb[id];
                for (int axis = 0; axis < 3; axis++) {
                    if (!live[axis]) continue;
                    int b = bin_of(cx[axis][id], lo_c[axis], scale[axis]);
                    bc[axis][b]++;
                    grow(bb[axis][b], pbox);
                }
            }
            float best_cost = INFINITY;
            int best_axis = -1, best_bin = -1;
            for (int axis = 0; axis < 3; axis++) {
                if (!live[axis]) continue;
                float right_area[BINS];
                uint32_t right_cnt[BINS];
                Box acc = empty_box();
                uint32_t cnt = 0;
                for (int b = BINS - 1; b > 0; b--) {
                    grow(acc, bb[axis][b]);
                    cnt += bc[axis][b];
                    right_area[b] = half_area(acc);
                    right_cnt[b] = cnt;
                }
                acc = empty_box();
                cnt = 0;
                for (int b = 0; b < BINS - 1; b++) {
                    grow(acc, bb[axis][b]);
                    cnt += bc[axis][b];
                    if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                    float cost = half_area(acc) * (float)cnt + right_area[b + 1] * (float)right_cnt[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
                }
            }
            if (best_axis >= 0) {
                float leaf_cost = half_area(box) * (float)n;
                float split_cost = 1.0f * half_area(box) + best_cost;  // one box-pair visit ~ one primitive test
                if (n > (uint32_t)LEAF_MAX || split_cost < leaf_cost) {
                    const float l = lo_c[best_axis], sc = scale[best_axis];
                    const std::vector<float>& c = cx[best_axis];
                    auto it = std::partition(ids.begin() + lo, ids.begin() + hi,
                                             [&](uint32_t id) { return bin_of(c[id], l, sc) <= best_bin; });
                    mid = (uint32_t)(it - ids.begin());
                    split = mid > lo && mid < hi;
                }
            }
        }
        if (!split) {
            if (n <= (uint32_t)LEAF_MAX) { make_leaf(lo, hi, slot); return; }
            // object median along the widest centroid axis (also the forced balanced split)
            int axis = 0;
            float e0 = cb.mx[0] - cb.mn[0], e1 = cb.mx[1] - cb.mn[1], e2 = cb.mx[2] - cb.mn[2];
            if (e1 > e0 && e1 >= e2) axis = 1;
            else if (e2 > e0 && e2 > e1) axis = 2;
            mid = lo + n / 2;
            std::nth_element(ids.begin() + lo, ids.begin() + mid, ids.begin() + hi,
                             [&](uint32_t a, uint32_t b) { return cx[axis][a] < cx[axis][b] || (cx[axis][a] == cx[axis][b] && a < b); });
        }
        const uint32_t pair = 2u * mid - 2u;
        scratch[slot].desc = pair;
        if (n >= fork_min) {
            std::thread left;
            try {
                left = std::thread([this, lo, mid, pair, depth] { build(lo, mid, pair, depth + 1); });
            } catch (const std::system_error&) {
                // no thread to be had: this one does both halves
            }
            if (left.joinable()) {
                struct Join { std::thread& th; ~Join() { th.join(); } } join{left};
                build(mid, hi, pair + 1, depth + 1);
                return;
            }
        }
        build(lo, mid, pair, depth + 1);
        build(mid, hi, pair + 1, depth + 1);
    }
    // Writes node `from` of the scratch tree, and everything below it, to `slot` of the final tree.
    void compact(uint32_t from, uint32_t slot, uint32_t depth)
    {
        const NodeRec& nd = scratch[from];
        if (depth > max_depth) max_depth = depth;
        nodes[slot].box = nd.box;
        if (nd.desc & DESC_LEAF) {
            nodes[slot].desc = nd.desc;
        } else {
            uint32_t pair = (uint32_t)nodes.size();
            nodes.resize(nodes.size() + 2);
            nodes[slot].desc = pair;
            compact(nd.desc, pair, depth + 1);
            compact(nd.desc + 1, pair + 1, depth + 1);
        }
    }
    // `members`: the primitives (indices into the boxes) the tree is built over, at least one
    void run(std::vector<uint32_t> members)
    {
        ids.swap(members);
        const uint32_t n = (uint32_t)ids.size();
        for (int k = 0; k < 3; k++) cx[k].resize(pb.size());
        parallel_slices(n, [&](uint32_t first, uint32_t last) {
            for (uint32_t i = first; i < last; i++) {
                const uint32_t id = ids[i];
                for (int k = 0; k < 3; k++) cx[k][id] = 0.5f * pb[id].mn[k] + 0.5f * pb[id].mx[k];
            }
        });
        // at most 2 * FORK_SHARE - 1 nodes hold n / FORK_SHARE primitives or more: that bounds the threads
        fork_min = std::max<uint32_t>(FORK_FLOOR, n / FORK_SHARE);
        if (const char* e = std::getenv("SRT_BUILD_FORK_MIN")) {
            long v = std::atol(e);
            fork_min = v > 0 ? (uint32_t)std::max<long>(v, (long)(n / 64)) : 0xFFFFFFFFu;   // 0: one thread
        }
        scratch.resize((size_t)2 * n);
        const uint32_t root = 2u * n - 2u;      // split positions are 1 .. n - 1: entries 0 .. 2n - 3 hold the pairs
        build(0, n, root, 1);
        nodes.reserve((size_t)n);
        nodes.assign(2, NodeRec());
        std::memset(nodes.data(), 0, sizeof(NodeRec) * 2);
        compact(root, 0, 1);
        std::vector<NodeRec>().swap(scratch);
        // Pad every box: the slab test and the primitive tests round differently, and a ray
        // that a primitive test accepts must never be culled by a box.
        float ext = 0.0f;
        for (int k = 0; k < 3; k++) {
            ext = std::fmax(ext, nodes[0].box.mx[k] - nodes[0].box.mn[k]);
            ext = std::fmax(ext, std::fmax(std::fabs(nodes[0].box.mx[k]), std::fabs(nodes[0].box.mn[k])));
        }
        float pad = ext * (1.0f / 65536.0f);
        if (!(pad > 0.0f)) pad = 1e-6f;
        for (auto& nd : nodes)
            for (int k = 0; k < 3; k++) { nd.box.mn[k] -= pad; nd.box.mx[k] += pad; }
    }
};


// ---------------------------------------------------------------------------------------
// Wide nodes: the SAH tree collapsed to up to four children per node, child boxes quantised to
// 8 bits per coordinate inside the box of the node, 64 bytes per node.  One visit then costs a
// lane four 16-byte loads for four boxes (the binary tree: four loads for two boxes), and a
// query visits about half as many nodes -- the texture-address unit, which handles one lane's
// load at a time, is what limits the traversal of a large mesh.
//   word 0: origin.xyz (the lower corner of the children's union), ex | ey << 8 | ez << 16 | children << 24
//           (e = biased float exponent of the grid step along the axis: step = 2^(e - 127))
//   word 1: lo.x of the 4 children (one byte each), lo.y, lo.z, hi.x
//   word 2: hi.y, hi.z, child 0, child 1
//   word 3: child 2, child 3, -, -
// Child descriptors: index of a wide node, or DESC_LEAF | (count - 1) << 27 | first primitive.
// A decoded box (origin + step * byte, evaluated in float exactly as the kernel does) always
// contains the padded box it stands for.
// The traversal pushes the children it does not enter, at most (children - 1) per node.  `budget` is the number
// of stack entries still free on the way down; a node only takes as many children as leave every child at least
// the height of its binary subtree (a binary subtree of height h can always be emitted with one push per level),
// so no traversal ever needs more than STACK_DEPTH entries.
// ---------------------------------------------------------------------------------------
struct WideBuilder {
    const std::vector<NodeRec>& n2;
    std::vector<uint32_t> height;     // of the binary subtree under each binary node (a leaf: 0)
    std::vector<q4> out;
    uint32_t max_depth = 0;

    explicit WideBuilder(const std::vector<NodeRec>& binary) : n2(binary) {}

    static float step_of(uint32_t e) { return bitsf(e << 23); }
    static float decode(float origin, uint32_t e, uint32_t q) { return origin + step_of(e) * (float)q; }

    uint32_t measure(uint32_t idx)
    {
        uint32_t h = 0;
        if (!(n2[idx].desc & DESC_LEAF)) {
            uint32_t a = measure(n2[idx].desc), b = measure(n2[idx].desc + 1);
            h = 1 + (a > b ? a : b);
        }
        height[idx] = h;
        return h;
    }

    void emit(uint32_t idx, uint32_t slot, uint32_t budget, uint32_t depth)
    {
        if (depth > max_depth) max_depth = depth;
        uint32_t kids[4];
        uint32_t nk = 0;
        if (n2[idx].desc & DESC_LEAF) {
            kids[nk++] = idx;
        } else {
            kids[0] = n2[idx].desc; kids[1] = n2[idx].desc + 1; nk = 2;
            while (nk < 4) {
                // expand the inner child with the largest box, if the stack budget of every child still holds its height
                int pick = -1;
                float area = -1.0f;
                for (uint32_t i = 0; i < nk; i++) {
                    if (n2[kids[i]].desc & DESC_LEAF) continue;
                    float a = SahBuilder::half_area(n2[kids[i]].box);
                    if (a > area) { area = a; pick = (int)i; }
                }
                if (pick < 0) break;
                uint32_t c = n2[kids[pick]].desc;
                uint32_t tallest = 0;
                for (uint32_t i = 0; i < nk; i++) {
                    uint32_t h = (int)i == pick ? (height[c] > height[c + 1] ? height[c] : height[c + 1]) : height[kids[i]];
                    if (h > tallest) tallest = h;
                }
                if (nk + tallest > budget) break;        // (nk + 1 - 1) pushes here + the tallest child's worst case
                for (uint32_t i = nk; i > (uint32_t)pick + 1; i--) kids[i] = kids[i - 1];
                kids[pick] = c; kids[pick + 1] = c + 1;
                nk++;
            }
        }
        // the grid: origin = lower corner of the children's union, step = the power of two that spans it in 255 steps
        Box u = SahBuilder::empty_box();
        for (uint32_t i = 0; i < nk; i++) SahBuilder::grow(u, n2[kids[i]].box);
        uint32_t e[3], qlo[3][4], qhi[3][4];
        for (int k = 0; k < 3; k++) {
            float ext = u.mx[k] - u.mn[k];
            int ex = 0;
            std::frexp(ext / 255.0f, &ex);               // ext / 255 = m * 2^ex with m in [0.5, 1)  =>  2^ex >= ext / 255
            int be = ex + 127;
            if (!(ext > 0.0f) || be < 1) be = 1;
            for (;; be++) {
                bool fits = true;
                for (uint32_t i = 0; i < nk && fits; i++) {
                    const Box& b = n2[kids[i]].box;
                    float st = step_of((uint32_t)be);
                    long lo = (long)std::floor((b.mn[k] - u.mn[k]) / st), hi = (long)std::ceil((b.mx[k] - u.mn[k]) / st);
                    if (lo < 0) lo = 0;
                    if (hi < 0) hi = 0;
                    if (lo > 255) lo = 255;
                    while (lo > 0 && decode(u.mn[k], (uint32_t)be, (uint32_t)lo) > b.mn[k]) lo--;
                    while (hi <= 255 && decode(u.mn[k], (uint32_t)be, (uint32_t)hi) < b.mx[k]) hi++;
                    if (hi > 255 || decode(u.mn[k], (uint32_t)be, (uint32_t)lo) > b.mn[k]) { fits = false; break; }
                    qlo[k][i] = (uint32_t)lo; qhi[k][i] = (uint32_t)hi;
                }
                if (fits) break;
                if (be >= 254) {
                    // boxes that are not finite numbers cannot be put on a grid: every child spans the whole cell range
                    // (for a NaN origin the decoded planes are NaN, which the slab test never culls on)
                    for (uint32_t i = 0; i < nk; i++) { qlo[k][i] = 0; qhi[k][i] = 255; }
                    break;
                }
            }
            e[k] = (uint32_t)be;
            for (uint32_t i = nk; i < 4; i++) { qlo[k][i] = 0; qhi[k][i] = 0; }
        }
        auto pack = [](const uint32_t q[4]) { return q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24); };
        uint32_t child[4] = {0u, 0u, 0u, 0u};
        for (uint32_t i = 0; i < nk; i++) {
            const NodeRec& c = n2[kids[i]];
            if (c.desc & DESC_LEAF) { child[i] = c.desc; continue; }
            uint32_t at = (uint32_t)(out.size() / 4);
            out.resize(out.size() + 4);
            child[i] = at;
            emit(kids[i], at, budget - (nk - 1u), depth + 1);
        }
        q4* w = &out[(size_t)slot * 4];
        w[0] = mkq(u.mn[0], u.mn[1], u.mn[2], bitsf(e[0] | (e[1] << 8) | (e[2] << 16) | (nk << 24)));
        w[1] = mkq(bitsf(pack(qlo[0])), bitsf(pack(qlo[1])), bitsf(pack(qlo[2])), bitsf(pack(qhi[0])));
        w[2] = mkq(bitsf(pack(qhi[1])), bitsf(pack(qhi[2])), bitsf(child[0]), bitsf(child[1]));
        w[3] = mkq(bitsf(child[2]), bitsf(child[3]), 0.0f, 0.0f);
    }
    void run()
    {
        height.assign(n2.size(), 0u);
        measure(0);
        out.clear();
        out.resize(4);
        emit(0, 0, (uint32_t)STACK_DEPTH, 1);
    }
};

// primitive record for global surface index `g`
// Writes the three words of surface `g` to `prims`.
void emit_prim(const HostScene& in, uint32_t g, uint32_t order, q4* prims)
{
    const uint32_t nt = in.n_tris(), ns = in.n_spheres();
    if (g < nt) {
        const float* p = &in.tri_pos[(size_t)g * 9];
        // Tri constructor (RaytracingClasses.cs:561-562): AB = B - A, AC = C - A
        float ab[3] = {p[3] - p[0], p[4] - p[1], p[5] - p[2]};
        float ac[3] = {p[6] - p[0], p[7] - p[1], p[8] - p[2]};
        uint8_t fl = in.tri_flags[g];
        uint32_t info = PRIM_TRI | ((fl & 1u) ? (uint32_t)INFO_FLIP : 0u) | ((fl & 2u) ? (uint32_t)INFO_BLEND : 0u) | (in.tri_mat[g] << 8);
        prims[0] = mkq(p[0], p[1], p[2], ab[0]);
        prims[1] = mkq(ab[1], ab[2], ac[0], ac[1]);
        prims[2] = mkq(ac[2], bitsf(info), bitsf(order), bitsf(g));
    } else if (g < nt + ns) {
        uint32_t i = g - nt;
        const float* s = &in.sph[(size_t)i * 4];
        uint32_t info = PRIM_SPHERE | (in.sph_mat[i] << 8);
        prims[0] = mkq(s[0], s[1], s[2], s[3]);
        prims[1] = mkq(0, 0, 0, 0);
        prims[2] = mkq(0, bitsf(info), bitsf(order), 0);
    } else {
        uint32_t i = g - nt - ns;
        const float* r = &in.rect[(size_t)i * 5];
        uint32_t info = PRIM_RECT | (in.rect_mat[i] << 8);
        prims[0] = mkq(r[0], r[1], r[2], r[3]);
        prims[1] = mkq(r[4], 0, 0, 0);
        prims[2] = mkq(0, bitsf(info), bitsf(order), 0);
    }
}

// Depth-first (pre)order copy of a small tree with skip indices, for the wave-uniform walk.
void emit_tiny(const std::vector<NodeRec>& nodes, uint32_t idx, std::vector<q4>& out)
{
    const NodeRec& n = nodes[idx];
    size_t at = out.size();
    out.push_back(mkq(n.box.mn[0], n.box.mn[1], n.box.mn[2], bitsf(n.desc)));
    out.push_back(mkq(n.box.mx[0], n.box.mx[1], n.box.mx[2], 0.0f));
    if (!(n.desc & DESC_LEAF)) {
        emit_tiny(nodes, n.desc, out);
        emit_tiny(nodes, n.desc + 1, out);
    }
    out[at + 1].w = bitsf((uint32_t)(out.size() / 2));  // first node after this subtree
}

void emit_nodes(const std::vector<NodeRec>& nodes, std::vector<q4>& out)
{
    out.reserve(nodes.size() * 2);
    for (const NodeRec& n : nodes) {
        out.push_back(mkq(n.box.mn[0], n.box.mn[1], n.box.mn[2], bitsf(n.desc)));
        out.push_back(mkq(n.box.mx[0], n.box.mx[1], n.box.mx[2], 0.0f));
    }
}

}  // namespace

int host_set_materials(HostScene& s, const srt_material* m, uint32_t n, std::string& err)
{
    if (n && !m) { err = "materials is NULL"; return SRT_E_INVALID; }
    s.materials.assign(m, m + n);
    return SRT_OK;
}

int host_set_spheres(HostScene& s, const float* cx, const float* cy, const float* cz, const float* r, const uint32_t* mat, uint32_t n, std::string& err)
{
    if (n && (!cx || !cy || !cz || !r || !mat)) { err = "sphere array is NULL"; return SRT_E_INVALID; }
    s.sph.resize((size_t)n * 4);
    s.sph_mat.assign(mat, mat + n);
    for (uint32_t i = 0; i < n; i++) {
        s.sph[(size_t)i * 4 + 0] = cx[i]; s.sph[(size_t)i * 4 + 1] = cy[i];
        s.sph[(size_t)i * 4 + 2] = cz[i]; s.sph[(size_t)i * 4 + 3] = r[i];
    }
    return SRT_OK;
}

int host_set_rects(HostScene& s, const float* x1, const float* x2, const float* y1, const float* y2, const float* k, const uint32_t* mat, uint32_t n, std::string& err)
{
    if (n && (!x1 || !x2 || !y1 || !y2 || !k || !mat)) { err = "rect array is NULL"; return SRT_E_INVALID; }
    s.rect.resize((size_t)n * 5);
    s.rect_mat.assign(mat, mat + n);
    for (uint32_t i = 0; i < n; i++) {
        float* d = &s.rect[(size_t)i * 5];
        d[0] = x1[i]; d[1] = x2[i]; d[2] = y1[i]; d[3] = y2[i]; d[4] = k[i];
    }
    return SRT_OK;
}

int host_set_triangles(HostScene& s, const float* pos9, const float* nrm9, const uint32_t* mat, const uint8_t* flags, uint32_t n, std::string& err)
{
    if (n && (!pos9 || !nrm9 || !mat)) { err = "triangle array is NULL"; return SRT_E_INVALID; }
    s.tri_pos.assign(pos9, pos9 + (size_t)n * 9);
    s.tri_nrm.assign(nrm9, nrm9 + (size_t)n * 9);
    s.tri_mat.assign(mat, mat + n);
    if (flags) s.tri_flags.assign(flags, flags + n);
    else s.tri_flags.assign(n, (uint8_t)2);  // flip = false, blend = true (ModelLoader.cs:60)
    return SRT_OK;
}

int host_set_surface_order(HostScene& s, const uint8_t* type, const uint32_t* index, uint32_t n, std::string& err)
{
    if (n && (!type || !index)) { err = "surface order array is NULL"; return SRT_E_INVALID; }
    // kept as (type, index) pairs until the commit: the surface arrays may still change
    s.order_type.assign(type, type + n);
    s.order_index.assign(index, index + n);
    return SRT_OK;
}

int host_set_lights(HostScene& s, const srt_point_light* l, uint32_t n, std::string& err)
{
    if (n && !l) { err = "lights is NULL"; return SRT_E_INVALID; }
    if (n > SRT_MAX_LIGHTS) { err = "too many lights"; return SRT_E_LIMIT; }
    s.lights.assign(l, l + n);
    return SRT_OK;
}

static int check_params(const srt_render_params* p, int& tw, int& th, int& world, std::string* err)
{
    auto fail = [&](int code, const char* msg) { if (err) *err = msg; return code; };
    if (!p) return fail(SRT_E_INVALID, "params is NULL");
    if (p->width <= 0 || p->height <= 0 || p->samples <= 0) return fail(SRT_E_INVALID, "width, height and samples must be positive");
    if ((int64_t)p->width * p->height >= (int64_t)1 << 31) return fail(SRT_E_LIMIT, "image has 2^31 pixels or more");
    if (p->max_depth < 0 || p->max_depth > SRT_MAX_DEPTH) return fail(SRT_E_LIMIT, "max_depth must be within 0..SRT_MAX_DEPTH");
    if (p->color_clamp != p->color_clamp) return fail(SRT_E_INVALID, "color_clamp is NaN");
    if (p->color_clamp < 0) return fail(SRT_E_REFERENCE, "Math.Clamp(x, 0, clamp) throws ArgumentException when clamp < 0");
    if (p->mode < SRT_MODE_COLOR || p->mode > SRT_MODE_REFLECTION) return fail(SRT_E_INVALID, "unknown mode");
    if (p->layout != SRT_LAYOUT_FRAME && p->layout != SRT_LAYOUT_TILES) return fail(SRT_E_INVALID, "unknown layout");
    tw = p->tile_w ? p->tile_w : 32;
    th = p->tile_h ? p->tile_h : 32;
    if (tw <= 0 || th <= 0 || (tw & 7) || (th & 7) || tw > 4096 || th > 4096) return fail(SRT_E_INVALID, "tile size must be a positive multiple of 8");
    world = p->world_size > 1 ? p->world_size : 1;
    if (p->rank < 0 || p->rank >= world) return fail(SRT_E_INVALID, "rank out of range");
    // a rank of several writes only its own tiles: a row-major frame would keep stale pixels everywhere else
    if (world > 1 && p->layout == SRT_LAYOUT_FRAME) return fail(SRT_E_INVALID, "a rank of several renders into SRT_LAYOUT_TILES; srt_assemble_tiles makes the frame");
    return SRT_OK;
}

static int64_t local_tiles(const srt_render_params* p, int tw, int th, int world)
{
    int64_t tx = (p->width + tw - 1) / tw, ty = (p->height + th - 1) / th;
    int64_t total = tx * ty;
    return total > p->rank ? (total - p->rank + world - 1) / world : 0;
}

int64_t output_pixels(const srt_render_params* p)
{
    int tw, th, world;
    int rc = check_params(p, tw, th, world, nullptr);
    if (rc != SRT_OK) return rc;
    if (p->layout == SRT_LAYOUT_FRAME) return (int64_t)p->width * p->height;
    return local_tiles(p, tw, th, world) * tw * th;
}

int make_render_args(const srt_render_params* p, const Camera& cam, RenderArgs& ra, std::string& err)
{
    int tw, th, world;
    int rc = check_params(p, tw, th, world, &err);
    if (rc != SRT_OK) return rc;
    ra.cam = cam;
    ra.width = p->width; ra.height = p->height; ra.samples = p->samples; ra.max_depth = p->max_depth;
    ra.color_clamp = p->color_clamp; ra.seed = p->seed; ra.mode = p->mode; ra.layout = p->layout;
    ra.tile_w = tw; ra.tile_h = th; ra.rank = p->rank; ra.world_size = world;
    ra.tiles_x = (p->width + tw - 1) / tw;
    ra.tiles_y = (p->height + th - 1) / th;
    int64_t work = local_tiles(p, tw, th, world) * tw * th;
    if (work >= ((int64_t)1 << 32)) { err = "more than 2^32 work items"; return SRT_E_LIMIT; }
    ra.n_work = (uint32_t)work;
    ra.sample_first = 0u;
    ra.sample_count = (uint32_t)p->samples;
    ra.n_items = 0u;  // set per pass by the launcher
    ra.coherent = ra.sample_count; ra.scatter = 0u;  // set per pass by the launcher
    return SRT_OK;
}

int flatten_scene(const HostScene& in, int world_kind, FlatScene& out, std::string& err)
{
    out = FlatScene();
    const uint32_t nt = in.n_tris(), ns = in.n_spheres(), nr = in.n_rects();
    const uint64_t total = (uint64_t)nt + ns + nr;
    if (total >= (1u << 27)) { err = "more than 2^27 surfaces"; return SRT_E_LIMIT; }
    if (in.materials.size() >= (1u << 24)) { err = "more than 2^24 materials"; return SRT_E_LIMIT; }
    if (in.lights.size() > SRT_MAX_LIGHTS) { err = "too many lights"; return SRT_E_LIMIT; }
    const uint32_t nm = (uint32_t)in.materials.size();
    for (uint32_t m : in.tri_mat) if (m >= nm) { err = "triangle material index out of range"; return SRT_E_INVALID; }
    for (uint32_t m : in.sph_mat) if (m >= nm) { err = "sphere material index out of range"; return SRT_E_INVALID; }
    for (uint32_t m : in.rect_mat) if (m >= nm) { err = "rect material index out of range"; return SRT_E_INVALID; }
    if (world_kind != SRT_WORLD_BVHNODE && world_kind != SRT_WORLD_SURFACELIST) { err = "unknown world kind"; return SRT_E_INVALID; }
    if (world_kind == SRT_WORLD_BVHNODE && nr > 0) {
        err = "RectAxis has no boundingBox(): new BVHNode(world) throws NullReferenceException in the reference; use SRT_WORLD_SURFACELIST";
        return SRT_E_REFERENCE;
    }
    // world.surfaces: `insertion[r]` = global index of the r-th surface added, `rank[g]` = position of surface g
    // (both empty for the default order, in which they are the identity)
    std::vector<uint32_t> insertion, rank;
    if (!in.order_type.empty()) {
        if (in.order_type.size() != total) { err = "the surface order names " + std::to_string(in.order_type.size()) + " surfaces, the world has " + std::to_string(total); return SRT_E_INVALID; }
        insertion.resize((size_t)total);
        rank.assign((size_t)total, 0xFFFFFFFFu);
        for (uint32_t r = 0; r < (uint32_t)total; r++) {
            const uint32_t t = in.order_type[r], i = in.order_index[r];
            const uint32_t count = t == PRIM_TRI ? nt : (t == PRIM_SPHERE ? ns : (t == PRIM_RECT ? nr : 0u));
            if (i >= count) { err = "surface order entry " + std::to_string(r) + " names a surface that does not exist"; return SRT_E_INVALID; }
            const uint32_t g = t == PRIM_TRI ? i : (t == PRIM_SPHERE ? nt + i : nt + ns + i);
            if (rank[g] != 0xFFFFFFFFu) { err = "surface order entry " + std::to_string(r) + " names a surface for the second time"; return SRT_E_INVALID; }
            rank[g] = r;
            insertion[r] = g;
        }
    }
    auto rank_of = [&](uint32_t g) { return rank.empty() ? g : rank[g]; };

    for (const srt_material& m : in.materials) {
        out.materials.push_back(mkq(m.albedo[0], m.albedo[1], m.albedo[2], m.metalness));
        out.materials.push_back(mkq(m.smoothness, m.emission[0], m.emission[1], m.emission[2]));
    }
    for (const srt_point_light& l : in.lights) {
        out.lights.push_back(mkq(l.position[0], l.position[1], l.position[2], l.brightness));
        out.lights.push_back(mkq(l.color[0], l.color[1], l.color[2], 0.0f));
    }
    out.n_lights = (uint32_t)in.lights.size();
    out.n_materials = nm;
    out.normals.resize((size_t)nt * 3);
    parallel_slices(nt, [&](uint32_t first, uint32_t last) {
        for (uint32_t i = first; i < last; i++) {
            const float* n = &in.tri_nrm[(size_t)i * 9];
            q4* dst = &out.normals[(size_t)i * 3];
            dst[0] = mkq(n[0], n[1], n[2], 0.0f);
            dst[1] = mkq(n[3], n[4], n[5], 0.0f);
            dst[2] = mkq(n[6], n[7], n[8], 0.0f);
        }
    });
    out.n_prims = (uint32_t)total;
    out.prims.resize((size_t)total * 3);
    q4* const prims = out.prims.data();

    const uint32_t n_boxed = nt + ns;
    const bool tiny = world_kind == SRT_WORLD_SURFACELIST && n_boxed <= 8;
    std::vector<Box> pb;
    std::vector<uint32_t> members, strays;
    uint32_t n_tree = n_boxed;
    if (n_boxed > 0 && !tiny) {
        primitive_boxes(in, pb);
        if (world_kind == SRT_WORLD_SURFACELIST) {
            // The library's own tree only holds surfaces whose tests stay inside float range (see
            // scan_all_surfaces in srt_core.h); the others are scanned linearly, like the rects.
            members.reserve(n_boxed);
            for (uint32_t i = 0; i < n_boxed; i++) {
                bool tame = true;
                for (int k = 0; k < 3; k++)
                    tame = tame && std::fabs(pb[i].mn[k]) <= kTameRange && std::fabs(pb[i].mx[k]) <= kTameRange;   // false for NaN
                (tame ? members : strays).push_back(i);
                // Sphere.boundingBox of a negative radius has its corners the wrong way round; the reference's
                // own tree keeps it so, this one needs the box the sphere fills
                for (int k = 0; k < 3; k++)
                    if (pb[i].mn[k] > pb[i].mx[k]) std::swap(pb[i].mn[k], pb[i].mx[k]);
            }
        }
    }
    if (world_kind == SRT_WORLD_SURFACELIST && !members.empty()) {
        // Far-origin guard (trav_begin in srt_core.h).  Sphere.hit accepts a ray that passes a sphere of radius r at
        // distance m when m^2 <= r^2 + kappa |oc|^2 (the rounding error of its discriminant), so the box of a sphere is
        // widened by sqrt(r^2 + kappa L0^2) - |r| for every origin within L0 = twice the radius of the ball around the
        // tree's surfaces (every bounce and shadow ray, every camera inside the ball); queries from farther away widen the
        // boxes themselves, by far_slope per unit of distance: sqrt(kappa) for spheres (the bound for r -> 0), 2^-18 for
        // the triangle test, whose error grows linearly with the distance.
        const double kappa = 32.0 / 16777216.0;     // 32 * 2^-24: the bound is 20 * 2^-24
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        bool spheres_in_tree = false;
        for (uint32_t i : members) {
            for (int k = 0; k < 3; k++) { lo[k] = std::fmin(lo[k], (double)pb[i].mn[k]); hi[k] = std::fmax(hi[k], (double)pb[i].mx[k]); }
            spheres_in_tree = spheres_in_tree || i >= nt;
        }
        double r2 = 0.0;
        for (int k = 0; k < 3; k++) { out.far_center[k] = (float)(0.5 * (lo[k] + hi[k])); const double h = 0.5 * (hi[k] - lo[k]); r2 += h * h; }
        const double R = std::sqrt(r2) * 1.001 + 1e-30, L0 = 2.0 * R;
        out.far_radius = (float)R;
        out.far_L0 = (float)(L0 * 0.999);
        out.far_slope = spheres_in_tree ? (float)(std::sqrt(kappa) * 1.01) : 0x1p-18f;
        if (spheres_in_tree)
            for (uint32_t i : members) {
                if (i < nt) continue;
                const double r = std::fabs((double)in.sph[(size_t)(i - nt) * 4 + 3]);
                const float e = (float)((std::sqrt(r * r + kappa * L0 * L0) - r) * 1.001);
                for (int k = 0; k < 3; k++) { pb[i].mn[k] -= e; pb[i].mx[k] += e; }
            }
    }
    if (n_boxed > 0 && !tiny && (world_kind == SRT_WORLD_BVHNODE || !members.empty())) {
        if (world_kind == SRT_WORLD_BVHNODE) {
            ReferenceBuilder b(pb);
            b.run(insertion);
            for (uint32_t i = 0; i < b.leaf_order.size(); i++) emit_prim(in, b.leaf_order[i], i, prims + (size_t)i * 3);
            emit_nodes(b.nodes, out.nodes);
            out.leaf_order = b.leaf_order;
            out.n_nodes = (uint32_t)b.nodes.size();
            out.n_tree_nodes = out.n_nodes - 1u;
            out.bvh_depth = b.max_depth;
            if (b.nodes.size() <= (size_t)TINY_TREE_NODES) {
                emit_tiny(b.nodes, 0, out.tiny);
                out.n_tiny = (uint32_t)(out.tiny.size() / 2);
                out.tiny.push_back(mkq(0, 0, 0, 0));  // one spare node: the walk fetches one node ahead
                out.tiny.push_back(mkq(0, 0, 0, 0));
            }
        } else {
            SahBuilder b(pb);
            b.run(std::move(members));
            parallel_slices((uint32_t)b.leaf_order.size(), [&](uint32_t first, uint32_t last) {
                for (uint32_t i = first; i < last; i++) emit_prim(in, b.leaf_order[i], rank_of(b.leaf_order[i]), prims + (size_t)i * 3);
            });
            n_tree = (uint32_t)b.leaf_order.size();
            WideBuilder w(b.nodes);
            w.run();
            out.nodes.swap(w.out);
            out.n_wide = (uint32_t)(out.nodes.size() / 4);
            out.n_nodes = out.n_wide;
            out.n_tree_nodes = out.n_wide;
            out.bvh_depth = w.max_depth;
        }
        if (out.bvh_depth > (uint32_t)STACK_DEPTH) { err = "BVH deeper than the traversal stack"; return SRT_E_LIMIT; }
        out.has_bvh = 1;
        // scanned linearly after the tree: the surfaces kept out of it, then the rects
        out.list_first = n_tree;
        for (uint32_t i = 0; i < (uint32_t)strays.size(); i++) emit_prim(in, strays[i], rank_of(strays[i]), prims + (size_t)(n_tree + i) * 3);
        for (uint32_t i = 0; i < nr; i++) emit_prim(in, n_boxed + i, rank_of(n_boxed + i), prims + (size_t)(n_boxed + i) * 3);
        out.list_count = (uint32_t)strays.size() + nr;
    } else {
        // a world that is scanned as a list is laid out in the order of the list: the scan then meets the surfaces as
        // SurfaceList.hit does, NaN distances included
        for (uint32_t r = 0; r < (uint32_t)total; r++) emit_prim(in, insertion.empty() ? r : insertion[r], r, prims + (size_t)r * 3);
        out.list_first = 0;
        out.list_count = (uint32_t)total;
        out.has_bvh = 0;
    }
    // Scenes whose queries are stack traversals leave out the shadow rays that cannot matter (shadow_ray_matters in
    // srt_core.h), provided every material and light value is of ordinary size.
    {
        const float big = 0x1p40f;
        bool ordinary = true;
        for (const srt_material& m : in.materials)
            ordinary = ordinary && std::fabs(m.albedo[0]) <= big && std::fabs(m.albedo[1]) <= big && std::fabs(m.albedo[2]) <= big && std::fabs(m.metalness) <= big;   // false for NaN
        for (const srt_point_light& l : in.lights)
            ordinary = ordinary && std::fabs(l.position[0]) <= big && std::fabs(l.position[1]) <= big && std::fabs(l.position[2]) <= big &&
                       std::fabs(l.color[0]) <= big && std::fabs(l.color[1]) <= big && std::fabs(l.color[2]) <= big && std::fabs(l.brightness) <= big;
        out.inert_lights = (ordinary && out.has_bvh != 0u && out.n_tiny == 0u) ? 1u : 0u;
    }
    // never hand the device an empty allocation
    q4 zero = mkq(0, 0, 0, 0);
    if (out.nodes.empty()) out.nodes.assign(2, zero);
    if (out.prims.empty()) out.prims.assign(3, zero);
    if (out.normals.empty()) out.normals.assign(3, zero);
    if (out.materials.empty()) out.materials.assign(2, zero);
    if (out.lights.empty()) out.lights.assign(2, zero);
    if (out.tiny.empty()) out.tiny.assign(4, zero);
    return SRT_OK;
}

}  // namespace srt
