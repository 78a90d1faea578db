// srt_scene.h -- host-side scene flattening and BVH construction (no HIP calls).
#ifndef SRT_SCENE_H
#define SRT_SCENE_H

#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/srt.h"
#include "srt_core.h"

namespace srt {

// What the srt_set_* calls copied from the caller: the reference's world.surfaces /
// lights.lights / materials, by type.  Global surface index: triangles first, then spheres, then rects.
struct HostScene {
    std::vector<uint8_t> order_type;      // srt_set_surface_order: world.surfaces entry by entry (empty: triangles, spheres, rects)
    std::vector<uint32_t> order_index;
    std::vector<srt_material> materials;
    std::vector<float> tri_pos, tri_nrm;  // 9 floats per triangle
    std::vector<uint32_t> tri_mat;
    std::vector<uint8_t> tri_flags;       // bit0 flip, bit1 blend
    std::vector<float> sph;               // cx cy cz r
    std::vector<uint32_t> sph_mat;
    std::vector<float> rect;              // x1 x2 y1 y2 k
    std::vector<uint32_t> rect_mat;
    std::vector<srt_point_light> lights;
    uint32_t n_tris() const { return (uint32_t)tri_mat.size(); }
    uint32_t n_spheres() const { return (uint32_t)sph_mat.size(); }
    uint32_t n_rects() const { return (uint32_t)rect_mat.size(); }
};

// Device-ready arrays in the layout srt_core.h documents.
struct FlatScene {
    std::vector<q4> nodes, prims, normals, materials, lights, tiny;
    uint32_t n_lights = 0, list_first = 0, list_count = 0, has_bvh = 0, n_tiny = 0, n_wide = 0;
    uint32_t n_nodes = 0, bvh_depth = 0, n_prims = 0, n_materials = 0;
    uint32_t n_tree_nodes = 0;   // nodes of the tree itself: BVHNode objects of the reference tree (its array has one unused slot), or wide nodes
    // SRT_WORLD_BVHNODE: the surfaces (global index: all triangles, then all spheres, then all rects) in the order the reference tree's leaves
    // hold them, left to right -- the order BVHNode.hit visits them in.  Host-side only (tests compare it).
    std::vector<uint32_t> leaf_order;
    // far-origin guard of the library's own tree (SceneView::far_*); far_L0 = infinity when there is no such tree
    float far_center[3] = {0.0f, 0.0f, 0.0f};
    float far_radius = 0.0f, far_L0 = INFINITY, far_slope = 0.0f;
    uint32_t inert_lights = 0;   // SceneView::inert_lights

    // The view the kernels take, over device (or, for the CPU harness, host) copies of the arrays.
    SceneView view(const q4* d_nodes, const q4* d_prims, const q4* d_normals, const q4* d_materials, const q4* d_lights, const q4* d_tiny) const
    {
        SceneView v;
        v.nodes = as_scene_table(d_nodes); v.prims = as_scene_table(d_prims); v.normals = as_scene_table(d_normals); v.materials = d_materials;
        v.lights = as_scene_table(d_lights); v.tiny = as_scene_table(d_tiny);
        v.n_lights = n_lights; v.list_first = list_first; v.list_count = list_count; v.has_bvh = has_bvh; v.n_tiny = n_tiny;
        v.n_nodes = n_nodes; v.n_prims = n_prims; v.n_materials = n_materials; v.n_wide = n_wide;
        v.hit_prims = d_prims;
        v.far_center = mk3(far_center[0], far_center[1], far_center[2]);
        v.far_radius = far_radius; v.far_L0 = far_L0; v.far_slope = far_slope;
        v.inert_lights = inert_lights;
        return v;
    }
};

// Returns SRT_OK or a negative SRT_E_* code with `err` set.
int flatten_scene(const HostScene& in, int world_kind, FlatScene& out, std::string& err);

// Argument checking + copying behind the srt_set_* entry points.
int host_set_materials(HostScene& s, const srt_material* m, uint32_t n, std::string& err);
int host_set_spheres(HostScene& s, const float* cx, const float* cy, const float* cz, const float* r, const uint32_t* mat, uint32_t n, std::string& err);
int host_set_rects(HostScene& s, const float* x1, const float* x2, const float* y1, const float* y2, const float* k, const uint32_t* mat, uint32_t n, std::string& err);
int host_set_triangles(HostScene& s, const float* pos9, const float* nrm9, const uint32_t* mat, const uint8_t* flags, uint32_t n, std::string& err);
int host_set_lights(HostScene& s, const srt_point_light* l, uint32_t n, std::string& err);
int host_set_surface_order(HostScene& s, const uint8_t* type, const uint32_t* index, uint32_t n, std::string& err);

// Validates render parameters and derives the kernel arguments (tile grid, work item count).
int make_render_args(const srt_render_params* p, const Camera& cam, RenderArgs& ra, std::string& err);
// Pixels in the output buffer for these parameters, or a negative SRT_E_* code.
int64_t output_pixels(const srt_render_params* p);

}  // namespace srt
#endif
