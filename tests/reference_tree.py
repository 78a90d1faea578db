"""The reference's own BVH (new BVHNode(world), RaytracingClasses.cs:346-426) as the library builds and walks it for
SRT_WORLD_BVHNODE worlds, pinned against the oracle's pointer tree: node count, depth, the order of the surfaces in the
leaves, and the number of box / sphere / triangle tests of a render.

The oracle counts every test the reference performs; the library reuses two answers the reference computes again (the
per-light re-intersection of the node's own ray, RaytracingClasses.cs:211, and the second test of a single-surface leaf,
:362-365), and the oracle counts those repeats separately, so tests - repeats must be the library's counts exactly.  A
tree of another shape that happened to give the same picture would fail here.

Scenes whose tree is walked with a stack also leave out the shadow rays that cannot change their light's contribution (a
light behind the surface); the library reports how many (shadow_unasked of a counted render), the oracle classifies
the same rays by its own restatement of the rule and counts their tests apart, so the number of such rays and
tests - repeats - inert must match exactly as well."""
import numpy as np

import helpers
import simple_raytracer_amd as srt

SCENES = {
    "game1": (lambda: helpers.default_spec(), 96, 54, 2, 8),
    "glossy": (lambda: helpers.glossy_spec(), 64, 36, 1, 8),
    "mesh+spheres": (lambda: helpers.mesh_spec(), 64, 36, 1, 4),
    "mesh-only": (lambda: helpers.mesh_spec(with_spheres=False), 48, 27, 1, 3),
    "teapot.dae": (lambda: helpers.teapot_spec(), 48, 27, 1, 2),
    # world.surfaces in an order of the caller's (srt_set_surface_order): another list, another tree
    "spheres-first": (lambda: helpers.reordered_spec("spheres-first"), 48, 27, 1, 3),
    "interleaved": (lambda: helpers.reordered_spec("interleaved"), 48, 27, 1, 3),
    "game1-reversed": (lambda: helpers.reordered_spec("game1-reversed"), 64, 36, 2, 6),
}

STACK_WALKED = ("mesh+spheres", "mesh-only", "teapot.dae", "spheres-first", "interleaved")   # trees too large for the stackless walk


def run(renderer, orc, name):
    """-> (library stats, oracle stats, library (nodes, depth), oracle (nodes, depth), max LSB)."""
    make, w, h, spp, depth = SCENES[name]
    spec = make()
    world, lights, cam = spec.host(w, h)
    renderer.set_scene(world, lights, cam, world_kind=srt.WORLD_BVHNODE)
    img = renderer.render(w, h, samples=spp, max_depth=depth, seed=5, counted=True)
    st = renderer.stats()
    osc = spec.oracle_scene()
    ref, ost = osc.render(spec.oracle_camera(w, h), w, h, samples=spp, max_depth=depth, seed=5, world_kind=orc.WORLD_BVHNODE)
    lsb, _ = helpers.max_lsb_diff(img, ref)
    return st, ost, (st["bvh_nodes"], st["bvh_depth"]), (osc.bvh_nodes(), osc.bvh_depth()), lsb, osc


def check_counts(st, ost):
    for k in ("rays_path", "rays_shadow", "hit_nodes"):
        assert st[k] == ost[k], k
    assert ost["rays_redundant"] == ost["rays_shadow"]              # one re-intersection per light of every hit node
    unasked = st["shadow_unasked"]     # shadow rays the library answered without a traversal
    if unasked:
        assert unasked == ost["shadow_inert"], (unasked, ost["shadow_inert"])
    for kind in ("box", "sphere", "tri"):
        expect = ost[kind + "_tests"] - ost[kind + "_repeats"] - (ost[kind + "_inert"] if unasked else 0)
        assert st[kind + "_tests"] == expect, (kind, st[kind + "_tests"], ost[kind + "_tests"], ost[kind + "_repeats"], ost[kind + "_inert"], unasked)
    assert st["box_tests"] > 0
    return unasked


def insertion_ranks(name, leaf_order, oracle_side=False):
    """Leaf order as positions in world.surfaces.  The two sides number the surfaces by type (all triangles, then all spheres,
    then all rects), but the host mirror fills its per-type arrays in the order of world.surfaces while the oracle is handed
    the scene description's own arrays, so the same surface can carry different numbers on the two sides; its position in
    world.surfaces is the same."""
    spec = SCENES[name][0]()
    if oracle_side:
        if spec.order is None:
            return np.asarray(leaf_order)
        first = np.concatenate([[0], np.cumsum([len(m[0]) for m in spec.meshes])]).astype(np.int64)
        n_tri = int(first[-1])
        surfaces = []
        for kind, i in spec.order:
            surfaces.extend(range(first[i], first[i + 1]) if kind == "mesh" else [n_tri + i if kind == "sphere" else n_tri + len(spec.spheres) + i])
    else:
        flat = spec.host(8, 8)[0].flatten()
        first = np.array([0, len(flat["tri_pos"]), len(flat["tri_pos"]) + len(flat["spheres"])], dtype=np.int64)
        surfaces = first[flat["order_type"]] + flat["order_index"]
    rank = np.empty(len(surfaces), dtype=np.int64)
    rank[np.asarray(surfaces, dtype=np.int64)] = np.arange(len(surfaces))
    return rank[np.asarray(leaf_order, dtype=np.int64)]
