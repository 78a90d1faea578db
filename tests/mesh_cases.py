"""Mesh scenes made from the reference's two models (fixtures extracted from Content/models/*.dae) for the parity
tests of the library's own tree (SRT_WORLD_SURFACELIST: a binned-SAH tree with padded boxes).  The oracle
answers the same queries with SurfaceList.hit's linear scan, so any box that culls a real hit shows up as a
different pixel or ray count."""
import numpy as np

import helpers
import simple_raytracer_amd as srt
from simple_raytracer_amd import game1, scenes
from simple_raytracer_amd.raytracing import Material, Sphere


def teapot_with_spheres(width, height, n_spheres=40):
    """teapot.dae (6,320 triangles) plus random spheres of mixed materials in front of and around it."""
    world, lights, cam = scenes.teapot_scene(width, height, levels=0)
    rng = np.random.default_rng(5)
    mats = game1.materials()
    for i in range(n_spheres):
        c = rng.uniform(-3.5, 3.5, size=3) + np.array([0.2, -1.6, 1.0])
        world.surfaces.append(Sphere(c, float(rng.uniform(0.05, 0.6)), mats[i % 3] if i % 4 else
                                     Material(rng.uniform(0, 1, 3), float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), (0, 0, 0))))
    world.Init()
    return world, lights, cam


def sphere_mesh(width, height, levels=1):
    """sphere.dae subdivided once (3,840 triangles), framed like the 1M-triangle benchmark scene."""
    return scenes.sphere_mesh_scene(width, height, levels=levels)


CASES = [
    # name, scene factory, width, height, spp, depth, seed
    ("teapot-spheres-list", teapot_with_spheres, 64, 36, 1, 4, 1),
    ("teapot-spheres-list-depth8", teapot_with_spheres, 40, 24, 2, 8, 7),
    ("sphere-mesh-list", sphere_mesh, 64, 36, 2, 4, 3),
]


def run(renderer, orc, case, counted=False):
    name, make, w, h, spp, depth, seed = case
    world, lights, cam = make(w, h)
    renderer.set_scene(world, lights, cam, world_kind=srt.WORLD_SURFACELIST)
    img = renderer.render(w, h, samples=spp, max_depth=depth, seed=seed, counted=counted)
    st = renderer.stats()
    osc = helpers.oracle_scene_from_world(orc, world, lights)
    ref, ost = osc.render(helpers.oracle_camera_from(orc, cam), w, h, samples=spp, max_depth=depth, seed=seed,
                          world_kind=orc.WORLD_SURFACELIST)
    lsb, npx = helpers.max_lsb_diff(img, ref)
    return lsb, npx, st, ost
