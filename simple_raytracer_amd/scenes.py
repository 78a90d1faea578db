"""Synthetic scenes for the mesh benchmark configurations (the reference ships only two small meshes,
Content/models/teapot.dae with 6,320 triangles and sphere.dae with 960; the ~100k and ~1M triangle inputs are made
from them by midpoint subdivision)."""
import os

import numpy as np

from . import game1
from .raytracing import Camera, LightList, Material, PointLight, Sphere, SurfaceList, TriMesh

_FIXTURES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def load_fixture_mesh(name):
    """(pos9, nrm9) of the reference's `name`.dae as its loader would hand it to the tracer (ModelLoader.cs:39-64)."""
    d = np.load(os.path.join(_FIXTURES, name + "_dae.npz"))
    return d["pos9"], d["nrm9"]


def subdivide(pos9, nrm9, levels=1):
    """Midpoint subdivision: 4x the triangles per level; corner normals interpolated and re-normalised."""
    p = np.asarray(pos9, dtype=np.float32).reshape(-1, 3, 3)
    n = np.asarray(nrm9, dtype=np.float32).reshape(-1, 3, 3)
    for _ in range(levels):
        def mid(a, b):
            return ((a.astype(np.float64) + b.astype(np.float64)) * 0.5).astype(np.float32)

        def nmid(a, b):
            m = a.astype(np.float64) + b.astype(np.float64)
            l = np.linalg.norm(m, axis=-1, keepdims=True)
            l[l == 0] = 1
            return (m / l).astype(np.float32)
        pa, pb, pc = p[:, 0], p[:, 1], p[:, 2]
        na, nb, nc = n[:, 0], n[:, 1], n[:, 2]
        pab, pbc, pca = mid(pa, pb), mid(pb, pc), mid(pc, pa)
        nab, nbc, nca = nmid(na, nb), nmid(nb, nc), nmid(nc, na)
        p = np.concatenate([np.stack(t, axis=1) for t in ((pa, pab, pca), (pab, pb, pbc), (pca, pbc, pc), (pab, pbc, pca))])
        n = np.concatenate([np.stack(t, axis=1) for t in ((na, nab, nca), (nab, nb, nbc), (nca, nbc, nc), (nab, nbc, nca))])
    return np.ascontiguousarray(p.reshape(-1, 9)), np.ascontiguousarray(n.reshape(-1, 9))


def _default_lights():
    lights = LightList()
    lights.lights.append(PointLight((6.0, 2.0, 3.0), 5.0))       # Game1.cs:140-141
    lights.lights.append(PointLight((0.0, 0.0, -12.0), 2.5))
    lights.init()
    return lights


def teapot_scene(width, height, levels=2):
    """The 100k-triangle mesh configuration: teapot.dae subdivided `levels` times (2 -> 101,120 triangles), material mat6 (Game1.cs:109),
    the loader's fallback camera (fov 29, ModelLoader.cs:105) pulled back to frame the model, the two default lights."""
    p, n = subdivide(*load_fixture_mesh("teapot"), levels=levels)
    mat6 = game1.materials()[5]
    world = SurfaceList()
    world.surfaces.append(TriMesh(p, n, mat6, False, True))
    world.Init()
    camera = Camera((0.2, -1.6, 11.0), (0.2, -1.6, 0.0), 29, game1.aspect_ratio(width, height))
    return world, _default_lights(), camera


def sphere_mesh_scene(width, height, levels=5, n_spheres=0, seed=42):
    """The 1M-triangle configurations: sphere.dae subdivided `levels` times (5 -> 983,040 triangles), scaled by 10 and centred at
    (0, 0, -30); optionally `n_spheres` random metal/rough spheres around it (config 4: 10,000).  Camera at the origin."""
    p, n = subdivide(*load_fixture_mesh("sphere"), levels=levels)
    p = (p.reshape(-1, 3) * np.float32(10.0) + np.array([0, 0, -30], dtype=np.float32)).reshape(-1, 9)
    mat6 = game1.materials()[5]
    world = SurfaceList()
    world.surfaces.append(TriMesh(np.ascontiguousarray(p), n, mat6, False, True))
    if n_spheres:
        rng = np.random.default_rng(seed)
        centres = rng.uniform(-24, 24, size=(n_spheres, 3)) + np.array([0, 0, -30])
        radii = rng.integers(2, 50, size=n_spheres) / 100.0          # the pattern of Game1.cs:126-127, scaled down
        albedo, metal, smooth = rng.uniform(0, 1, size=(n_spheres, 3)), rng.uniform(0, 1, size=n_spheres), rng.uniform(0, 1, size=n_spheres)
        for i in range(n_spheres):
            world.surfaces.append(Sphere(centres[i], radii[i], Material(albedo[i], metal[i], smooth[i], (0, 0, 0))))
    world.Init()
    camera = Camera((0, 0, 0), (0, 0, -30), 50, game1.aspect_ratio(width, height))
    return world, _default_lights(), camera
