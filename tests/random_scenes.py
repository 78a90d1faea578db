"""Randomly generated small scenes (seeded): spheres, optional rects, optional triangle meshes, mixed materials incl.
emissive / fully metallic / rough ones, 0-4 lights with and without colour, random cameras, both world kinds, all
render modes.  Every one is rendered by the device path (or the CPU harness) and by the oracle and compared."""
import numpy as np

import helpers
import simple_raytracer_amd as srt


def make_spec(seed):
    rng = np.random.default_rng(1000 + seed)
    s = helpers.SceneSpec()
    n_mat = int(rng.integers(1, 7))
    for i in range(n_mat):
        kind = rng.integers(0, 5)
        albedo = tuple(float(v) for v in rng.uniform(0, 1, 3))
        if kind == 0:
            s.materials.append(((0, 0, 0), 0.0, 0.0, tuple(float(v) for v in rng.uniform(0, 20, 3))))      # emissive, depth 1
        elif kind == 1:
            s.materials.append((albedo, 1.0, float(rng.uniform(0.3, 1.0)), (0, 0, 0)))                      # pure metal
        elif kind == 2:
            s.materials.append((albedo, 0.0, float(rng.uniform(0.0, 1.0)), (0, 0, 0)))                      # dielectric
        else:
            s.materials.append((albedo, float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), (0, 0, 0)))     # three children
    n_sph = int(rng.integers(0, 12))
    for i in range(n_sph):
        c = rng.uniform(-4, 4, 3) + np.array([0, 0, -8.0])
        s.spheres.append((tuple(float(v) for v in c), float(rng.uniform(0.2, 2.0)), int(rng.integers(0, n_mat))))
    if rng.random() < 0.5:
        p, n = helpers.uv_sphere(tuple(float(v) for v in rng.uniform(-2, 2, 3) + np.array([0, 0, -7.0])), float(rng.uniform(0.5, 2.0)),
                                 int(rng.integers(3, 9)), int(rng.integers(4, 12)))
        s.meshes.append((p, n, int(rng.integers(0, n_mat)), bool(rng.random() < 0.2), bool(rng.random() < 0.7)))
    if rng.random() < 0.3:
        gp, gn = helpers.grid_plane(float(rng.uniform(1.5, 3.0)), -6, 6, -14, -1, int(rng.integers(1, 5)), int(rng.integers(1, 5)), up=False)
        s.meshes.append((gp, gn, int(rng.integers(0, n_mat)), False, bool(rng.random() < 0.5)))
    has_rect = rng.random() < 0.3
    if has_rect:
        x1, y1 = rng.uniform(-4, 0, 2)
        s.rects.append((float(x1), float(x1 + rng.uniform(1, 5)), float(y1), float(y1 + rng.uniform(1, 5)), float(rng.uniform(6, 14)),
                        int(rng.integers(0, n_mat))))
    for i in range(int(rng.integers(0, 5))):
        color = None if rng.random() < 0.5 else tuple(float(v) for v in rng.uniform(0, 2.5, 3))
        s.lights.append((tuple(float(v) for v in rng.uniform(-8, 8, 3)), float(rng.uniform(0.5, 8.0)), color))
    eye = rng.uniform(-1, 1, 3)
    s.camera = (tuple(float(v) for v in eye), tuple(float(v) for v in rng.uniform(-1, 1, 3) + np.array([0, 0, -8.0])), float(rng.uniform(35, 80)))
    empty = n_sph == 0 and not s.meshes
    kind = srt.WORLD_SURFACELIST if (has_rect or empty or rng.random() < 0.5) else srt.WORLD_BVHNODE
    mode = [srt.MODE_COLOR] * 5 + [srt.MODE_NORMAL, srt.MODE_DEPTH, srt.MODE_LIGHTS, srt.MODE_DIFFUSE, srt.MODE_REFLECTION]
    return s, dict(width=int(rng.integers(3, 9)) * 8 + int(rng.integers(0, 8)), height=int(rng.integers(20, 40)),
                   samples=int(rng.integers(1, 4)), max_depth=int(rng.integers(1, 9)), seed=int(rng.integers(0, 2**31)),
                   mode=int(mode[int(rng.integers(0, len(mode)))]), color_clamp=float(rng.choice([1.0, 1.0, 0.5, 4.0]))), kind


def run(renderer, orc, seed):
    spec, kw, kind = make_spec(seed)
    w, h = kw.pop("width"), kw.pop("height")
    world, lights, cam = spec.host(w, h)
    renderer.set_scene(world, lights, cam, world_kind=kind)
    img = renderer.render(w, h, **kw)
    st = renderer.stats()
    ref, ost = spec.oracle_scene().render(spec.oracle_camera(w, h), w, h, world_kind=kind, **kw)
    lsb, npx = helpers.max_lsb_diff(img, ref)
    same_counts = all(st[k] == ost[k] for k in ("rays_path", "rays_shadow", "hit_nodes")) if kw["mode"] == srt.MODE_COLOR else True
    return lsb, npx, same_counts, (w, h, kw, kind)
