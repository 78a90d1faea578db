"""Degenerate inputs through the library, against the CPU oracle (a checking tool: it imports oracle/).

    python tools/degenerate_fuzz.py [cases] [--first SEED] [--emu]

Two families of random small scenes, each rendered for both world kinds:
  geometry   helpers.nasty_spec: zero-length edges, collinear / duplicated / coplanar triangles, lattice vertices,
             spheres of radius 0 or of negative radius, concentric spheres, a rect through the camera plane;
  shading    the Game1 scene or a small mesh scene with materials outside [0, 1] or NaN, lights of zero, negative
             or 1e30 brightness, at the camera, inside a sphere or at infinity, no lights, sixteen lights, and
             cameras with a field of view of 0, 1, 179 or 200 degrees, looking along the up vector or nowhere.
  far        helpers.far_camera_spec-like scenes: random spheres (radius 0.01 .. 2) and optionally a triangle mesh, seen
             through a narrow lens from 10 .. 10^7 scene sizes away, where the float sphere test reports hits for rays that
             miss by a wide margin and the library's tree must not cull them.
A case passes when the image is within 1 LSB per channel of the oracle's and the ray counts are equal.
--emu runs the kernel source on the CPU (tests/emu) instead of the GPU library.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np  # noqa: E402

import simple_raytracer_amd as srt  # noqa: E402
import helpers  # noqa: E402
import parity_cases  # noqa: E402
import orc  # noqa: E402


def f32(v):
    return float(np.float32(v))


def shading_spec(seed):
    rng = np.random.default_rng(seed)
    s = helpers.default_spec() if rng.integers(0, 2) else helpers.mesh_spec(n=6)
    mats = []
    for a, m, sm, e in s.materials:
        k = rng.integers(0, 10)
        if k == 0:
            m = f32(rng.choice([-0.5, 1.5, 0.0, 1.0]))
        elif k == 1:
            sm = f32(rng.choice([0.0, 1.0, -0.2, 1.3]))
        elif k == 2:
            a = tuple(f32(v) for v in rng.choice([0.0, 2.0, -1.0, 1.0], 3))
        elif k == 3:
            e = tuple(f32(v) for v in rng.choice([0.0, 50.0, -1.0, 1e20], 3))
        elif k == 4:
            a = (float("nan"), 0.5, 0.5)
        elif k == 5:
            sm = float("nan")
        mats.append((a, m, sm, e))
    s.materials = mats
    lights = []
    for p, b, c in s.lights:
        k = rng.integers(0, 8)
        if k == 0:
            b = 0.0
        elif k == 1:
            b = -3.0
        elif k == 2:
            p = s.camera[0]
        elif k == 3 and s.spheres:
            p = s.spheres[0][0]
        elif k == 4:
            b = 1e30
        elif k == 5:
            p = (float("inf"), 0.0, 0.0)
        lights.append((p, b, c))
    if rng.integers(0, 6) == 0:
        lights = []
    if rng.integers(0, 6) == 0:
        lights = lights * 4
    s.lights = lights
    o, l, fov = s.camera
    k = rng.integers(0, 10)
    if k == 0:
        fov = 1.0
    elif k == 1:
        fov = 179.0
    elif k == 2:
        l = (o[0], o[1] - 5.0, o[2])
    elif k == 3:
        l = o
    elif k == 4:
        fov = 0.0
    elif k == 5:
        o = (0.0, 0.0, -5.0)
    elif k == 6:
        fov = 200.0
    s.camera = (o, l, fov)
    return s


def far_spec(seed):
    rng = np.random.default_rng(7000 + seed)
    s = helpers.SceneSpec()
    s.materials = [((0.8, 0.5, 0.5), 0.0, 0.5, (0, 0, 0)), ((0.6, 0.7, 0.9), 0.5, 0.8, (0, 0, 0))]
    n = int(rng.integers(9, 40))
    spread = float(rng.choice([1.0, 4.0, 20.0]))
    s.spheres = [((f32(x), f32(y), f32(z)), f32(rad), int(rng.integers(0, 2)))
                 for (x, y, z), rad in zip(rng.uniform(-spread, spread, (n, 3)), 10 ** rng.uniform(-2, 0.3, n))]
    if rng.integers(0, 2):
        p, nr = helpers.uv_sphere(tuple(rng.uniform(-spread, spread, 3)), float(rng.uniform(0.3, 2.0)), 6, 12)
        s.meshes.append((p, nr, 1, False, True))
    s.lights = [((0.0, -30.0 * spread, 0.0), 5.0, None)]
    dist = float(10 ** rng.uniform(1, 7)) * spread
    axis = int(rng.integers(0, 3))
    o = [0.0, 0.0, 0.0]
    o[axis] = dist * float(rng.choice([-1.0, 1.0]))
    if rng.integers(0, 2):
        o[(axis + 1) % 3] = 0.37 * dist
    s.camera = (tuple(o), (0, 0, 0), float(np.degrees(2 * np.arctan(1.5 * spread / np.linalg.norm(o)))))
    return s


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100
    np.seterr(all="ignore")
    lib = helpers.build_emu() if "--emu" in sys.argv else None
    r = srt.Renderer(device=0, library=lib, strict=lib is None)
    modes = [srt.MODE_COLOR] * 4 + [srt.MODE_NORMAL, srt.MODE_DEPTH, srt.MODE_LIGHTS, srt.MODE_DIFFUSE, srt.MODE_REFLECTION]
    bad = done = 0
    first = int(sys.argv[sys.argv.index("--first") + 1]) if "--first" in sys.argv else 0
    for seed in range(first, first + n):
        rng = np.random.default_rng(1000 + seed)
        for family, spec in (("geometry", helpers.nasty_spec(seed)), ("shading", shading_spec(seed)), ("far", far_spec(seed))):
            depth, spp, mode = int(rng.choice([0, 1, 3, 6, 9])), int(rng.choice([1, 2, 3])), int(rng.choice(modes))
            if family == "far":
                depth, mode = min(depth, 3), srt.MODE_COLOR
            if family == "shading":
                depth = min(depth, 6)   # a camera that looks nowhere sends NaN rays: three children per level, every surface tested
            for kind in (srt.WORLD_SURFACELIST, srt.WORLD_BVHNODE):
                if kind == srt.WORLD_BVHNODE and spec.rects:
                    continue
                case = ("%s-%d" % (family, seed), lambda: spec, 40, 24, spp, depth, mode, kind, seed)
                lsb, npx, st, ost = parity_cases.run_case(r, orc, case)
                done += 1
                counts = mode != srt.MODE_COLOR or (st["rays_path"] == ost["rays_path"] and st["hit_nodes"] == ost["hit_nodes"])
                if lsb > 1 or not counts:
                    bad += 1
                    print(family, seed, "kind", kind, "mode", mode, "depth", depth, "lsb", lsb, "pixels", npx,
                          "rays", st["rays_path"], ost["rays_path"], flush=True)
        if seed % 50 == 49:
            print("... seed", seed, flush=True)
    print("degenerate fuzz:", done, "renders,", bad, "outside 1 LSB or with different ray counts")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
