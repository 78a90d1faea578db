"""Shared test helpers: one scene description, turned into oracle inputs and into host-mirror objects."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import orc  # noqa: E402  (test infrastructure)
import simple_raytracer_amd as srt  # noqa: E402
from simple_raytracer_amd import game1  # noqa: E402

EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libsrt_emu.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def build_emu():
    """Compiles the CPU harness of the kernel source (tests/emu)."""
    subprocess.check_call(["make", "-C", EMU_DIR, "-s"])
    return EMU_LIB


class SceneSpec:
    """Materials / surfaces / lights / camera in plain numbers."""

    def __init__(self):
        self.materials = []   # (albedo3, metal, smooth, emission3)
        self.spheres = []     # (center3, radius, material)
        self.rects = []       # (x1, x2, y1, y2, k, material)
        self.meshes = []      # (pos9 (n,9), nrm9 (n,9), material, flip, blend)
        self.lights = []      # (position3, brightness, color3 or None)
        self.camera = ((0, 0, 0), (0, 0, -1), 60.0)
        # world.surfaces entry by entry: ("mesh", i) / ("sphere", j) / ("rect", k); None = meshes, spheres, rects
        self.order = None

    # -- oracle side ---------------------------------------------------------------------------
    def oracle_scene(self):
        mats = [orc.material(a, m, s, e) for a, m, s, e in self.materials]
        spheres = [orc.Sphere(orc.f3(c), float(r), int(mi)) for c, r, mi in self.spheres]
        rects = [orc.Rect(*[float(v) for v in r[:5]], int(r[5])) for r in self.rects]
        tris = None
        if self.meshes:
            tris = np.concatenate([orc.tris_from_arrays(p, n, mi, fl, bl) for p, n, mi, fl, bl in self.meshes])
        lights = [orc.point_light(p, b, c) for p, b, c in self.lights]
        order = None
        if self.order is not None:
            first = np.concatenate([[0], np.cumsum([len(m[0]) for m in self.meshes])]).astype(np.int64)
            types, indices = [], []
            for kind, i in self.order:
                if kind == "mesh":
                    types.append(np.zeros(first[i + 1] - first[i], dtype=np.uint8))
                    indices.append(np.arange(first[i], first[i + 1], dtype=np.uint32))
                else:
                    types.append(np.array([1 if kind == "sphere" else 2], dtype=np.uint8))
                    indices.append(np.array([i], dtype=np.uint32))
            order = (np.concatenate(types), np.concatenate(indices))
        return orc.Scene(mats, spheres=spheres, rects=rects, tris=tris, lights=lights, order=order)

    def oracle_camera(self, width, height):
        o, l, fov = self.camera
        return orc.camera(o, l, fov, float(np.float32(width) / np.float32(height)))

    # -- host mirror side ------------------------------------------------------------------------
    def host(self, width, height):
        mats = [srt.Material(a, m, s, e) for a, m, s, e in self.materials]
        world = srt.SurfaceList()
        order = self.order
        if order is None:
            order = [("mesh", i) for i in range(len(self.meshes))] + [("sphere", i) for i in range(len(self.spheres))] + \
                    [("rect", i) for i in range(len(self.rects))]
        for kind, i in order:
            if kind == "mesh":
                p, n, mi, fl, bl = self.meshes[i]
                world.surfaces.append(srt.TriMesh(p, n, mats[mi], fl, bl))
            elif kind == "sphere":
                c, r, mi = self.spheres[i]
                world.surfaces.append(srt.Sphere(c, r, mats[mi]))
            else:
                r = self.rects[i]
                world.surfaces.append(srt.RectAxis(r[0], r[1], r[2], r[3], r[4], mats[r[5]]))
        world.Init()
        lights = srt.LightList()
        for p, b, c in self.lights:
            lights.lights.append(srt.PointLight(p, b, c))
        lights.init()
        o, l, fov = self.camera
        cam = srt.Camera(o, l, fov, game1.aspect_ratio(width, height))
        return world, lights, cam


def default_spec(with_rect=False):
    """Game1.Initialize (Game1.cs:96-144)."""
    s = SceneSpec()
    z = (0, 0, 0)
    s.materials = [((0.78, 0.79, 0.33), 0.0, 0.995, z), ((1.0, 0.65, 0.62), 1.0, 0.65, z), ((0.5, 0.425, 0.15), 0.0, 0.45, z),
                   (z, 0, 0.0, (50.0, 0.5, 0.5)), (z, 0, 0.0, (0.5, 0.5, 50.0)), ((1.0, 0.5, 0.5), 0.0, 0.85, z)]
    s.spheres = [((2.05, 0.9, -7.25), 3.5, 0), ((-4.5, 2.0, -8.0), 2.3, 1), ((0, 45, -20.0), 42.5, 2), ((-20, 0, 0.0), 2, 3),
                 ((-15, 0, 6.0), 6, 4)]
    if with_rect:
        s.rects = [(-6, -3, -4, 4, -20, 3)]
    s.lights = [((6.0, 2.0, 3.0), 5.0, None), ((0.0, 0.0, -12.0), 2.5, None)]
    s.camera = ((0, 0, 0), (-0.03, 0.875, -7.6), 63)
    return s


def glossy_spec():
    """The default scene with smooth, half-metallic materials: every hit keeps the full recursion depth and starts all
    three children (diffuse, metal, reflect), so a sample's ray tree is as deep and as wide as it can get."""
    s = default_spec()
    z = (0, 0, 0)
    s.materials = [((0.78, 0.79, 0.33), 0.5, 0.995, z), ((1.0, 0.65, 0.62), 0.5, 0.65, z), ((0.5, 0.425, 0.15), 0.25, 0.7, z),
                   (z, 0, 0.0, (50.0, 0.5, 0.5)), (z, 0, 0.0, (0.5, 0.5, 50.0)), ((1.0, 0.5, 0.5), 0.75, 0.85, z)]
    return s


def many_materials_spec():
    """The glossy scene with 80 unused materials on top: more than the kernels for small scenes keep in LDS, so the
    general stackless kernel renders it."""
    s = glossy_spec()
    s.materials = s.materials + [((0.1 * (i % 7), 0.2, 0.3), 0.5, 0.5, (0, 0, 0)) for i in range(80)]
    return s


# -- procedural meshes ---------------------------------------------------------------------------
def uv_sphere(center, radius, n_lat, n_lon):
    """Triangle soup of a latitude/longitude sphere with outward vertex normals -> (pos9, nrm9)."""
    c = np.asarray(center, dtype=np.float64)
    th = np.linspace(0, np.pi, n_lat + 1)
    ph = np.linspace(0, 2 * np.pi, n_lon + 1)
    T, P = np.meshgrid(th, ph, indexing="ij")
    N = np.stack([np.sin(T) * np.cos(P), np.cos(T), np.sin(T) * np.sin(P)], axis=-1)
    V = c + radius * N
    pos, nrm = [], []
    for i in range(n_lat):
        for j in range(n_lon):
            a, b, cc, d = (i, j), (i + 1, j), (i + 1, j + 1), (i, j + 1)
            for tri in ((a, d, cc), (a, cc, b)):  # counter-clockwise seen from outside
                p = [V[t] for t in tri]
                if np.linalg.norm(np.cross(p[1] - p[0], p[2] - p[0])) < 1e-12:
                    continue
                pos.append(np.concatenate(p))
                nrm.append(np.concatenate([N[t] for t in tri]))
    return np.array(pos, dtype=np.float32), np.array(nrm, dtype=np.float32)


def grid_plane(y, x0, x1, z0, z1, nx, nz, up=True):
    """Axis-aligned flat grid (zero-thickness bounding boxes) at height y -> (pos9, nrm9)."""
    xs = np.linspace(x0, x1, nx + 1)
    zs = np.linspace(z0, z1, nz + 1)
    pos, nrm = [], []
    n = np.array([0, 1 if up else -1, 0], dtype=np.float64)
    for i in range(nx):
        for j in range(nz):
            a = np.array([xs[i], y, zs[j]]); b = np.array([xs[i + 1], y, zs[j]])
            c = np.array([xs[i + 1], y, zs[j + 1]]); d = np.array([xs[i], y, zs[j + 1]])
            tris = ((a, c, b), (a, d, c)) if up else ((a, b, c), (a, c, d))
            for tri in tris:
                pos.append(np.concatenate(tri))
                nrm.append(np.concatenate([n, n, n]))
    return np.array(pos, dtype=np.float32), np.array(nrm, dtype=np.float32)


def subdivide(pos9, nrm9, levels=1):
    """Midpoint subdivision (x4 triangles per level); normals interpolated and re-normalised."""
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
    return p.reshape(-1, 9), n.reshape(-1, 9)


def mesh_spec(kind="sphere", n=16, with_spheres=True, with_rect=False):
    """A small mixed scene: a triangle mesh + a ground grid + a few spheres, two lights."""
    s = SceneSpec()
    z = (0, 0, 0)
    s.materials = [((1.0, 0.5, 0.5), 0.0, 0.85, z), ((0.6, 0.7, 0.9), 0.4, 0.7, z), ((0.5, 0.45, 0.3), 0.0, 0.3, z),
                   ((0.9, 0.9, 0.9), 1.0, 0.9, z), (z, 0.0, 0.0, (4.0, 3.0, 1.0))]
    if kind == "sphere":
        p, nr = uv_sphere((0.2, 0.1, -6.0), 1.6, n, 2 * n)
    else:
        raise ValueError(kind)
    s.meshes.append((p, nr, 0, False, True))
    gp, gn = grid_plane(1.9, -6, 6, -12, -1, 6, 6, up=False)  # the image is y-down: the "floor" faces -y
    s.meshes.append((gp, gn, 2, False, False))
    if with_spheres:
        s.spheres = [((-2.6, 0.6, -5.5), 1.0, 1), ((2.7, 0.9, -5.0), 0.8, 3), ((0.0, -3.0, -9.0), 0.7, 4)]
    if with_rect:
        s.rects = [(-3, 3, -3, 1, 9.0, 1)]  # RectAxis evaluates t = -k - o.z/d.z: with o.z = 0 the plane sits at z = -9 for d.z = -1 rays
    s.lights = [((4.0, -3.0, 1.0), 5.0, None), ((-3.0, -2.0, -2.0), 2.0, (1.2, 0.4, 0.4))]
    s.camera = ((0, 0, 0), (0.0, 0.2, -6.0), 55)
    return s


def reordered_spec(kind):
    """Worlds whose surfaces were not added type by type: world.surfaces is one ordered list (RaytracingClasses.cs:340), the
    BVHNode constructor splits it by position (:387-425) and SurfaceList.hit lets the later surface win a tie (:298-316).
    "spheres-first": mesh_spec() with its spheres added before the meshes, as Game1.cs:117-122 followed by
    ModelLoader.CreateScene would; "interleaved": a sphere, the ground grid, a sphere, the mesh, a sphere;
    "game1-reversed": the Game1 spheres in reverse order; "rect-between": the Game1 scene with its rect added second."""
    if kind == "game1-reversed":
        s = default_spec()
        s.order = [("sphere", i) for i in reversed(range(len(s.spheres)))]
    elif kind == "rect-between":
        s = default_spec(with_rect=True)
        s.order = [("sphere", 0), ("rect", 0)] + [("sphere", i) for i in range(1, len(s.spheres))]
    else:
        s = mesh_spec(n=8)
        if kind == "spheres-first":
            s.order = [("sphere", 0), ("sphere", 1), ("sphere", 2), ("mesh", 0), ("mesh", 1)]
        elif kind == "interleaved":
            s.order = [("sphere", 2), ("mesh", 1), ("sphere", 0), ("mesh", 0), ("sphere", 1)]
        else:
            raise ValueError(kind)
    return s


def inert_light_spec(kind):
    """mesh_spec() bent towards the edges of the rule for shadow rays that cannot matter (a light behind the surface):
    "plain" keeps it as it is; "mirror" makes every material perfectly smooth (roughness 0, where the rule does not
    apply); "rough" uses the smoothness values at and next to the ends of [0, 1], to which the Material constructor clamps;
    "huge-brightness", "huge-light" and "nan-metal" carry values for which the scene as a whole must keep all its shadow
    rays (albedo and metalness are clamped to [0, 1] by the constructor, RaytracingClasses.cs:221-262, but a NaN passes)."""
    s = mesh_spec()
    z = (0, 0, 0)
    if kind == "mirror":
        s.materials = [(a, m, 1.0, e) for a, m, _, e in s.materials]
    elif kind == "rough":
        s.materials = [((1.0, 0.5, 0.5), 0.0, -0.5, z), ((0.6, 0.7, 0.9), 0.4, 2.3, z), ((0.5, 0.45, 0.3), 0.0, 1.0 - 2.0 ** -11, z),
                       ((0.9, 0.9, 0.9), 1.0, 1.0 + 2.0 ** -9, z), (z, 0.0, 0.0, (4.0, 3.0, 1.0))]
    elif kind == "huge-brightness":
        s.lights[0] = ((4.0, -3.0, 1.0), 3.0e12, None)
    elif kind == "huge-light":
        s.lights[1] = ((-3.0e13, -2.0, -2.0), 2.0, (1.2, 0.4, 0.4))
    elif kind == "nan-metal":
        s.materials[1] = ((0.6, 0.7, 0.9), float("nan"), 0.7, z)
    elif kind != "plain":
        raise ValueError(kind)
    return s


def stray_spec(kind):
    """300 random triangles and a sphere, plus surfaces whose coordinates leave the range in which a surface test
    stays finite: NaN or infinite vertices, or a triangle 1e20 / 3e38 across.  The reference's tests then report
    hits at t = NaN, which its list scan accepts whatever came before; a SurfaceList world has to reproduce that
    scan, not the geometry."""
    rng = np.random.default_rng(1)
    base = (rng.normal(0, 3, (300, 9)) + np.array([0, 0, -8] * 3)).astype(np.float32)
    sign = np.array([1, 1, 1, -1, 1, 1, 1, -1, 1], np.float32)
    extra = {
        "nan": np.full((3, 9), np.nan, np.float32),
        "one-nan": np.array([[np.nan, 0, -5, 1, 0, -5, 0, 1, -5]], np.float32),
        "inf": np.full((2, 9), np.inf, np.float32),
        "3e38": np.full((1, 9), 3e38, np.float32) * sign,
        "1e20": np.full((1, 9), 1e20, np.float32) * sign,
        "1e20-first": np.full((1, 9), 1e20, np.float32) * sign,
    }[kind]
    pos = np.concatenate([extra, base] if kind.endswith("first") else [base[:150], extra, base[150:]])
    s = SceneSpec()
    s.materials = [((0.8, 0.5, 0.5), 0.0, 0.5, (0, 0, 0)), ((0.6, 0.7, 0.9), 0.5, 0.8, (0, 0, 0))]
    s.meshes.append((pos, np.tile(np.array([[0, 0, 1] * 3], np.float32), (len(pos), 1)), 0, False, True))
    s.spheres = [((0.0, 0.0, -5.0), 1.0, 1)]
    s.lights = [((4.0, -3.0, 1.0), 5.0, None)]
    s.camera = ((0, 0, 0), (0.0, 0.0, -6.0), 55)
    return s


def far_camera_spec(distance, with_mesh=False):
    """Thirteen spheres (radius 0.02 .. 1) around the origin, optionally with a triangle-mesh sphere among them, seen
    through a narrow lens from `distance` away.  Far from the scene the float surface tests lose their low bits:
    Sphere.hit's discriminant half_b^2 - a (|oc|^2 - r^2) cancels catastrophically, and at a distance of 10,000 a ray
    that passes a unit sphere 1.5 units away still "hits" it.  The reference's list scan reports those hits, so a tree
    that answers SurfaceList.hit must not cull them."""
    s = SceneSpec()
    s.materials = [((0.8, 0.5, 0.5), 0.0, 0.5, (0, 0, 0)), ((0.6, 0.7, 0.9), 0.5, 0.8, (0, 0, 0))]
    rng = np.random.default_rng(3)
    s.spheres = [((0.0, 0.0, 0.0), 1.0, 0)] + [((float(x), float(y), float(z)), float(r), 1)
                                               for (x, y, z), r in zip(rng.uniform(-4, 4, (12, 3)), rng.uniform(0.02, 0.6, 12))]
    if with_mesh:
        p, n = uv_sphere((0.5, 1.0, 0.5), 1.5, 8, 16)
        s.meshes.append((p, n, 1, False, True))
    s.lights = [((0.0, -30.0, 0.0), 5.0, None)]
    s.camera = ((distance, 0, 0), (0, 0, 0), float(np.degrees(2 * np.arctan(6.0 / distance))))
    return s


def nasty_spec(seed):
    """Random small scenes made of the inputs a model file can hold and a tree builder trips over: zero-length
    edges, collinear and duplicated triangles, vertices on a lattice (equal hit distances), coplanar triangles,
    spheres of radius 0, of negative radius (Sphere.boundingBox then has its corners the wrong way round), spheres
    the camera is inside of, concentric spheres, a rect through the camera plane."""
    rng = np.random.default_rng(seed)
    s = SceneSpec()
    s.materials = [((0.8, 0.5, 0.5), 0.0, 0.5, (0, 0, 0)), ((0.6, 0.7, 0.9), 0.7, 0.9, (0, 0, 0)), ((0.9, 0.9, 0.2), 1.0, 0.3, (0.5, 0.5, 0))]
    nt = int(rng.integers(0, 60))
    pos = (rng.normal(0, 2.5, (nt, 9)) + np.array([0, 0, -7] * 3)).astype(np.float32)
    for i in range(nt):
        k = rng.integers(0, 12)
        if k == 0:
            pos[i, 3:6] = pos[i, 0:3]
        elif k == 1:
            pos[i, 6:9] = pos[i, 0:3] + 2 * (pos[i, 3:6] - pos[i, 0:3])
        elif k == 2 and i > 0:
            pos[i] = pos[i - 1]
        elif k == 3:
            pos[i] = np.round(pos[i])
        elif k == 4:
            pos[i, 2::3] = -6.0
    if nt:
        nrm = np.tile(np.array([[0, 0, 1] * 3], np.float32), (nt, 1))
        s.meshes.append((pos, nrm, int(rng.integers(0, 3)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))))
    s.spheres = []
    for i in range(int(rng.integers(0, 14))):
        c = tuple(float(v) for v in (rng.normal(0, 2.5, 3) + np.array([0, 0, -7])).astype(np.float32))
        rad = float(np.float32(rng.choice([0.0, -0.8, 0.5, 1.0, 1e-3, 30.0, 1.0, 0.7])))
        if i > 0 and rng.integers(0, 5) == 0:
            c = s.spheres[-1][0]
        s.spheres.append((c, rad, int(rng.integers(0, 3))))
    if rng.integers(0, 3) == 0:
        s.rects = [(-3, 3, -3, 1, float(rng.choice([9.0, 6.0, 0.0])), 1)]
    s.lights = [((4.0, -3.0, 1.0), 5.0, None), ((0.0, 0.0, -7.0), 2.0, (1.0, 0.5, 0.5))]
    s.camera = ((0, 0, 0), (0.0, 0.0, -6.0), 55) if rng.integers(0, 2) else ((0.1, 0.2, -6.5), (0.3, 0.1, -9.0), 80)
    return s


def teapot_spec():
    """The reference's teapot.dae (6,320 triangles, from the committed fixture) with mat6 (Game1.cs:109), the loader's
    fallback camera Camera((0,0,6), (0,0,0), 29) (ModelLoader.cs:105) and the two Game1 lights (Game1.cs:140-141)."""
    d = np.load(os.path.join(GOLDEN, "teapot_dae.npz"))
    s = SceneSpec()
    s.materials = [((1.0, 0.5, 0.5), 0.0, 0.85, (0, 0, 0))]
    s.meshes.append((d["pos9"], d["nrm9"], 0, False, True))
    s.lights = [((6.0, 2.0, 3.0), 5.0, None), ((0.0, 0.0, -12.0), 2.5, None)]
    s.camera = ((0, 0, 6), (0, 0, 0), 29)
    return s


def obj_spec():
    """tests/golden/tiny_scene.obj through the ModelLoader mirror (LoadModel + CreateScene, ModelLoader.cs:16-119): a
    pyramid and one triangle of a floor quad, material mat6 (Game1.cs:109), the loader's fallback camera (an OBJ file has
    none) and the two Game1 lights (it has no lights either)."""
    from simple_raytracer_amd import model_loader
    model = model_loader.LoadModel(os.path.join(GOLDEN, "tiny_scene.obj"))
    s = SceneSpec()
    s.materials = [((1.0, 0.5, 0.5), 0.0, 0.85, (0, 0, 0))]
    for mesh in model.Meshes:
        pos9, nrm9 = model_loader._mesh_arrays(mesh, (0.0, 0.0, 0.0))
        s.meshes.append((pos9, nrm9, 0, False, True))
    s.lights = [((6.0, 2.0, 3.0), 5.0, None), ((0.0, 0.0, -12.0), 2.5, None)]
    s.camera = ((0, 0, 6), (0, 0, 0), 29)
    return s


def max_lsb_diff(a, b):
    """Largest per-channel difference between two packed RGBA images, and the number of differing pixels."""
    da = orc.rgba_to_rgb8(np.asarray(a)).astype(np.int32)
    db = orc.rgba_to_rgb8(np.asarray(b)).astype(np.int32)
    d = np.abs(da - db)
    return int(d.max()) if d.size else 0, int((d.max(axis=-1) > 0).sum()) if d.size else 0


def psnr(a, b):
    da = orc.rgba_to_rgb8(np.asarray(a)).astype(np.float64)
    db = orc.rgba_to_rgb8(np.asarray(b)).astype(np.float64)
    mse = np.mean((da - db) ** 2)
    return float("inf") if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)


def oracle_scene_from_world(orc, world, lights):
    """The oracle's scene for a host-side world (SurfaceList of spheres, rects and triangle meshes) and LightList."""
    flat = world.flatten()
    mats = [orc.material(m.albedo, m.metalness, m.smoothness, m.emission) for m in flat["materials"]]
    tris = None
    if len(flat["tri_pos"]):
        tris = orc.tris_from_arrays(flat["tri_pos"], flat["tri_nrm"], 0)
        tris["material"] = flat["tri_mat"]
    sph = [orc.Sphere(orc.f3(s[:3]), float(s[3]), int(m)) for s, m in zip(flat["spheres"], flat["sphere_mat"])]
    ol = [orc.point_light(l.position, l.brightness, None) for l in lights.lights]
    return orc.Scene(mats, spheres=sph, tris=tris, lights=ol)


def oracle_camera_from(orc, cam):
    return orc.Camera(orc.f3(cam.position), orc.f3(cam.lowerLeftCorner), orc.f3(cam.horizontal), orc.f3(cam.vertical))
