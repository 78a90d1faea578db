"""Known-answer tests that pin the CPU oracle.

The reference ships no tests or golden vectors (parity unpinned against the real C# binary), so these
answers are derived by hand from the cited C# lines.
"""
import ctypes as C
import math

import numpy as np
import pytest

INF = float("inf")


def _rec():
    return (C.c_float * 9)()


def test_camera_default_scene(oracle):
    # Game1.cs:96 + RaytracingClasses.cs:38-49: fov 63 deg, 1280x720
    cam = oracle.default_camera(1280, 720)
    h = math.tan(math.radians(63.0) / 2)
    assert h == pytest.approx(0.612801, abs=1e-6)
    horiz, vert = np.array(cam.horizontal), np.array(cam.vertical)
    assert np.linalg.norm(vert) == pytest.approx(2 * h, rel=1e-6)                    # viewportHeight 1.225602
    assert np.linalg.norm(horiz) == pytest.approx(2 * h * 1280 / 720, rel=1e-6)      # viewportWidth 2.178848
    w = np.array([0.03, -0.875, 7.6])
    w /= np.linalg.norm(w)
    ll = -horiz / 2 - vert / 2 - w
    assert np.allclose(np.array(cam.lower_left), ll, atol=1e-6)
    assert abs(np.dot(horiz, vert)) < 1e-6 and abs(np.dot(horiz, w)) < 1e-6
    # u = Normalize(Cross(Up, w)) has no Y component
    assert horiz[1] == 0.0


def test_sphere_hit_outside_and_inside(oracle):
    L = oracle.lib()
    s = oracle.Sphere(oracle.f3((0, 0, -5)), 1.0, 3)
    rec = _rec()
    assert L.orc_sphere_hit(C.byref(s), oracle.f3((0, 0, 0)), oracle.f3((0, 0, -1)), 0.01, INF, rec) == 1
    assert list(rec) == [4.0, 0.0, 0.0, -4.0, 0.0, 0.0, 1.0, 1.0, 3.0]  # t, P, N, frontFace, material
    # from the centre: near root is negative, the far root t = 1 is taken; normal flipped to face the ray
    assert L.orc_sphere_hit(C.byref(s), oracle.f3((0, 0, -5)), oracle.f3((0, 0, -1)), 0.01, INF, rec) == 1
    assert rec[0] == 1.0 and list(rec[1:4]) == [0.0, 0.0, -6.0] and list(rec[4:7]) == [0.0, 0.0, 1.0] and rec[7] == 0.0
    # t_max below the near root rejects the sphere altogether (RaytracingClasses.cs:473-478)
    assert L.orc_sphere_hit(C.byref(s), oracle.f3((0, 0, 0)), oracle.f3((0, 0, -1)), 0.01, 3.5, rec) == 0
    # a ray that misses
    assert L.orc_sphere_hit(C.byref(s), oracle.f3((0, 0, 0)), oracle.f3((0, 1, 0)), 0.01, INF, rec) == 0


def test_sky_gradient(oracle):
    L = oracle.lib()
    out = (C.c_float * 3)()
    for dy, want in ((0.0, (0.225, 0.195, 0.2)), (1.0, (0.2, 0.17, 0.17)), (-1.0, (0.25, 0.22, 0.23))):
        L.orc_sky(oracle.f3((0, dy, 0)), out)
        assert np.allclose(list(out), want, atol=1e-7)


def test_depth_heuristic_table(oracle):
    # RaytracingOperations.cs:88-116.  d=4 -> 3 / 2 / 1 / 1 and d=8 -> 6 / 3 / 2 / 1 for s < .6 / .4 / .2 / <= .05
    L = oracle.lib()
    for d, want in ((4, (4, 3, 2, 1, 1)), (8, (8, 6, 3, 2, 1))):
        got = tuple(L.orc_depth_heuristic(d, s) for s in (0.995, 0.45, 0.3, 0.1, 0.05))
        assert got == want
    assert L.orc_depth_heuristic(1, 0.0) == 1 and L.orc_depth_heuristic(0, 0.9) == 1  # clamped to >= 1
    assert L.orc_depth_heuristic(4, 0.6) == 4 and L.orc_depth_heuristic(4, 0.5999) == 3


def test_light_list_doubles_the_running_level(oracle):
    # RaytracingClasses.cs:213: formerLightLevel += light.calculateLight(formerLightLevel, ...) => f <- 2 f + c_i
    mats = [oracle.material((0.8, 0.6, 0.4), 0.0, 0.5)]
    sph = [oracle.Sphere(oracle.f3((0, 0, -5)), 1.0, 0)]
    la, lb = oracle.point_light((3, -4, 0), 6.0), oracle.point_light((-2, -3, 1), 3.0)
    o, d = oracle.f3((0, 0, 0)), oracle.f3((0.05, -0.1, -1))
    L = oracle.lib()

    def lights_only(lights):
        sc = oracle.Scene(mats, spheres=sph, lights=lights)
        out = (C.c_float * 3)()
        assert L.orc_get_lights(sc.handle, oracle.WORLD_SURFACELIST, o, d, 1, 0, 0, 1, out) == 1
        return np.array(list(out), dtype=np.float64)
    ca, cb, cab = lights_only([la]), lights_only([lb]), lights_only([la, lb])
    assert ca.min() > 0 and cb.min() > 0
    assert np.allclose(cab, 2 * ca + cb, rtol=1e-5)
    assert not np.allclose(cab, ca + cb, rtol=1e-3)


def test_triangle_hit_and_backface_cull(oracle):
    L = oracle.lib()
    n = [0, 0, 1] * 3
    tri = oracle.Tri((C.c_float * 9)(0, 0, -1, 1, 0, -1, 0, 1, -1), (C.c_float * 9)(*n), 2, 0, 1)
    rec = _rec()
    o, d = oracle.f3((0.25, 0.25, 0)), oracle.f3((0, 0, -1))
    assert L.orc_tri_hit(C.byref(tri), o, d, 0.01, INF, rec) == 1
    assert rec[0] == 1.0 and list(rec[1:4]) == [0.25, 0.25, -1.0] and list(rec[4:7]) == [0.0, 0.0, 1.0] and rec[7] == 1.0
    rev = oracle.Tri((C.c_float * 9)(0, 0, -1, 0, 1, -1, 1, 0, -1), (C.c_float * 9)(*n), 2, 0, 1)
    assert L.orc_tri_hit(C.byref(rev), o, d, 0.01, INF, rec) == 0          # determinant < 1e-8 and not flipped
    rev.flip = 1
    assert L.orc_tri_hit(C.byref(rev), o, d, 0.01, INF, rec) == 1          # flipNormal accepts the other winding ...
    assert list(rec[4:7]) == [-0.0, -0.0, -1.0]                             # ... and negates the blended normal
    assert L.orc_tri_hit(C.byref(tri), oracle.f3((0.8, 0.8, 0)), d, 0.01, INF, rec) == 0   # u + v > 1
    assert L.orc_tri_hit(C.byref(tri), o, d, 0.01, 0.5, rec) == 0                           # beyond t_max
    # flat shading: Normalize(Cross(AB, AC)) ignores the vertex normals
    flat = oracle.Tri((C.c_float * 9)(0, 0, -1, 1, 0, -1, 0, 1, -1), (C.c_float * 9)(*([1, 0, 0] * 3)), 2, 0, 0)
    assert L.orc_tri_hit(C.byref(flat), o, d, 0.01, INF, rec) == 1 and list(rec[4:7]) == [0.0, 0.0, 1.0]


def test_rect_axis_formula_as_written(oracle):
    # RaytracingClasses.cs:518: t = -k - origin.Z / direction.Z  (not (k - origin.Z) / direction.Z)
    L = oracle.lib()
    r = oracle.Rect(-100, 100, -100, 100, -20.0, 1)
    rec = _rec()
    assert L.orc_rect_hit(C.byref(r), oracle.f3((0, 0, 0)), oracle.f3((0, 0, -1)), 0.01, INF, rec) == 1 and rec[0] == 20.0
    assert L.orc_rect_hit(C.byref(r), oracle.f3((0, 0, 5)), oracle.f3((0, 0, -1)), 0.01, INF, rec) == 1 and rec[0] == 25.0
    # direction (0, sqrt(3)/2, -1/2): as written t = 20 - 5/(-0.5) = 30, the geometric answer would be 50
    d = oracle.f3((0, math.sqrt(3) / 2, -0.5))
    assert L.orc_rect_hit(C.byref(r), oracle.f3((0, 0, 5)), d, 0.01, INF, rec) == 1
    assert rec[0] == pytest.approx(30.0, abs=1e-4)
    assert list(rec[4:7]) == [0.0, 0.0, 1.0] and rec[7] == 1.0   # d.z < 0: front face, outward normal (0,0,1)
    # outside the x range
    small = oracle.Rect(-6, -3, -4, 4, -20.0, 1)
    assert L.orc_rect_hit(C.byref(small), oracle.f3((0, 0, 0)), oracle.f3((0, 0, -1)), 0.01, INF, rec) == 0
    assert L.orc_rect_hit(C.byref(small), oracle.f3((-4, 0, 0)), oracle.f3((0, 0, -1)), 0.01, INF, rec) == 1


def test_aabb_slab_and_nan(oracle):
    L = oracle.lib()
    bmin, bmax = oracle.f3((1, -1, -1)), oracle.f3((2, 1, 1))
    assert L.orc_aabb_hit(bmin, bmax, oracle.f3((0, 0, 0)), oracle.f3((1, 0, 0))) == 1
    assert L.orc_aabb_hit(bmin, bmax, oracle.f3((3, 0, 0)), oracle.f3((1, 0, 0))) == 0     # box behind: t_max < 0
    assert L.orc_aabb_hit(bmin, bmax, oracle.f3((0, 5, 0)), oracle.f3((1, 0, 0))) == 0     # parallel, outside the slab
    # parallel to x, origin exactly on the y = 1 face: (1 - 1) * inf = NaN propagates through Math.Min/Max => counted as hit,
    # even though the ray points away from the box
    assert L.orc_aabb_hit(bmin, bmax, oracle.f3((5, 1, 0)), oracle.f3((1, 0, 0))) == 1
    # the caller's [t_min, t_max] is ignored: a box far beyond any t_max still reports a hit
    assert L.orc_aabb_hit(oracle.f3((1e6, -1, -1)), oracle.f3((1e6 + 1, 1, 1)), oracle.f3((0, 0, 0)), oracle.f3((1, 0, 0))) == 1


def test_color_packing(oracle):
    L = oracle.lib()
    assert L.orc_pack_color(0.5, 0.5, 0.5) == 0xFF7F7F7F          # (int)(0.5 * 255) = 127
    assert L.orc_pack_color(1.0, 0.0, 1.2) == 0xFFFF00FF          # 255 / 0 / clamped 306 -> 255
    assert L.orc_pack_color(-0.3, float("nan"), 0.999) == 0xFFFE0000  # negative and NaN -> 0; (int)254.745 = 254
    assert L.orc_pack_color(1e10, 0.0, 0.0) == 0xFF000000         # beyond int range: int.MinValue, clamped to 0
    assert L.orc_pack_color(0.2, 0.4, 0.6) == (51 | (102 << 8) | (153 << 16) | 0xFF000000)


def test_emissive_sphere_is_depth_one(oracle):
    # mat4 (Game1.cs:106): smoothness 0 => depth = 1 => all children return 0 => clamp(emission + light) => R = 1
    sc = oracle.default_scene()
    out = (C.c_float * 3)()
    L = oracle.lib()
    o = oracle.f3((0, -3, 0))
    d = oracle.f3((-20, 3, 0))  # straight at the red emissive sphere at (-20, 0, 0), clear of the blue one
    L.orc_get_ray_color(sc.handle, oracle.WORLD_BVHNODE, o, d, 4, 1.0, 1, 0, 0, out)
    assert out[0] == 1.0 and 0.5 <= out[1] <= 1.0 and 0.5 <= out[2] <= 1.0
    L.orc_get_ray_color(sc.handle, oracle.WORLD_BVHNODE, o, d, 4, 100.0, 1, 0, 0, out)
    assert out[0] >= 50.0   # unclamped: emission 50 plus light
    L.orc_get_ray_color(sc.handle, oracle.WORLD_BVHNODE, o, d, 0, 1.0, 1, 0, 0, out)
    assert list(out) == [0.0, 0.0, 0.0]   # depth < 1 returns black without tracing


def test_diffuse_and_reflection_debug_modes(oracle):
    # RaytracingOperations.cs:38-73.  One sphere of radius 1 at (0, 0, -5), no lights.
    mats = [oracle.material((0.5, 0.5, 0.5), 0.0, 0.5, (0, 0, 0))]
    sc = oracle.Scene(mats, spheres=[oracle.Sphere(oracle.f3((0, 0, -5)), 1.0, 0)])
    L = oracle.lib()
    out = (C.c_float * 3)()
    o = oracle.f3((0, 0, 0))
    for mode in (oracle.MODE_DIFFUSE, oracle.MODE_REFLECTION):
        # a ray straight up misses: t = 0.5 (1 + 1) = 1 => (0.85, 0.75, 1) exactly
        L.orc_get_debug_color(sc.handle, oracle.WORLD_SURFACELIST, mode, o, oracle.f3((0, 1, 0)), 5, 1, 0, 0, out)
        assert list(out) == [np.float32(0.85), np.float32(0.75), 1.0]
        # depth 1: the hit's only child runs at depth 0 and returns zero; depth 0 returns zero at once
        for depth in (1, 0):
            L.orc_get_debug_color(sc.handle, oracle.WORLD_SURFACELIST, mode, o, oracle.f3((0, 0, -1)), depth, 1, 0, 0, out)
            assert list(out) == [0.0, 0.0, 0.0]
    # reflection, head-on: hit at (0, 0, -4), normal (0, 0, 1), reflected direction (0, 0, 1) leaves the scene with
    # t = 0.5 (0 + 1) = 0.5 => 0.5 * (0.5 (0.85, 0.75, 1) + 0.5 (1, 1, 1)) = (0.4625, 0.4375, 0.5)
    L.orc_get_debug_color(sc.handle, oracle.WORLD_SURFACELIST, oracle.MODE_REFLECTION, o, oracle.f3((0, 0, -1)), 3, 1, 0, 0, out)
    assert np.allclose(list(out), [0.4625, 0.4375, 0.5], rtol=0, atol=1e-7)
    # diffuse, head-on: the child direction is normal + a unit vector, so it leaves through the hemisphere of the normal
    # (z >= 0 up to rounding) and misses: 0.5 * sky, each channel within the sky's range halved
    L.orc_get_debug_color(sc.handle, oracle.WORLD_SURFACELIST, oracle.MODE_DIFFUSE, o, oracle.f3((0, 0, -1)), 3, 1, 0, 0, out)
    assert 0.425 <= out[0] <= 0.5 and 0.375 <= out[1] <= 0.5 and out[2] == 0.5


def _pcg4d(x, y, z, w):
    M = 0xFFFFFFFF
    x, y, z, w = [(v * 1664525 + 1013904223) & M for v in (x, y, z, w)]
    x = (x + y * w) & M; y = (y + z * x) & M; z = (z + x * y) & M; w = (w + y * z) & M
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16
    x = (x + y * w) & M; y = (y + z * x) & M; z = (z + x * y) & M; w = (w + y * z) & M
    return [x, y, z, w]


def test_counter_rng_matches_its_definition(oracle):
    L = oracle.lib()
    out = (C.c_uint32 * 4)()
    rng = np.random.default_rng(0)
    for _ in range(200):
        seed, pixel, sample, node, slot = [int(v) for v in rng.integers(0, 2 ** 32, size=5)]
        L.orc_rng_words(seed, pixel, sample, node, slot, out)
        assert list(out) == _pcg4d(pixel, sample, node, (slot + 0x9E3779B9 * seed) & 0xFFFFFFFF)
    # Next(lo, lo + B) = lo + next base-B digit of word / 2^32: covers [lo, lo + B) and nothing else
    for word, lo, rng_, want in ((0, -100, 200, -100), (0xFFFFFFFF, -100, 200, 99), (0x80000000, -50, 100, 0), (0xFFFFFFFF, -50, 100, 49)):
        w = C.c_uint32(word)
        assert L.orc_rng_next(C.byref(w), lo, rng_) == want
    # successive draws from one word are the base-B digits of word / 2^32: 0.(100)(50)(7)... in base 200
    w = C.c_uint32(int((100 + (50 + (7 + 0.5) / 200) / 200) / 200 * 2 ** 32))
    assert [L.orc_rng_next(C.byref(w), 0, 200) for _ in range(3)] == [100, 50, 7]
    # ... and they are uniform and pairwise independent (chi-square over a 20 x 20 coarse grid of (d1, d2), (d2, d3))
    rng = np.random.default_rng(1)
    counts = np.zeros((2, 20, 20))
    n = 200000
    for word in rng.integers(0, 2 ** 32, size=n):
        w = C.c_uint32(int(word))
        d = [L.orc_rng_next(C.byref(w), 0, 200) for _ in range(3)]
        counts[0, d[0] // 10, d[1] // 10] += 1
        counts[1, d[1] // 10, d[2] // 10] += 1
    expected = n / 400.0
    chi2 = ((counts - expected) ** 2 / expected).sum(axis=(1, 2))
    assert (chi2 < 399 + 6 * np.sqrt(2 * 399)).all(), chi2   # 399 degrees of freedom, 6 sigma


def test_random_vector_helpers(oracle):
    L = oracle.lib()
    v = (C.c_float * 3)()
    words = (C.c_uint32 * 4)()
    grid = {np.float32(k) / np.float32(100) for k in range(-100, 100)}
    seen = set()
    for i in range(500):
        L.orc_rng_words(5, i, 0, 1, 1, words)
        for w in words:
            L.orc_random_vector(w, v)
            assert all(np.float32(c) in grid for c in v)
            seen.update(np.float32(c) for c in v)
    assert len(seen) == 200                                  # every value of Next(-100,100)/100 shows up
    n = oracle.f3((0.0, -0.6, 0.8))
    for i in range(2000):
        L.orc_hemisphere_vector(5, i, 3, 7, n, v)
        assert v[0] * v[0] + v[1] * v[1] + v[2] * v[2] < 1.0  # inside the unit sphere
        assert v[0] * n[0] + v[1] * n[1] + v[2] * n[2] >= -1e-6  # flipped into the normal's hemisphere (float32 dot)


def test_angle_between_is_180_minus_angle(oracle):
    L = oracle.lib()
    assert L.orc_angle_between(oracle.f3((1, 0, 0)), oracle.f3((1, 0, 0))) == pytest.approx(180.0, abs=1e-4)
    assert L.orc_angle_between(oracle.f3((1, 0, 0)), oracle.f3((0, 1, 0))) == pytest.approx(90.0, abs=1e-4)
    assert L.orc_angle_between(oracle.f3((1, 0, 0)), oracle.f3((-1, 0, 0))) == pytest.approx(0.0, abs=1e-4)


def test_material_and_light_constructors(oracle):
    m = oracle.material((1.5, -0.2, 0.5), 2.0, -1.0, (50, 0.5, 0.5))
    assert list(m.albedo) == [1.0, 0.0, 0.5] and m.metalness == 1.0 and m.smoothness == 0.0 and list(m.emission) == [50.0, 0.5, 0.5]
    l2 = oracle.point_light((1, 2, 3), 5.0)
    assert l2.brightness == 5.0 and list(l2.color) == [0, 0, 0]
    l3 = oracle.point_light((1, 2, 3), 5.0, (1.25, 0, 0))      # RaytracingClasses.cs:90-93: brightness /= 1.25
    assert l3.brightness == 4.0 and list(l3.color) == [1.25, 0, 0]
    l4 = oracle.point_light((1, 2, 3), 5.0, (0.5, 0.2, 0.1))   # largest channel < 1: unchanged
    assert l4.brightness == 5.0


def test_reference_bvh_shape_of_default_scene(oracle):
    sc = oracle.default_scene()
    # 5 spheres: root(5) -> left(2) leaf pair, right(3) -> left(1), right(2): five BVHNode objects, depth 3
    assert sc.bvh_nodes() == 5 and sc.bvh_depth() == 3
    order = sc.bvh_leaf_order()
    assert sorted(order.tolist()) == [0, 1, 2, 3, 4]


def test_rect_world_cannot_build_bvhnode(oracle):
    mats = [oracle.material((0.5, 0.5, 0.5), 0, 0.5)]
    sc = oracle.Scene(mats, spheres=[oracle.Sphere(oracle.f3((0, 0, -5)), 1.0, 0)], rects=[oracle.Rect(-1, 1, -1, 1, -3.0, 0)])
    cam = oracle.default_camera(16, 8)
    with pytest.raises(RuntimeError):
        sc.render(cam, 16, 8, world_kind=oracle.WORLD_BVHNODE)
    img, _ = sc.render(cam, 16, 8, world_kind=oracle.WORLD_SURFACELIST)
    assert img.shape == (8, 16)


def test_drop_remainder_rows_quirk(oracle):
    # Game1.cs:378: thread i renders rows [(H/N)*i, (H/N)*(i+1)); H mod N rows stay default(Color) = 0
    sc = oracle.default_scene()
    cam = oracle.default_camera(16, 10)
    img, _ = sc.render(cam, 16, 10, samples=1, threads=4, drop_remainder_rows=True)
    assert (img[8:] == 0).all() and (img[:8] != 0).all()
    full, _ = sc.render(cam, 16, 10, samples=1, threads=4)
    assert (full[:8] == img[:8]).all() and (full[8:] != 0).all()
    one, _ = sc.render(cam, 16, 10, samples=1, threads=1)
    assert (one == full).all()   # the counter RNG makes the image independent of the thread count
