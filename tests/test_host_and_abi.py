"""Host-side mirror classes vs the oracle's constructors, scene flattening, and the C ABI surface of libsrt.so.
No compute call is made here (the container that runs these has no GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import helpers
import simple_raytracer_amd as srt
from simple_raytracer_amd import _native, game1


@pytest.mark.parametrize("origin,look,fov,w,h", [
    ((0, 0, 0), (-0.03, 0.875, -7.6), 63, 1280, 720),      # Game1.cs:96
    ((0, 0, 6), (0, 0, 0), 29, 1920, 1080),                # ModelLoader.cs:105
    ((-16.5, -9.55, 10), (-0.8, 1.25, -7.6), 15, 640, 360),  # Game1.cs:97 (commented alternative)
    ((-2, -1.2, 2), (-1, 0, -3), 32, 333, 777),
])
def test_camera_constructor_bit_exact(oracle, origin, look, fov, w, h):
    ratio = game1.aspect_ratio(w, h)
    ref = oracle.camera(origin, look, fov, float(ratio))
    cam = srt.Camera(origin, look, fov, ratio)
    for mine, theirs in ((cam.position, ref.position), (cam.lowerLeftCorner, ref.lower_left), (cam.horizontal, ref.horizontal),
                         (cam.vertical, ref.vertical)):
        assert np.array(mine, dtype=np.float32).tobytes() == np.array(list(theirs), dtype=np.float32).tobytes()


def test_material_and_light_constructors_match_oracle(oracle):
    for args in (((1.5, -0.2, 0.5), 2.0, -1.0, (50, 0.5, 0.5)), ((0.78, 0.79, 0.33), 0.0, 0.995, (0, 0, 0))):
        ref = oracle.material(*args)
        m = srt.Material(*args)
        assert list(m.albedo) == list(ref.albedo) and m.metalness == ref.metalness and m.smoothness == ref.smoothness
        assert list(m.emission) == list(ref.emission)
    for args in (((1, 2, 3), 5.0, None), ((1, 2, 3), 5.0, (1.25, 0, 0)), ((1, 2, 3), 5.0, (0.5, 0.2, 3.0)), ((1, 2, 3), 5.0, (0, 0, 0))):
        ref = oracle.point_light(*args)
        l = srt.PointLight(*args)
        assert np.float32(l.brightness) == np.float32(ref.brightness) and list(l.lightColor) == list(ref.color)


def test_flatten_buckets_by_type_and_keeps_order():
    m1, m2 = srt.Material((1, 0, 0), 0, 0.5, (0, 0, 0)), srt.Material((0, 1, 0), 1, 0.5, (0, 0, 0))
    world = srt.SurfaceList()
    world.surfaces.append(srt.Sphere((1, 2, 3), 4, m1))
    v = [srt.Vert((0, 0, 0), (0, 0, 1)), srt.Vert((1, 0, 0), (0, 0, 1)), srt.Vert((0, 1, 0), (0, 0, 1))]
    world.surfaces.append(srt.Tri(v[0], v[1], v[2], m2))
    world.surfaces.append(srt.RectAxis(-1, 1, -2, 2, -5, m1))
    world.surfaces.append(srt.Sphere((5, 6, 7), 8, m2))
    world.surfaces.append(srt.Tri(v[0], v[2], v[1], m1, True, False))
    flat = world.flatten()
    assert flat["spheres"].tolist() == [[1, 2, 3, 4], [5, 6, 7, 8]] and flat["sphere_mat"].tolist() == [0, 1]
    assert flat["rects"].tolist() == [[-1, 1, -2, 2, -5]] and flat["rect_mat"].tolist() == [0]
    assert flat["tri_pos"].shape == (2, 9) and flat["tri_mat"].tolist() == [1, 0] and flat["tri_flags"].tolist() == [2, 1]
    assert flat["tri_pos"][1].tolist() == [0, 0, 0, 0, 1, 0, 1, 0, 0]
    assert len(flat["materials"]) == 2


def _declared_symbols():
    text = open(os.path.join(helpers.ROOT, "include", "srt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(srt_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(name for name, _, _ in _native.SYMBOLS)


def test_library_loads_and_exports_every_declared_symbol():
    """libsrt.so (built in-tree for gfx950) must load on a machine without a GPU and export the whole ABI."""
    lib = _native.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.srt_version() == 300
    # struct sizes the header promises
    assert C.sizeof(_native.Material) == 32 and C.sizeof(_native.PointLight) == 28


def test_no_device_fails_loudly():
    """Without a usable gfx950 device there is no fallback: srt_create reports SRT_E_NO_DEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(srt.SrtError) as e:
        srt.Renderer(device=0)
    assert e.value.code == _native.E_NO_DEVICE


def test_local_pixels_needs_no_device():
    lib = _native.load()
    p = srt.Renderer.params(100, 50, layout=srt.LAYOUT_TILES, tile_w=32, tile_h=16, rank=1, world_size=3)
    # tile grid 4 x 4 = 16 tiles; rank 1 of 3 owns tiles 1,4,7,10,13 -> 5 tiles of 512 pixels
    assert lib.srt_local_pixels(C.byref(p)) == 5 * 512
    p = srt.Renderer.params(100, 50)
    assert lib.srt_local_pixels(C.byref(p)) == 5000
    bad = srt.Renderer.params(100, 50, tile_w=12)
    assert lib.srt_local_pixels(C.byref(bad)) == _native.E_INVALID
    neg = srt.Renderer.params(100, 50, color_clamp=-1.0)
    assert lib.srt_local_pixels(C.byref(neg)) == _native.E_REFERENCE   # Math.Clamp(x, 0, clamp) throws when clamp < 0
    deep = srt.Renderer.params(100, 50, max_depth=17)
    assert lib.srt_local_pixels(C.byref(deep)) == _native.E_LIMIT
