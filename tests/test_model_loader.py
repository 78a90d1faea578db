"""ModelLoader.cs mirror: the COLLADA subset reader and CreateModel / CreateScene with their quirks."""
import os

import numpy as np
import pytest

import helpers
from simple_raytracer_amd import game1, model_loader
from simple_raytracer_amd.raytracing import LightList, SurfaceList

TINY = os.path.join(helpers.GOLDEN, "tiny_scene.dae")


def test_load_model_applies_node_transforms_and_up_axis():
    sc = model_loader.LoadModel(TINY)
    assert len(sc.Meshes) == 1 and len(sc.Lights) == 1 and len(sc.Cameras) == 1
    m = sc.Meshes[0]
    assert m.Faces.tolist() == [[0, 1, 2], [3, 4, 5]]          # one vertex per face corner
    # corner (1, 1, 0) -> node matrix (scale 2, translate 10 20 30) -> (12, 22, 30) -> Z_UP root rotation (x, z, -y)
    assert m.Vertices[2].tolist() == [12.0, 30.0, -22.0]
    assert m.Vertices[0].tolist() == [10.0, 30.0, -20.0]
    # normal (0, 0, 2): scaled by the inverse transpose, normalised, rotated -> (0, 1, 0)
    assert np.allclose(m.Normals[0], [0, 1, 0], atol=1e-7)
    assert np.allclose(m.Normals[2], [0, 0.8, -0.6], atol=1e-6)
    assert sc.Lights[0].Position.tolist() == [4.0, 6.0, -5.0] and sc.Lights[0].ColorDiffuse.tolist() == [250.0, 125.0, 0.0]
    assert sc.Cameras[0].Position.tolist() == [1.0, 3.0, -2.0]
    assert np.allclose(sc.Cameras[0].Direction, [0, -1, 0], atol=1e-7)   # local -Z after the Z_UP rotation
    assert abs(sc.Cameras[0].FieldOfview - np.radians(40)) < 1e-6


def test_create_scene_quirks():
    sc = model_loader.LoadModel(TINY)
    mat6 = game1.materials()[5]
    world, lights = SurfaceList(), LightList()
    cam = model_loader.CreateScene(sc, mat6, world, lights, 1.5)
    flat = world.flatten()
    assert flat["tri_pos"].shape == (2, 9) and flat["tri_flags"].tolist() == [2, 2]      # flip false, blend true
    # ModelLoader.cs:99: brightness = ColorDiffuse.R / 125; the colour itself is not passed on
    assert len(lights.lights) == 1 and lights.lights[0].brightness == 2.0
    assert tuple(lights.lights[0].position) == (4.0, 6.0, -5.0)
    # ModelLoader.cs:113-114: the camera looks at MINUS (position + direction)
    ref = model_loader.Camera((1.0, 3.0, -2.0), (-1.0, -2.0, 2.0), float(np.float32(np.degrees(np.float32(np.radians(40))))), 1.5)
    for a in ("position", "lowerLeftCorner", "horizontal", "vertical"):
        assert np.allclose(getattr(cam, a), getattr(ref, a), atol=1e-6), a
    # no cameras in the file: the fallback of ModelLoader.cs:105
    sc.Cameras = []
    cam = model_loader.CreateScene(sc, mat6, SurfaceList(), LightList(), 1.5)
    ref = model_loader.Camera((0, 0, 6), (0, 0, 0), 29, 1.5)
    assert np.allclose(cam.lowerLeftCorner, ref.lowerLeftCorner)


def test_create_model_offsets_positions_and_renormalises_normals():
    sc = model_loader.Scene([model_loader.Mesh([[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 0, 2], [0, 3, 0], [4, 0, 0]], [[0, 1, 2]])])
    world = SurfaceList()
    model_loader.CreateModel(sc, game1.materials()[0], world, (1.0, 2.0, 3.0))
    flat = world.flatten()
    assert flat["tri_pos"][0].tolist() == [1, 2, 3, 2, 2, 3, 1, 3, 3]
    assert flat["tri_nrm"][0].tolist() == [0, 0, 1, 0, 1, 0, 1, 0, 0]


@pytest.mark.skipif(not os.path.exists("/root/reference/Content/models/teapot.dae"), reason="reference tree not present")
def test_reference_models_match_the_committed_fixtures():
    from simple_raytracer_amd import scenes
    for name, tris in (("teapot", 6320), ("sphere", 960)):
        sc = model_loader.LoadModel("/root/reference/Content/models/%s.dae" % name)
        world = SurfaceList()
        model_loader.CreateModel(sc, game1.materials()[5], world)
        flat = world.flatten()
        p, n = scenes.load_fixture_mesh(name)
        assert flat["tri_pos"].shape == (tris, 9)
        assert np.array_equal(flat["tri_pos"], p)
        assert np.allclose(flat["tri_nrm"], n, atol=2e-7)
