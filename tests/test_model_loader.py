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


TINY_OBJ = os.path.join(helpers.GOLDEN, "tiny_scene.obj")


def test_load_model_reads_wavefront_obj():
    sc = model_loader.LoadModel(TINY_OBJ)
    assert len(sc.Lights) == 0 and len(sc.Cameras) == 0
    # a mesh per run of faces under one o / g / usemtl, a vertex per face corner
    assert [m.Faces.tolist() for m in sc.Meshes] == [[[0, 1, 2], [3, 4, 5]], [[0, 1, 2], [3, 4, 5]], [[0, 1, 2]]]
    front, back, floor = sc.Meshes
    assert np.array_equal(front.Vertices[[0, 1, 2]], np.float32([[-1.0, -0.6, -5.0], [1.0, -0.6, -5.0], [0.0, 1.2, -6.0]]))       # f 1//1 2//1 5//5
    assert np.allclose(front.Normals[[0, 2]], [[0.0, 0.5145, 0.8575], [0.0, 2.0, 0.0]])
    assert np.array_equal(front.Vertices[3], np.float32([1.0, -0.6, -5.0])) and np.allclose(front.Normals[3], [0.8575, 0.5145, 0.0])   # v/vt/vn
    # negative indices count back from the end of what has been read so far: -5 is vertex 1, -1 vertex 5
    assert np.array_equal(back.Vertices[[3, 4, 5]], np.float32([[-1.0, -0.6, -5.0], [1.0, -0.6, -5.0], [0.0, 1.2, -6.0]]))
    # the quad keeps its four corners, but only the first three indices of a face are read (ModelLoader.cs:60)
    assert floor.Vertices.shape == (4, 3) and floor.Faces.tolist() == [[0, 1, 2]]
    world, lights = SurfaceList(), LightList()
    cam = model_loader.CreateScene(sc, game1.materials()[5], world, lights, 16 / 9)
    flat = world.flatten()
    assert flat["tri_pos"].shape == (5, 9) and flat["tri_flags"].tolist() == [2] * 5
    assert np.allclose(np.linalg.norm(flat["tri_nrm"].reshape(-1, 3), axis=1), 1.0, atol=1e-6)          # Vector3.Normalize
    ref = model_loader.Camera((0, 0, 6), (0, 0, 0), 29, 16 / 9)                                         # no camera in the file
    assert np.allclose(cam.lowerLeftCorner, ref.lowerLeftCorner)


def test_load_model_refuses_what_the_reference_would_trip_over(tmp_path):
    p = tmp_path / "no_normals.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    with pytest.raises(ValueError):
        model_loader.LoadModel(str(p))           # mesh.Normals is empty: normalList[i] throws in the reference
    p = tmp_path / "line.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nvn 0 0 1\nf 1//1 2//1\n")
    with pytest.raises(ValueError):
        model_loader.LoadModel(str(p))           # indices[2] does not exist
    with pytest.raises(ValueError):
        model_loader.LoadModel(str(tmp_path / "model.fbx"))


def test_collada_camera_reports_the_horizontal_field_of_view(tmp_path):
    """Assimp derives aiCamera::mHorizontalFOV: xfov as given, else from yfov and the aspect ratio, else 45 degrees."""
    text = open(TINY).read()
    assert abs(model_loader.LoadModel(TINY).Cameras[0].FieldOfview - np.radians(40)) < 1e-6
    p = tmp_path / "yfov.dae"
    p.write_text(text.replace('<xfov sid="xfov">40</xfov>', '<yfov sid="yfov">40</yfov>'))
    want = 2 * np.arctan(1.5 * np.tan(np.radians(40) / 2))
    assert abs(model_loader.LoadModel(str(p)).Cameras[0].FieldOfview - want) < 1e-6
    p.write_text(text.replace('<xfov sid="xfov">40</xfov>', '<yfov sid="yfov">40</yfov>').replace("<aspect_ratio>1.5</aspect_ratio>", ""))
    assert abs(model_loader.LoadModel(str(p)).Cameras[0].FieldOfview - np.radians(45)) < 1e-6


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
