"""Host-side mirror of ModelLoader.ModelOperations (ModelLoader.cs:16-119): meshes, lights and a camera from a model
file into the world the tracer renders.

The reference imports files with Assimp (AssimpContext.ImportFile with PostProcessSteps.PreTransformVertices,
ModelLoader.cs:32; any path Assimp can read), a third-party library that is not part of the reference tree.
`LoadModel` here reads two formats and returns the same shape of data Assimp would (`Scene.Meshes[i].Vertices / Normals /
Faces`, `Scene.Lights`, `Scene.Cameras`), with the node transforms already applied to the vertices ("pre-transformed"):

* `.dae`: the subset of COLLADA 1.4 that the reference's own models use -- <triangles> / <polylist> of triangles with
  VERTEX and NORMAL inputs, node <matrix> transforms, point lights, perspective cameras, the Z_UP root rotation.
  Transforms are evaluated in double and rounded once; Assimp's own float rounding is not reproduced.
* `.obj`: Wavefront `v` / `vn` / `vt` / `f` with the index forms v, v/vt, v//vn and v/vt/vn (negative indices count from
  the end), one mesh per run of faces under one `o` / `g` / `usemtl`, one vertex per face corner.  Faces keep all their
  corners: the reference imports without the Triangulate step and then reads the first three indices of every face
  (ModelLoader.cs:60), so a quad contributes one triangle.  An OBJ file holds no lights, cameras or transforms.

What the importer itself does -- which corners share a vertex, how meshes are split, the camera angle it derives from a
COLLADA <perspective>, where it puts a light -- is restated from Assimp 4.1's published behaviour and is not pinned by
anything in the reference tree (parity with the importer is unpinned):
  - a COLLADA camera reports its HORIZONTAL field of view (aiCamera::mHorizontalFOV): <xfov> as given; with only <yfov>
    and <aspect_ratio>, 2 atan(aspect tan(yfov / 2)); with <yfov> alone, aiCamera's default of 45 degrees.  The
    reference then hands that angle to its Camera as the VERTICAL field of view (ModelLoader.cs:111-114);
  - lights and cameras are not touched by PreTransformVertices' vertex baking; their position / direction are taken in
    world space (node transforms and the up-axis rotation applied), which is what aiLight / aiCamera hold after that step
    moved every node transform into the data.

`CreateModel` and `CreateScene` then do what ModelLoader.cs:39-119 does with that data, quirks included.
"""
import xml.etree.ElementTree as ET

import numpy as np

from .raytracing import Camera, PointLight, TriMesh

_NS = {"c": "http://www.collada.org/2005/11/COLLADASchema"}
_Z_UP = np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, 0], [0, 0, 0, 1]], dtype=np.float64)   # (x, y, z) -> (x, z, -y)
_X_UP = np.array([[0, -1, 0, 0], [1, 0, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)   # (x, y, z) -> (-y, x, z)


class Mesh:
    """Assimp.Mesh: one vertex per face corner (the importer does not join identical vertices unless asked to)."""

    def __init__(self, vertices, normals, faces):
        self.Vertices = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        self.Normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        self.Faces = np.ascontiguousarray(faces, dtype=np.int64).reshape(-1, 3)


class ImportedLight:
    def __init__(self, position, color_diffuse):
        self.Position = np.asarray(position, dtype=np.float32)
        self.ColorDiffuse = np.asarray(color_diffuse, dtype=np.float32)   # R, G, B


class ImportedCamera:
    def __init__(self, position, direction, field_of_view):
        self.Position = np.asarray(position, dtype=np.float32)
        self.Direction = np.asarray(direction, dtype=np.float32)
        self.FieldOfview = float(field_of_view)   # radians, as Assimp reports it


class Scene:
    def __init__(self, meshes=(), lights=(), cameras=()):
        self.Meshes, self.Lights, self.Cameras = list(meshes), list(lights), list(cameras)


def _floats(text):
    return np.array(text.split(), dtype=np.float64)


def _node_matrix(node):
    """Product of the node's transform elements in document order (<matrix>, <translate>, <scale>, <rotate>)."""
    M = np.eye(4)
    for el in node:
        tag = el.tag.split("}")[-1]
        if tag == "matrix":
            M = M @ _floats(el.text).reshape(4, 4)
        elif tag == "translate":
            T = np.eye(4)
            T[:3, 3] = _floats(el.text)
            M = M @ T
        elif tag == "scale":
            M = M @ np.diag(list(_floats(el.text)) + [1.0])
        elif tag == "rotate":
            x, y, z, deg = _floats(el.text)
            a = np.array([x, y, z])
            a = a / (np.linalg.norm(a) or 1.0)
            c, s = np.cos(np.radians(deg)), np.sin(np.radians(deg))
            K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
            R = np.eye(4)
            R[:3, :3] = c * np.eye(3) + s * K + (1 - c) * np.outer(a, a)
            M = M @ R
    return M


def _read_geometries(root):
    geoms = {}
    for g in root.findall("c:library_geometries/c:geometry", _NS):
        mesh = g.find("c:mesh", _NS)
        if mesh is None:
            continue
        sources = {}
        for s in mesh.findall("c:source", _NS):
            fa = s.find("c:float_array", _NS)
            acc = s.find("c:technique_common/c:accessor", _NS)
            if fa is None or acc is None:
                continue
            sources["#" + s.get("id")] = _floats(fa.text).reshape(-1, int(acc.get("stride")))
        verts = mesh.find("c:vertices", _NS)
        vmap = {"#" + verts.get("id"): verts.find("c:input[@semantic='POSITION']", _NS).get("source")}
        pos_all, nrm_all = [], []
        for prim in list(mesh.findall("c:triangles", _NS)) + list(mesh.findall("c:polylist", _NS)):
            inputs = {i.get("semantic"): (int(i.get("offset")), i.get("source")) for i in prim.findall("c:input", _NS)}
            if "VERTEX" not in inputs or "NORMAL" not in inputs:
                raise ValueError("geometry %s: primitives need VERTEX and NORMAL inputs (the reference reads mesh.Normals)" % g.get("id"))
            vcount = prim.find("c:vcount", _NS)
            if vcount is not None and any(int(v) != 3 for v in vcount.text.split()):
                raise ValueError("geometry %s: only triangles are supported (the reference indexes three corners per face)" % g.get("id"))
            n_off = max(o for o, _ in inputs.values()) + 1
            idx = np.array(prim.find("c:p", _NS).text.split(), dtype=np.int64).reshape(-1, 3, n_off)
            voff, vsrc = inputs["VERTEX"]
            noff, nsrc = inputs["NORMAL"]
            pos_all.append(sources[vmap[vsrc]][idx[:, :, voff]][..., :3])
            nrm_all.append(sources[nsrc][idx[:, :, noff]][..., :3])
        if pos_all:
            geoms["#" + g.get("id")] = (np.concatenate(pos_all), np.concatenate(nrm_all))
    return geoms


def _horizontal_fov(persp):
    """aiCamera::mHorizontalFOV (radians) of a COLLADA <perspective>, as Assimp 4.1's loader derives it."""
    if persp is None:
        return np.radians(45.0)
    x, y, a = persp.find("c:xfov", _NS), persp.find("c:yfov", _NS), persp.find("c:aspect_ratio", _NS)
    if x is not None:
        return np.radians(float(x.text))
    if y is not None and a is not None:
        return 2.0 * np.arctan(float(a.text) * np.tan(np.radians(float(y.text)) * 0.5))
    return np.radians(45.0)


def _load_obj(loadPath):
    """Wavefront OBJ -> Scene: what Assimp's importer hands back without the Triangulate and JoinIdenticalVertices steps."""
    pos, nrm = [], []
    meshes, faces = [], []          # faces of the mesh being read: lists of (v, vn) index pairs

    def close():
        if not faces:
            return
        verts, normals, idx = [], [], []
        for f in faces:
            corners = []
            for v, n in f:
                if n is None:
                    raise ValueError("%s: a face without normals (the reference indexes mesh.Normals, which is then empty)" % loadPath)
                corners.append(len(verts))
                verts.append(pos[v])
                normals.append(nrm[n])
            idx.append(corners[:3])     # ModelLoader.cs:60 reads indices[0..2] only
        meshes.append(Mesh(verts, normals, idx))
        faces.clear()

    def resolve(token, count):
        i = int(token)
        i = i - 1 if i > 0 else count + i
        if not 0 <= i < count:
            raise ValueError("%s: index %s out of range" % (loadPath, token))
        return i

    with open(loadPath, "r", encoding="utf-8", errors="replace") as fh:
        for raw in fh:
            line = raw.split("#", 1)[0].split()
            if not line:
                continue
            key = line[0]
            if key == "v":
                pos.append([float(x) for x in line[1:4]])
            elif key == "vn":
                nrm.append([float(x) for x in line[1:4]])
            elif key in ("o", "g", "usemtl"):
                close()
            elif key == "f":
                if len(line) - 1 < 3:
                    raise ValueError("%s: a face with fewer than three corners (the reference indexes three)" % loadPath)
                corners = []
                for tok in line[1:]:
                    parts = tok.split("/")
                    v = resolve(parts[0], len(pos))
                    n = resolve(parts[2], len(nrm)) if len(parts) > 2 and parts[2] else None
                    corners.append((v, n))
                faces.append(corners)
    close()
    return Scene(meshes)


def LoadModel(loadPath):
    """ModelLoader.cs:16-37.  Returns a Scene with pre-transformed meshes, lights and cameras."""
    if str(loadPath).lower().endswith(".obj"):
        return _load_obj(loadPath)
    if not str(loadPath).lower().endswith(".dae"):
        raise ValueError("%s: only COLLADA (.dae) and Wavefront (.obj) files are read" % loadPath)
    root = ET.parse(loadPath).getroot()
    up = root.find("c:asset/c:up_axis", _NS)
    up = up.text.strip() if up is not None and up.text else "Y_UP"
    root_m = _Z_UP if up == "Z_UP" else (_X_UP if up == "X_UP" else np.eye(4))
    geoms = _read_geometries(root)
    lights_lib = {}
    for l in root.findall("c:library_lights/c:light", _NS):
        col = l.find("c:technique_common/*/c:color", _NS)
        lights_lib["#" + l.get("id")] = _floats(col.text)[:3] if col is not None else np.ones(3)
    cams_lib = {}
    for c in root.findall("c:library_cameras/c:camera", _NS):
        persp = c.find("c:optics/c:technique_common/c:perspective", _NS)
        cams_lib["#" + c.get("id")] = _horizontal_fov(persp)

    scene = Scene()

    def walk(node, parent):
        M = parent @ _node_matrix(node)
        for inst in node.findall("c:instance_geometry", _NS):
            if inst.get("url") not in geoms:
                continue
            p, n = geoms[inst.get("url")]
            p = p @ M[:3, :3].T + M[:3, 3]
            n = n @ np.linalg.inv(M[:3, :3])           # inverse transpose, applied to row vectors
            ln = np.linalg.norm(n, axis=-1, keepdims=True)
            n = n / np.where(ln == 0, 1.0, ln)
            count = p.shape[0] * 3
            scene.Meshes.append(Mesh(p.reshape(-1, 3), n.reshape(-1, 3), np.arange(count).reshape(-1, 3)))
        for inst in node.findall("c:instance_light", _NS):
            scene.Lights.append(ImportedLight(M[:3, 3], lights_lib.get(inst.get("url"), np.ones(3))))
        for inst in node.findall("c:instance_camera", _NS):
            # a COLLADA camera looks down its local -Z
            scene.Cameras.append(ImportedCamera(M[:3, 3], M[:3, :3] @ np.array([0.0, 0.0, -1.0]), cams_lib.get(inst.get("url"), np.radians(45.0))))
        for child in node.findall("c:node", _NS):
            walk(child, M)

    for vs in root.findall("c:library_visual_scenes/c:visual_scene", _NS):
        for node in vs.findall("c:node", _NS):
            walk(node, root_m)
    return scene


def _mesh_arrays(mesh, offset):
    """Per-face corner positions / normals the way ModelLoader.cs:50-61 builds its Vert list: position + offset (three
    float additions), normal re-normalised with Vector3.Normalize (factor = 1 / sqrt(x^2 + y^2 + z^2), three products)."""
    off = np.asarray(offset, dtype=np.float32)
    pos = mesh.Vertices + off
    n = mesh.Normals
    with np.errstate(divide="ignore", invalid="ignore"):
        factor = np.float32(1.0) / np.sqrt((n[:, 0] * n[:, 0] + n[:, 1] * n[:, 1]) + n[:, 2] * n[:, 2], dtype=np.float32)
    nrm = (n * factor[:, None]).astype(np.float32)
    f = mesh.Faces
    return np.ascontiguousarray(pos[f].reshape(-1, 9)), np.ascontiguousarray(nrm[f].reshape(-1, 9))


def CreateModel(model, inMat, world, offset=(0.0, 0.0, 0.0)):
    """ModelLoader.cs:39-64: every face of every mesh becomes Tri(v0, v1, v2, inMat, false, true), moved by `offset`."""
    for mesh in model.Meshes:
        pos9, nrm9 = _mesh_arrays(mesh, offset)
        world.surfaces.append(TriMesh(pos9, nrm9, inMat, False, True))


def CreateScene(model, inMat, world, lights, ratio):
    """ModelLoader.cs:66-119: meshes as in CreateModel without offset; every imported light becomes
    PointLight(position, ColorDiffuse.R / 125) (the colour is dropped, the red channel alone sets the brightness);
    the camera is Camera((0,0,6), (0,0,0), 29, ratio) unless the file has cameras, in which case the last one wins and
    looks at MINUS (position + direction) (:113-114)."""
    CreateModel(model, inMat, world, (0.0, 0.0, 0.0))
    for light in model.Lights:
        lights.lights.append(PointLight(tuple(float(v) for v in light.Position), float(np.float32(light.ColorDiffuse[0]) / np.float32(125))))
        lights.init()
    cam = Camera((0, 0, 6), (0, 0, 0), 29, ratio)
    for camera in model.Cameras:
        pos = camera.Position
        look_at = pos + camera.Direction
        fov = float(np.float32(np.degrees(np.float32(camera.FieldOfview))))
        cam = Camera(tuple(float(v) for v in pos), tuple(float(-v) for v in look_at), fov, ratio)
    return cam
