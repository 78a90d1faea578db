"""Host-side scene classes, mirroring the data classes of the reference's `Raytracing` namespace.

The reference keeps its scene as objects on the C# heap (RaytracingClasses.cs) and the trace loop
moves to the GPU; these classes are what stays on the host: same names, same constructor
arguments, same constructor arithmetic (in float32, in the C#'s operation order), no hit() methods.
`SurfaceList.flatten()` produces the structure-of-arrays buffers the C ABI takes.
"""
import math

import numpy as np

F = np.float32


def _v3(v):
    a = np.asarray(v, dtype=np.float32).reshape(3).copy()
    return a


def _normalize(v):
    # MonoGame Vector3.Normalize: factor = 1 / sqrt(x*x + y*y + z*z); three multiplies
    x, y, z = F(v[0]), F(v[1]), F(v[2])
    factor = F(1) / np.sqrt((x * x) + (y * y) + (z * z), dtype=np.float32)
    return np.array([x * factor, y * factor, z * factor], dtype=np.float32)


def _cross(a, b):
    ax, ay, az = F(a[0]), F(a[1]), F(a[2])
    bx, by, bz = F(b[0]), F(b[1]), F(b[2])
    return np.array([ay * bz - by * az, -(ax * bz - bx * az), ax * by - bx * ay], dtype=np.float32)


def _clamp(v, lo, hi):
    # System.Math.Clamp (NaN passes through)
    v = F(v)
    if v < lo:
        return F(lo)
    if v > hi:
        return F(hi)
    return v


def to_radians(deg):
    """MathHelper.ToRadians: (float)(degrees * 0.017453292519943295)."""
    return F(float(F(deg)) * 0.017453292519943295769236907684886)


class Camera:
    """RaytracingClasses.cs:32-64."""

    def __init__(self, inputOrigin, lookAt, inFov, ratio):
        self.position = _v3(inputOrigin)
        look = _v3(lookAt)
        theta = to_radians(inFov)
        h = F(math.tan(float(theta / F(2))))
        self.viewportHeight = F(2) * h
        self.viewportWidth = self.viewportHeight * F(ratio)
        w = _normalize(self.position - look)
        u = _normalize(_cross(np.array([0, 1, 0], dtype=np.float32), w))  # Vector3.Up
        v = _cross(w, u)
        self.horizontal = self.viewportWidth * u
        self.vertical = self.viewportHeight * v
        half = F(1) / F(2)  # Vector3 / float multiplies by the reciprocal
        self.lowerLeftCorner = self.position - self.horizontal * half - self.vertical * half - w


class Material:
    """RaytracingClasses.cs:221-262: albedo, metalness and smoothness are clamped to [0, 1]."""

    def __init__(self, inputAlbedo, inputMetal, inputSmooth, inputEmission, inMaxDepth=0):
        a = _v3(inputAlbedo)
        self.albedo = np.array([_clamp(a[0], 0, 1), _clamp(a[1], 0, 1), _clamp(a[2], 0, 1)], dtype=np.float32)
        self.metalness = _clamp(inputMetal, 0, 1)
        self.smoothness = _clamp(inputSmooth, 0, 1)
        self.emission = _v3(inputEmission)
        self.maxDepth = int(inMaxDepth)  # stored and never read by the reference


class Surface:
    """RaytracingClasses.cs:265-279."""


class Sphere(Surface):
    """RaytracingClasses.cs:449-500."""

    def __init__(self, inputOrigin, inputRadius, inputMat):
        self.origin = _v3(inputOrigin)
        self.center = self.origin
        self.radius = F(inputRadius)
        self.material = inputMat


class RectAxis(Surface):
    """RaytracingClasses.cs:503-544: an XY rectangle at z = k (its hit test evaluates t = -k - o.z/d.z)."""

    def __init__(self, inX1, inX2, inY1, inY2, inK, inputMat):
        self.x1, self.x2, self.y1, self.y2, self.k = F(inX1), F(inX2), F(inY1), F(inY2), F(inK)
        self.material = inputMat


class Vert:
    """RaytracingClasses.cs:682-692."""

    def __init__(self, inPos, inNorm):
        self.position = _v3(inPos)
        self.normal = _v3(inNorm)


class Tri(Surface):
    """RaytracingClasses.cs:547-595.  The three constructors differ only in defaults."""

    def __init__(self, v0, v1, v2, inputMat, inFlip=False, blendNormals=True):
        self.verts = (v0, v1, v2)
        self.material = inputMat
        self.flipNormal = bool(inFlip)
        self.blend = bool(blendNormals)


class TriMesh(Surface):
    """A run of Tri objects sharing one material, kept as arrays (what ModelLoader.CreateModel appends to
    world.surfaces, ModelLoader.cs:39-64, without a Python object per triangle)."""

    def __init__(self, pos9, nrm9, inputMat, inFlip=False, blendNormals=True):
        self.pos9 = np.ascontiguousarray(pos9, dtype=np.float32).reshape(-1, 9)
        self.nrm9 = np.ascontiguousarray(nrm9, dtype=np.float32).reshape(-1, 9)
        if self.pos9.shape != self.nrm9.shape:
            raise ValueError("pos9 and nrm9 differ in shape")
        self.material = inputMat
        self.flipNormal = bool(inFlip)
        self.blend = bool(blendNormals)


class Light:
    """RaytracingClasses.cs:67-73."""


class PointLight(Light):
    """RaytracingClasses.cs:75-94: the three-argument form divides brightness by the largest colour channel when >= 1."""

    def __init__(self, inputOrigin, inputBrightness, inputColor=None):
        self.position = _v3(inputOrigin)
        self.brightness = F(inputBrightness)
        if inputColor is None:
            self.lightColor = np.zeros(3, dtype=np.float32)
        else:
            self.lightColor = _v3(inputColor)
            c = self.lightColor
            biggest = max(max(c[0], c[1]), max(c[1], c[2]))
            if biggest != 0 and biggest >= 1:
                self.brightness = self.brightness / F(biggest)


class LightList(Light):
    """RaytracingClasses.cs:195-203."""

    def __init__(self):
        self.lights = []

    def init(self):
        self.lightArray = list(self.lights)


class SurfaceList(Surface):
    """RaytracingClasses.cs:282-342, data side only."""

    def __init__(self):
        self.surfaces = []

    def Init(self):
        self.surfaceArray = list(self.surfaces)

    def flatten(self):
        """-> dict of the arrays the C ABI takes.  Surfaces are bucketed by type, and inside a type the insertion order
        is kept; "order_type" / "order_index" name the surfaces in the order of world.surfaces itself (type 0 triangle,
        1 sphere, 2 rect; index into that type's array), which is what srt_set_surface_order takes."""
        mats, mat_index = [], {}

        def mid(m):
            if id(m) not in mat_index:
                mat_index[id(m)] = len(mats)
                mats.append(m)
            return mat_index[id(m)]

        sph, sph_m, rect, rect_m = [], [], [], []
        tp, tn, tm, tf = [], [], [], []
        ot, oi, n_tri = [], [], 0     # world.surfaces entry by entry: type (0 triangle, 1 sphere, 2 rect), index inside the type
        for s in self.surfaces:
            if isinstance(s, Sphere):
                ot.append(np.array([1], dtype=np.uint8)); oi.append(np.array([len(sph)], dtype=np.uint32))
                sph.append([s.origin[0], s.origin[1], s.origin[2], s.radius])
                sph_m.append(mid(s.material))
            elif isinstance(s, RectAxis):
                ot.append(np.array([2], dtype=np.uint8)); oi.append(np.array([len(rect)], dtype=np.uint32))
                rect.append([s.x1, s.x2, s.y1, s.y2, s.k])
                rect_m.append(mid(s.material))
            elif isinstance(s, Tri):
                ot.append(np.zeros(1, dtype=np.uint8)); oi.append(np.array([n_tri], dtype=np.uint32))
                n_tri += 1
                tp.append(np.concatenate([v.position for v in s.verts])[None, :])
                tn.append(np.concatenate([v.normal for v in s.verts])[None, :])
                tm.append(np.array([mid(s.material)], dtype=np.uint32))
                tf.append(np.array([(1 if s.flipNormal else 0) | (2 if s.blend else 0)], dtype=np.uint8))
            elif isinstance(s, TriMesh):
                n = s.pos9.shape[0]
                ot.append(np.zeros(n, dtype=np.uint8)); oi.append(np.arange(n_tri, n_tri + n, dtype=np.uint32))
                n_tri += n
                tp.append(s.pos9)
                tn.append(s.nrm9)
                tm.append(np.full(n, mid(s.material), dtype=np.uint32))
                tf.append(np.full(n, (1 if s.flipNormal else 0) | (2 if s.blend else 0), dtype=np.uint8))
            else:
                raise TypeError("unsupported surface %r" % type(s))
        cat = lambda parts, width, dt: (np.ascontiguousarray(np.concatenate(parts), dtype=dt) if parts
                                        else np.zeros((0, width) if width else (0,), dtype=dt))
        return {
            "materials": mats,
            "spheres": np.array(sph, dtype=np.float32).reshape(-1, 4), "sphere_mat": np.array(sph_m, dtype=np.uint32),
            "rects": np.array(rect, dtype=np.float32).reshape(-1, 5), "rect_mat": np.array(rect_m, dtype=np.uint32),
            "tri_pos": cat(tp, 9, np.float32), "tri_nrm": cat(tn, 9, np.float32),
            "tri_mat": cat(tm, 0, np.uint32), "tri_flags": cat(tf, 0, np.uint8),
            "order_type": cat(ot, 0, np.uint8), "order_index": cat(oi, 0, np.uint32),
        }
