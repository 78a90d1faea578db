"""simple_raytracer_amd -- the per-pixel trace loop of helpvisa/simple_raytracer on MI355X (gfx950).

The compute path is libsrt.so (hand-written HIP kernels behind the C ABI of include/srt.h); this package
is the host side: scene classes mirroring the reference's, the Game1 scene literals, and the render driver.
"""
from . import _native, model_loader
from ._native import (LAYOUT_FRAME, LAYOUT_TILES, MODE_COLOR, MODE_DEPTH, MODE_DIFFUSE, MODE_LIGHTS, MODE_NORMAL, MODE_REFLECTION, WORLD_BVHNODE,
                      WORLD_SURFACELIST, GATHER_RCCL, GATHER_PEER_COPY, SrtError)
from .raytracing import (Camera, Light, LightList, Material, PointLight, RectAxis, Sphere, Surface, SurfaceList, Tri, TriMesh,
                         Vert)
from .renderer import Renderer, rgba_to_rgb8
from .sorting import Sorting

__all__ = ["Sorting", "Camera", "Light", "LightList", "Material", "PointLight", "RectAxis", "Sphere", "Surface", "SurfaceList", "Tri",
           "TriMesh", "Vert", "Renderer", "rgba_to_rgb8", "SrtError", "WORLD_BVHNODE", "WORLD_SURFACELIST", "MODE_COLOR",
           "MODE_NORMAL", "MODE_DEPTH", "MODE_LIGHTS", "MODE_DIFFUSE", "MODE_REFLECTION", "LAYOUT_FRAME", "LAYOUT_TILES", "GATHER_RCCL", "GATHER_PEER_COPY"]
