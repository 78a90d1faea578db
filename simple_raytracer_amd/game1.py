"""The scene and settings Game1.Initialize hard-codes (Game1.cs:41-43, 78-79, 96-144)."""
import numpy as np

from .raytracing import Camera, LightList, Material, PointLight, RectAxis, SurfaceList, Sphere

SAMPLES = 500        # Game1.cs:41
MAX_DEPTH = 4        # Game1.cs:42
COLOR_CLAMP = 1.0    # Game1.cs:43
WIDTH, HEIGHT = 1280, 720  # Game1.cs:78-79


def aspect_ratio(width, height):
    return np.float32(width) / np.float32(height)  # Game1.cs:86


def materials():
    """mat1..mat6 of Game1.cs:102-109."""
    zero = (0, 0, 0)
    return [Material((0.78, 0.79, 0.33), 0.0, 0.995, zero),
            Material((1.0, 0.65, 0.62), 1.0, 0.65, zero),
            Material((0.5, 0.425, 0.15), 0.0, 0.45, zero),
            Material(zero, 0, 0.0, (50.0, 0.5, 0.5)),
            Material(zero, 0, 0.0, (0.5, 0.5, 50.0)),
            Material((1.0, 0.5, 0.5), 0.0, 0.85, zero)]


def default_scene(width=WIDTH, height=HEIGHT, with_rect=False):
    """-> (world, lights, camera).  `with_rect` adds the RectAxis the reference keeps commented out (Game1.cs:134);
    such a world can only be traced as a SurfaceList."""
    mat1, mat2, mat3, mat4, mat5, _mat6 = materials()
    camera = Camera((0, 0, 0), (-0.03, 0.875, -7.6), 63, aspect_ratio(width, height))
    world = SurfaceList()
    world.surfaces.append(Sphere((2.05, 0.9, -7.25), 3.5, mat1))
    world.surfaces.append(Sphere((-4.5, 2.0, -8.0), 2.3, mat2))
    world.surfaces.append(Sphere((0, 45, -20.0), 42.5, mat3))
    world.surfaces.append(Sphere((-20, 0, 0.0), 2, mat4))
    world.surfaces.append(Sphere((-15, 0, 6.0), 6, mat5))
    if with_rect:
        world.surfaces.append(RectAxis(-6, -3, -4, 4, -20, mat4))
    world.Init()
    lights = LightList()
    lights.lights.append(PointLight((6.0, 2.0, 3.0), 5.0))
    lights.lights.append(PointLight((0.0, 0.0, -12.0), 2.5))
    lights.init()
    return world, lights, camera
