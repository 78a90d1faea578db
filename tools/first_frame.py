"""Wall time of the first and of later blocking renders of a scene in a fresh context (1080p): how much of the level-by-level
tracer's gain a host that renders one frame per scene sees.

    python tools/first_frame.py [c4|c5|c3] [spp]
"""
import sys
import time

sys.path.insert(0, ".")
import simple_raytracer_amd as srt  # noqa: E402
from simple_raytracer_amd import scenes  # noqa: E402

W, H = 1920, 1080
name = sys.argv[1] if len(sys.argv) > 1 else "c4"
make, depth, spp = {"c4": (lambda: scenes.sphere_mesh_scene(W, H, 5, 10000), 8, 16), "c5": (lambda: scenes.sphere_mesh_scene(W, H, 5), 8, 64),
                    "c3": (lambda: scenes.teapot_scene(W, H, 2), 4, 64)}[name]
if len(sys.argv) > 2:
    spp = int(sys.argv[2])
world, lights, cam = make()
r = srt.Renderer(device=0)
r.set_scene(world, lights, cam, world_kind=srt.WORLD_SURFACELIST)
for i in range(3):
    t = time.perf_counter()
    r.render(W, H, samples=spp, max_depth=depth, seed=1)
    wall = (time.perf_counter() - t) * 1e3
    st = r.stats()
    print("%s %d spp, render %d: wall %.1f ms, device %.1f ms, level-by-level items %d" % (name, spp, i + 1, wall, st["kernel_ms"], st["wavefront_items"]), flush=True)
