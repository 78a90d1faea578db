"""Device time of the sphere-cloud scene (configuration 4's world) at 1080p for several sample counts: what a launch costs
beyond its samples (ramp, drain, per-level launches) shows as ms per sample falling with the count.

    python tools/spp_sweep_mesh.py [spp ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import simple_raytracer_amd as srt
from simple_raytracer_amd import scenes

W, H = 1920, 1080
r = srt.Renderer(device=0)
r.set_scene(*scenes.sphere_mesh_scene(W, H, 5, 10000), world_kind=srt.WORLD_SURFACELIST)
r.render(W, H, samples=16, max_depth=8, seed=1)
for spp in [int(a) for a in sys.argv[1:]] or [2, 4, 8, 16, 32]:
    ms = []
    for _ in range(3):
        r.render(W, H, samples=spp, max_depth=8, seed=1)
        ms.append(r.stats()["kernel_ms"])
    st = r.stats()
    print("spp %3d  min %.2f ms  per sample %.2f ms  level-by-level items %d" % (spp, min(ms), min(ms) / spp, st["wavefront_items"]), flush=True)
r.close()
