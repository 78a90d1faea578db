"""Frame time against samples per pixel on the mesh configurations (see tools/spp_sweep.py).

    python tools/mesh_spp_sweep.py [c3|c5|c4 ...]
"""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import simple_raytracer_amd as srt  # noqa: E402
from big_scenes import CONFIGS, W, H  # noqa: E402

r = srt.Renderer(device=0)
for name in [a for a in sys.argv[1:] if a in CONFIGS] or ["c5"]:
    make, depth = CONFIGS[name]
    world, lights, cam = make()
    r.set_scene(world, lights, cam, world_kind=srt.WORLD_SURFACELIST)
    r.render(W, H, samples=1, max_depth=depth, seed=1)
    for spp in (1, 2, 4, 8, 16, 32, 64):
        r.render(W, H, samples=spp, max_depth=depth, seed=1)
        st = r.stats()
        print(name, "spp", spp, round(st["kernel_ms"], 3), "ms", round(st["rays_path"] / st["kernel_ms"] / 1e3, 1), "Mrays/s", flush=True)
