import sys; sys.path.insert(0, ".")
import simple_raytracer_amd as srt
from simple_raytracer_amd import game1
r = srt.Renderer(device=0)
w, l, c = game1.default_scene(1920, 1080)
r.set_scene(w, l, c)
r.render(1920, 1080, samples=1, max_depth=8, seed=1)
for spp in (1, 2, 4, 8, 16, 64):
    r.render(1920, 1080, samples=spp, max_depth=8, seed=1)
    st = r.stats()
    print(spp, round(st["kernel_ms"], 3), round(st["rays_path"] / st["kernel_ms"] / 1e3, 1), "Mrays/s", flush=True)
