"""Frame time against samples per pixel on the default scene: the intercept is what a frame pays once (launch, ramp-up,
the tail of the last ray trees), the slope is the steady-state cost of a sample.

    python tools/spp_sweep.py [max_depth ...]
"""
import sys

sys.path.insert(0, ".")
import simple_raytracer_amd as srt  # noqa: E402
from simple_raytracer_amd import game1  # noqa: E402

r = srt.Renderer(device=0)
w, l, c = game1.default_scene(1920, 1080)
r.set_scene(w, l, c)
r.render(1920, 1080, samples=1, max_depth=8, seed=1)
for depth in [int(a) for a in sys.argv[1:]] or [8]:
    for spp in (1, 2, 4, 8, 16, 32, 64):
        r.render(1920, 1080, samples=spp, max_depth=depth, seed=1)
        st = r.stats()
        print("depth", depth, "spp", spp, round(st["kernel_ms"], 3), "ms", round(st["rays_path"] / st["kernel_ms"] / 1e3, 1), "Mrays/s", flush=True)
