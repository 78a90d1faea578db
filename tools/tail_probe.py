"""Lane utilisation of the default scene's trace sections against samples per pixel (counting build)."""
import sys

sys.path.insert(0, ".")
import simple_raytracer_amd as srt  # noqa: E402
from simple_raytracer_amd import game1  # noqa: E402

r = srt.Renderer(device=0)
w, l, c = game1.default_scene(1920, 1080)
r.set_scene(w, l, c)
r.render(1920, 1080, samples=1, max_depth=8, seed=1)
for depth in (4, 8):
    for spp in (1, 4, 16):
        r.render(1920, 1080, counted=True, samples=spp, max_depth=depth, seed=1)
        st = r.stats()
        print("depth", depth, "spp", spp, round(st["kernel_ms"], 3), "ms",
              "trace sections", st["trace_sections"], "lanes/section", round(st["trace_lanes"] / max(1, st["trace_sections"]), 1),
              "light sections", st["light_sections"], "lanes/section", round(st["light_lanes"] / max(1, st["light_sections"]), 1),
              "nodes", st["hit_nodes"], flush=True)
        sc = st["section_cycles"]
        print("   tail after the counter ran out: max", sc[11] / 100.0, "us, mean over waves", round(sc[12] / 100.0 / 3072, 1), "us, waves > 250 us:", sc[13], " > 1 ms:", sc[14], "subtrees handed over:", sc[15], flush=True)
