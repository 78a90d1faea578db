import sys, json
sys.path.insert(0, '.')
from simple_raytracer_amd import Renderer
from simple_raytracer_amd.game1 import default_scene
import simple_raytracer_amd as S
for kind in ("bvhnode", "surfacelist"):
    r = Renderer()
    world, lights, cam = default_scene(1920, 1080)
    r.set_scene(world, lights, cam, world_kind=S._native.WORLD_BVHNODE if kind == "bvhnode" else S._native.WORLD_SURFACELIST)
    img = r.render(1920, 1080, samples=16, max_depth=8, seed=1, counted=True)
    st = r.stats()
    print(kind, json.dumps(st))
    cy = st["section_cycles"]; tot = sum(cy) or 1
    names = ["deal", "return", "closest", "on_closest", "aux", "shadow_setup", "shadow_query", "light_eval", "child_dirs", "fresnel", "child_start"]
    print(" cycles share: " + "  ".join("%s %.3f" % (n, c / tot) for n, c in zip(names, cy)))
    print(" trace avg lanes", st["trace_lanes"]/max(1,st["trace_sections"]), "light avg lanes", st["light_lanes"]/max(1,st["light_sections"]))
