"""Where the lanes of the stackless trace kernel spend their time (counting build, srt_render_counted):
lane-cycles of work per part and lane-cycles spent waiting for the other lanes of the wavefront.

    python tools/lane_profile.py [surfacelist] [--spp N]
"""
import json
import sys

sys.path.insert(0, '.')
import simple_raytracer_amd as S
from simple_raytracer_amd.game1 import default_scene

NAMES = ["deal", "return", "closest", "on_closest", "aux", "shadow_setup", "shadow_query", "light_eval", "child_dirs", "fresnel",
         "child_start", "wait_return", "wait_deal", "wait_trace", "wait_light"]


def main():
    kind = S.WORLD_SURFACELIST if "surfacelist" in sys.argv else S.WORLD_BVHNODE
    spp = int(sys.argv[sys.argv.index("--spp") + 1]) if "--spp" in sys.argv else 16
    r = S.Renderer()
    world, lights, cam = default_scene(1920, 1080)
    r.set_scene(world, lights, cam, world_kind=kind)
    r.render(1920, 1080, samples=spp, max_depth=8, seed=1, counted=True)
    st = r.stats()
    lc = st["section_lane_cycles"][:15]
    tot = float(sum(lc)) or 1.0
    out = {"kernel_ms_counting_build": round(st["kernel_ms"], 2),
           "lane_time_share": {n: round(c / tot, 4) for n, c in zip(NAMES, lc)},
           "working_share": round(sum(lc[:11]) / tot, 4),
           "lanes_per_trace_section": round(st["trace_lanes"] / max(1, st["trace_sections"]), 1),
           "lanes_per_light_section": round(st["light_lanes"] / max(1, st["light_sections"]), 1),
           "sections": [st["trace_sections"], st["light_sections"]]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
