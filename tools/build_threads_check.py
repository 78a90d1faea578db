"""The tree of a large mesh built on one host thread and on several: same picture, same node count, same number of
box and triangle tests (the counting build), and the time of each build.

    python tools/build_threads_check.py [c3|c5|c4 ...]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402

import simple_raytracer_amd as srt  # noqa: E402
from big_scenes import CONFIGS  # noqa: E402


def main():
    r = srt.Renderer(device=0)
    bad = 0
    for name in [a for a in sys.argv[1:] if a in CONFIGS] or ["c3", "c5"]:
        make, depth = CONFIGS[name]
        world, lights, cam = make()
        seen = []
        for fork_min in ("0", None):
            if fork_min is None:
                os.environ.pop("SRT_BUILD_FORK_MIN", None)
            else:
                os.environ["SRT_BUILD_FORK_MIN"] = fork_min
            t = time.time()
            r.set_scene(world, lights, cam, world_kind=srt.WORLD_SURFACELIST)
            dt = time.time() - t
            img = r.render(480, 270, samples=2, max_depth=depth, seed=5, counted=True)
            st = r.stats()
            seen.append((img, st["bvh_nodes"], st["bvh_depth"], st["box_tests"], st["tri_tests"], st["rays_path"]))
            print(name, "one thread" if fork_min == "0" else "threads", "set_scene %.3f s" % dt, seen[-1][1:], flush=True)
        same = np.array_equal(seen[0][0], seen[1][0]) and seen[0][1:] == seen[1][1:]
        print(name, "identical" if same else "DIFFERENT", flush=True)
        bad += not same
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
