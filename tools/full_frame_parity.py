"""One-off check: the whole bench frame (Game1 scene, 1920x1080, 16 spp, depth 8) from the GPU
against the CPU oracle, both world kinds.  Takes about 40 s of CPU time on 16 threads.

    python tools/full_frame_parity.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import helpers  # noqa: E402
import simple_raytracer_amd as srt  # noqa: E402
from oracle import orc  # noqa: E402  (checker)

W, H, SPP, DEPTH = 1920, 1080, 16, 8
r = srt.Renderer(device=0)
spec = helpers.default_spec()
world, lights, cam = spec.host(W, H)
for kind, okind, name in ((srt.WORLD_BVHNODE, orc.WORLD_BVHNODE, "bvhnode"), (srt.WORLD_SURFACELIST, orc.WORLD_SURFACELIST, "surfacelist")):
    r.set_scene(world, lights, cam, world_kind=kind)
    img = r.render(W, H, samples=SPP, max_depth=DEPTH, seed=1)
    st = r.stats()
    ref, ost = spec.oracle_scene().render(spec.oracle_camera(W, H), W, H, samples=SPP, max_depth=DEPTH, seed=1, world_kind=okind,
                                          threads=min(16, os.cpu_count()))
    lsb, npx = helpers.max_lsb_diff(img, ref)
    print("%s: max difference %d LSB, %d of %d pixels differ; rays %d (gpu) / %d (oracle); gpu %.2f ms, oracle %.1f s on %d threads"
          % (name, lsb, npx, W * H, st["rays_path"], ost["rays_path"], st["kernel_ms"], ost["seconds"], ost["threads_used"]), flush=True)
