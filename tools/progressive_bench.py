"""Device time of a progressive render (srt_render_progressive) of the default scene, 1920x1080, 16 spp, depth 8, in
2, 4 and 16 passes, against one call of srt_render.  (Passes on alternating streams, so that a pass starts while the
previous one ends, were measured with this tool: 17.1 / 19.7 / 40.7 ms against 18.1 / 21.1 / 30.8 ms one after the
other -- two persistent kernels that compete for the same slots -- and not kept.)

    python tools/progressive_bench.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simple_raytracer_amd as srt  # noqa: E402
from simple_raytracer_amd import game1  # noqa: E402

r = srt.Renderer(device=0)
w, l, c = game1.default_scene(1920, 1080)
r.set_scene(w, l, c)
ref = r.render(1920, 1080, samples=16, max_depth=8, seed=1)
print("one call of srt_render:", round(r.stats()["kernel_ms"], 2), "ms")
for passes in (2, 4, 16):
    for _ in range(2):
        p = r.params(1920, 1080, samples=16, max_depth=8, seed=1)
        out = np.zeros((1080, 1920), dtype=np.uint32)
        r.render_progressive(p, out, passes)
        r.wait()
        st = r.stats()
    print(passes, "passes:", round(st["kernel_ms"], 2), "ms, final frame identical:", bool((out == ref).all()), flush=True)
