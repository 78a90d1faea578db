"""Time of one Despeckle / BlurArray / ContrastBlurArray pass over a 1920x1080 picture: the library's kernels (time
per pass from the difference between 11 passes and 1 pass in one call, so that upload, download and allocation
cancel) against the oracle's sequential loop on one host core (the reference's filter is sequential too).

    python tools/post_bench.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import simple_raytracer_amd as srt  # noqa: E402
from oracle import orc  # noqa: E402  (the CPU side of the comparison)
from test_post_filters import random_image  # noqa: E402

r = srt.Renderer(device=0)
img = random_image(1080, 1920, 1)
cases = [("Despeckle", lambda p: r.despeckle(img, 1.0, 200.0, passes=p), lambda: orc.Despeckle(img, 1.0, 200.0)),
         ("BlurArray", lambda p: r.blur_array(img, 0.5, passes=p), lambda: orc.BlurArray(img, 0.5)),
         ("ContrastBlurArray", lambda p: r.contrast_blur_array(img, 0.5, 30.0, passes=p), lambda: orc.ContrastBlurArray(img, 0.5, 30.0))]
for name, gpu, cpu in cases:
    gpu(1)
    t = time.perf_counter(); gpu(1); t1 = time.perf_counter() - t
    t = time.perf_counter(); gpu(11); t11 = time.perf_counter() - t
    t = time.perf_counter(); cpu(); tc = time.perf_counter() - t
    per_pass = (t11 - t1) / 10
    print("%-18s GPU %.2f ms per pass (call with 1 pass incl. transfers: %.2f ms), oracle on one core %.1f ms; %.1f Mpixel/s, %.2f GB/s of 12 B/pixel"
          % (name, per_pass * 1e3, t1 * 1e3, tc * 1e3, 1920 * 1080 / per_pass / 1e6, 1920 * 1080 * 12 / per_pass / 1e9), flush=True)
