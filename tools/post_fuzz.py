"""Random pictures, sizes and parameters through the three post-process filters on the GPU and through the oracle:
identical pictures expected.

    python tools/post_fuzz.py [cases] [seed]
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import simple_raytracer_amd as srt  # noqa: E402
from oracle import orc  # noqa: E402  (the checker)
from test_post_filters import random_image  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
r = srt.Renderer(device=0)
bad = 0
for case in range(n):
    h, w = rng.choice([(rng.randint(1, 12), rng.randint(1, 12)), (rng.randint(3, 400), rng.randint(3, 400)), (rng.randint(120, 1500), rng.randint(3, 60))])
    img = random_image(h, w, rng.randint(0, 1 << 30), bright=rng.choice([0.02, 0.1, 0.5, 1.0]))
    weight = rng.choice([1.0, 0.5, 0.15, rng.uniform(-0.2, 1.3)])
    which = rng.choice(["despeckle", "blur", "contrast"])
    passes = rng.choice([1, 1, 2, 3])
    if which == "despeckle":
        t = rng.choice([200.0, 120.0, rng.uniform(0, 260)])
        a, b = r.despeckle(img, weight, t, passes=passes), img
        for _ in range(passes):
            b = orc.Despeckle(b, weight, t)
    elif which == "blur":
        a, b = r.blur_array(img, weight, passes=passes), img
        for _ in range(passes):
            b = orc.BlurArray(b, weight)
    else:
        c = rng.choice([0.0, 5.0, 40.0, rng.uniform(0, 120)])
        a, b = r.contrast_blur_array(img, weight, c, passes=passes), img
        for _ in range(passes):
            b = orc.ContrastBlurArray(b, weight, c)
    if not (a == b).all():
        bad += 1
        print("MISMATCH", case, which, h, w, weight, passes, int((a != b).sum()), flush=True)
print("cases", n, "mismatches", bad)
sys.exit(1 if bad else 0)
