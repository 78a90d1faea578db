"""Per-wavefront clock records of a -DSRT_TIMELINE build of the library (make -C simple_raytracer_amd/csrc timeline -> tools/_timeline/libsrt.so): when each
wavefront started, found the item counter exhausted, and finished.

    SRT_LIBRARY=tools/_timeline/libsrt.so SRT_TIMELINE_FILE=/tmp/tl.bin python tools/timeline.py [c3|c4|c5] [spp ...]
"""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
import simple_raytracer_amd as srt  # noqa: E402
from simple_raytracer_amd import game1  # noqa: E402

path = os.environ["SRT_TIMELINE_FILE"]
r = srt.Renderer(device=0)
mesh = [a for a in sys.argv[1:] if a in ("c3", "c4", "c5")]
depth = int(os.environ.get("SRT_TL_DEPTH", "8"))
if mesh:
    sys.path.insert(0, "tools")
    from big_scenes import CONFIGS
    make, depth = CONFIGS[mesh[0]]
    w, l, c = make()
    r.set_scene(w, l, c, world_kind=srt.WORLD_SURFACELIST)
else:
    w, l, c = game1.default_scene(1920, 1080)
    r.set_scene(w, l, c)
r.render(1920, 1080, samples=1, max_depth=depth, seed=1)
r.stats()
for spp in [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 16]:
    r.render(1920, 1080, samples=spp, max_depth=depth, seed=1)
    st = r.stats()
    rec = np.fromfile(path, dtype=np.uint64).reshape(-1, 4).astype(np.int64)
    rec = rec[rec[:, 2] >= rec[:, 0]][: (3072 if not mesh else 4096)]
    rec = rec[rec[:, 0] > 0]
    t0 = rec[:, 0].min()
    start, dry, end = (rec[:, 0] - t0) / 100.0, (rec[:, 1] - t0) / 100.0, (rec[:, 2] - t0) / 100.0   # us
    main_trips, tail_trips = rec[:, 3] >> 32, rec[:, 3] & 0xFFFFFFFF
    q = lambda a: " ".join("%8.1f" % v for v in np.percentile(a, [0, 10, 50, 90, 99, 100]))
    print("spp", spp, "kernel", round(st["kernel_ms"], 3), "ms;   percentiles 0 10 50 90 99 100 (us)")
    print("   start      ", q(start))
    print("   dry        ", q(dry))
    print("   exit       ", q(end))
    print("   exit - dry ", q(end - dry))
    print("   main trips ", q(main_trips), "  tail trips", q(tail_trips))
    print("   us per main trip", round(float((dry - start).sum() / max(1, main_trips.sum())), 2),
          " us per tail trip", round(float((end - dry).sum() / max(1, tail_trips.sum())), 2), flush=True)
