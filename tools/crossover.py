"""Where the level-by-level tracer starts to pay: the 983,040-triangle mesh inside n random spheres, 1080p 16 spp depth 8,
traced by both tracers (device ms of a frame, rays per sample).

    python tools/crossover.py [n ...]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(n):
    sys.path.insert(0, ROOT)
    import simple_raytracer_amd as srt
    from simple_raytracer_amd import scenes
    W, H = 1920, 1080
    r = srt.Renderer(device=0)
    world, lights, cam = scenes.sphere_mesh_scene(W, H, 5, n)
    r.set_scene(world, lights, cam, world_kind=srt.WORLD_SURFACELIST)
    r.render(W, H, samples=4, max_depth=8, seed=1)
    r.render(W, H, samples=16, max_depth=8, seed=1)
    ms = []
    for _ in range(2):
        r.render(W, H, samples=16, max_depth=8, seed=1)
        st = r.stats()
        ms.append(st["kernel_ms"])
    print("%.2f %.3f %d" % (min(ms), st["rays_path"] / (W * H * 16.0), st["wavefront_items"]))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
    else:
        for n in [int(a) for a in sys.argv[1:]] or [0, 30, 100, 300, 1000, 3000, 10000]:
            out = []
            for env in ({"SRT_NO_WAVEFRONT": "1"}, {"SRT_WAVEFRONT_MIN_RAYS": "0"}, {}):
                e = dict(os.environ); e.update(env)
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n)], env=e, capture_output=True, text=True, timeout=300)
                if p.returncode != 0:
                    print("failed", n, p.stderr[-300:]); sys.exit(1)
                out.append(p.stdout.strip().splitlines()[-1].split())
            print("%6d spheres: %.2f rays per sample; one kernel %8.2f ms, level by level %8.2f ms (%.2f x); the library's own choice %8.2f ms (%s)" % (
                n, float(out[0][1]), float(out[0][0]), float(out[1][0]), float(out[0][0]) / float(out[1][0]), float(out[2][0]),
                "level by level" if int(out[2][2]) else "one kernel"), flush=True)
