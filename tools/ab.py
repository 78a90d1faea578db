"""A/B timing of library variants on the benchmark scenes (device time of the trace + resolve kernels, 1080p).

    python tools/ab.py [--configs c2,c3,c5,c4] [--reps 3] <library.so | default> ...

Every library runs in a process of its own (SRT_LIBRARY is read at import); per configuration the first render after
the commit is a warm-up (it also calibrates the traversal quantum), then `reps` timed renders: min and median in ms.
c2: 16 spp depth 8; c3: 64 spp depth 4; c5: 64 spp depth 8; c4: 16 spp depth 8.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(configs, reps):
    sys.path.insert(0, ROOT)
    import numpy as np
    import simple_raytracer_amd as srt
    from simple_raytracer_amd import scenes, game1
    W, H = 1920, 1080
    table = {
        "c2": (lambda: game1.default_scene(W, H), srt.WORLD_BVHNODE, 16, 8),
        "c3": (lambda: scenes.teapot_scene(W, H, 2), srt.WORLD_SURFACELIST, 64, 4),
        "c5": (lambda: scenes.sphere_mesh_scene(W, H, 5), srt.WORLD_SURFACELIST, 64, 8),
        "c4": (lambda: scenes.sphere_mesh_scene(W, H, 5, 10000), srt.WORLD_SURFACELIST, 16, 8),
    }
    r = srt.Renderer(device=0)
    out = {}
    for name in configs:
        make, kind, spp, depth = table[name]
        world, lights, cam = make()
        r.set_scene(world, lights, cam, world_kind=kind)
        r.render(W, H, samples=min(spp, 4), max_depth=depth, seed=1)
        r.render(W, H, samples=spp, max_depth=depth, seed=1)
        ms = []
        img = None
        for _ in range(reps):
            img = r.render(W, H, samples=spp, max_depth=depth, seed=1)
            st = r.stats()
            ms.append(st["kernel_ms"])
        out[name] = {"min": round(min(ms), 3), "med": round(float(np.median(ms)), 3), "rays": st["rays_path"], "tune": st["walk_tune"], "wf_nodes_per_ray": round(st["wavefront_nodes"] / max(st["rays_path"], 1), 3), "wf_given_up": st["wavefront_given_up"], "wf_twins": st["wavefront_twins"],
                     "crc": int(np.bitwise_xor.reduce(img.ravel().astype(np.uint64) * np.arange(1, img.size + 1, dtype=np.uint64)) & np.uint64(0xFFFFFFFF))}
    print(json.dumps(out), flush=True)


def main():
    args = sys.argv[1:]
    if args and args[0] == "--child":
        child(args[1].split(","), int(args[2]))
        return
    configs, reps, libs = "c3,c5,c4", 3, []
    i = 0
    while i < len(args):
        if args[i] == "--configs":
            configs = args[i + 1]; i += 2
        elif args[i] == "--reps":
            reps = int(args[i + 1]); i += 2
        else:
            libs.append(args[i]); i += 1
    for lib in libs or ["default"]:
        env = dict(os.environ)
        label = lib
        if "=" in lib and not lib.endswith(".so"):
            # NAME=VALUE[,NAME=VALUE...]@library: environment knobs for this run
            knobs, _, lib = lib.partition("@")
            for kv in knobs.split(","):
                k, _, v = kv.partition("=")
                env[k] = v
            lib = lib or "default"
        if lib != "default":
            env["SRT_LIBRARY"] = os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", configs, str(reps)], env=env, capture_output=True, text=True, timeout=int(os.environ.get("AB_TIMEOUT", "300")))
        except subprocess.TimeoutExpired:
            print(label, "TIMEOUT", flush=True)
            sys.exit(1)
        line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else "FAILED rc=%d %s" % (p.returncode, p.stderr[-400:])
        print(label, line, flush=True)
        if p.returncode != 0:
            sys.exit(1)      # a failed GPU run: start nothing more


if __name__ == "__main__":
    main()
