"""Random render parameters (size, samples, depth, seed, world kind, scene, tile layout) rendered on the GPU and by
the CPU harness of the kernel source (tests/emu): packed images and ray counts must be identical.

    python tools/gpu_fuzz.py [cases] [seed]
"""
import random
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import helpers  # noqa: E402
import simple_raytracer_amd as srt  # noqa: E402

SPECS = {"default": helpers.default_spec, "glossy": helpers.glossy_spec, "rect": lambda: helpers.default_spec(with_rect=True),
         "mesh": lambda: helpers.mesh_spec(n=6)}


def run(gpu, emu, n_cases, seed, verbose=False):
    """-> list of descriptions of the cases that differ."""
    rng = random.Random(seed)
    bad = []
    for case in range(n_cases):
        name = rng.choice(list(SPECS))
        kind = srt.WORLD_SURFACELIST if name == "rect" else rng.choice([srt.WORLD_BVHNODE, srt.WORLD_SURFACELIST])
        w, h = rng.choice([(rng.randint(1, 40), rng.randint(1, 30)), (rng.randint(40, 200), rng.randint(30, 120))])
        spp = rng.choice([1, 1, 2, 3, 5, 8])
        depth = rng.choice([0, 1, 2, 4, 6, 8, 8, 10, 13, 16])
        if name == "glossy":
            depth = min(depth, 6)
        if name == "mesh":
            depth = min(depth, 8)
        kw = dict(samples=spp, max_depth=depth, seed=rng.randint(0, 2 ** 31 - 1), color_clamp=rng.choice([1.0, 1.0, 0.5]))
        world = rng.choice([1, 1, 2, 3])
        if world > 1:   # one rank's tile buffer; small pictures leave some ranks without any tile
            tile = rng.choice([(8, 8), (16, 8), (32, 32)])
            kw.update(layout=srt.LAYOUT_TILES, tile_w=tile[0], tile_h=tile[1], rank=rng.randrange(world), world_size=world)
        scene = SPECS[name]().host(w, h)
        gpu.set_scene(*scene, world_kind=kind)
        emu.set_scene(*scene, world_kind=kind)
        a, b = gpu.render(w, h, **kw), emu.render(w, h, **kw)
        sa, sb = gpu.stats(), emu.stats()
        same = a.shape == b.shape and bool((a == b).all()) and all(sa[k] == sb[k] for k in ("rays_path", "rays_shadow", "hit_nodes"))
        if not same:
            bad.append((case, name, kind, w, h, kw, int((a != b).sum()) if a.shape == b.shape else "shape"))
            print("MISMATCH", bad[-1], flush=True)
        elif verbose and case % 20 == 0:
            print("ok", case, name, w, h, spp, depth, world, flush=True)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    bad = run(srt.Renderer(device=0), srt.Renderer(library=helpers.EMU_LIB, strict=False), n, int(sys.argv[2]) if len(sys.argv) > 2 else 1, True)
    print("cases", n, "mismatches", len(bad))
    sys.exit(1 if bad else 0)
