"""A few larger renders with deep and wide ray trees (many wavefronts in the closing loop at once) on the GPU and by the
CPU harness of the kernel source: identical pictures and ray counts expected.

    python tools/big_fuzz.py
"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import helpers, simple_raytracer_amd as srt
gpu = srt.Renderer(device=0); emu = srt.Renderer(library=helpers.EMU_LIB, strict=False)
cases = [("glossy", helpers.glossy_spec, 640, 360, 2, 6, srt.WORLD_BVHNODE), ("default", helpers.default_spec, 960, 540, 3, 12, srt.WORLD_BVHNODE),
         ("glossy-list", helpers.glossy_spec, 480, 270, 1, 7, srt.WORLD_SURFACELIST), ("mesh", lambda: helpers.mesh_spec(n=12), 640, 360, 2, 8, srt.WORLD_SURFACELIST),
         ("mesh-bvh", lambda: helpers.mesh_spec(n=10), 480, 270, 2, 8, srt.WORLD_BVHNODE)]
def cloud_spec():
    """A mesh inside a cloud of half-metallic spheres: three children per hit, two of them along one ray."""
    import numpy as np
    spec = helpers.mesh_spec(n=8)
    rng = np.random.default_rng(7)
    for _ in range(200):
        c = rng.uniform(-3, 3, 3) + np.array([0, 0, -6.0])
        spec.materials.append((tuple(float(v) for v in rng.uniform(0.3, 1, 3)), float(rng.uniform(0, 1)), float(rng.uniform(0.5, 1)), (0, 0, 0)))
        spec.spheres.append((tuple(float(v) for v in c), float(rng.uniform(0.1, 0.5)), len(spec.materials) - 1))
    return spec
cases.append(("cloud", cloud_spec, 480, 270, 3, 8, srt.WORLD_SURFACELIST))
bad = 0
for name, f, w, h, spp, d, kind in cases:
    sc = f().host(w, h)
    gpu.set_scene(*sc, world_kind=kind); emu.set_scene(*sc, world_kind=kind)
    a = gpu.render(w, h, samples=spp, max_depth=d, seed=5); sa = gpu.stats()
    b = emu.render(w, h, samples=spp, max_depth=d, seed=5); sb = emu.stats()
    ok = bool((a == b).all()) and sa["rays_path"] == sb["rays_path"] and sa["rays_shadow"] == sb["rays_shadow"]
    bad += 0 if ok else 1
    print(name, w, h, spp, d, "ok" if ok else "MISMATCH %d px" % int((a != b).sum()), "rays", sa["rays_path"], "gpu ms", round(sa["kernel_ms"], 2), flush=True)
sys.exit(bad)
