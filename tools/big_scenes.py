"""Measures the mesh benchmark configurations on one GPU (bench.py stays on the sphere scene).

    python tools/big_scenes.py [c3] [c5] [c4] [--spp N] [--cpu]

c3: teapot.dae x16 (101,120 triangles), 1920x1080, depth 4
c5: sphere.dae x1024 (983,040 triangles), 1920x1080 here (the 8K frame of the config is 16 such frames), depth 8
c4: c5's mesh + 10,000 random spheres, depth 8
--cpu times the CPU oracle (the reference's algorithm incl. its own BVH) on a small window of the same frame.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import simple_raytracer_amd as srt  # noqa: E402
from simple_raytracer_amd import scenes  # noqa: E402
from simple_raytracer_amd.distributed import tile_number  # noqa: E402

W, H = 1920, 1080
CONFIGS = {
    "c3": (lambda: scenes.teapot_scene(W, H, 2), 4),
    "c5": (lambda: scenes.sphere_mesh_scene(W, H, 5), 8),
    "c4": (lambda: scenes.sphere_mesh_scene(W, H, 5, 10000), 8),
}


def main():
    args = [a for a in sys.argv[1:] if a in CONFIGS] or ["c3", "c5"]
    spp = int(sys.argv[sys.argv.index("--spp") + 1]) if "--spp" in sys.argv else 4
    r = srt.Renderer(device=0)
    for name in args:
        make, depth = CONFIGS[name]
        world, lights, cam = make()
        t = time.time()
        r.set_scene(world, lights, cam, world_kind=srt.WORLD_SURFACELIST)
        commit_s = time.time() - t
        if "--once" in sys.argv:      # one render and nothing else: the run that counter passes profile
            r.render(W, H, samples=spp, max_depth=depth, seed=1)
            print(json.dumps({"config": name, "spp": spp, "kernel_ms": round(r.stats()["kernel_ms"], 3)}), flush=True)
            continue
        r.render(W, H, samples=1, max_depth=depth, seed=1)   # warm-up
        img = r.render(W, H, samples=spp, max_depth=depth, seed=1)
        st = r.stats()
        r.render(W, H, samples=spp, max_depth=depth, seed=1, counted=True)
        cst = r.stats()
        q = cst["rays_path"] + cst["rays_shadow"]
        line = {"config": name, "triangles_and_spheres": st["n_prims"], "bvh_nodes": st["bvh_nodes"], "bvh_depth": st["bvh_depth"],
                "commit_s": round(commit_s, 2), "spp": spp, "depth": depth, "kernel_ms": round(st["kernel_ms"], 3),
                "Mrays_s": round(st["rays_path"] / st["kernel_ms"] / 1e3, 1),
                "Mqueries_s": round((st["rays_path"] + st["rays_shadow"]) / st["kernel_ms"] / 1e3, 1),
                "box_tests_per_query": round(cst["box_tests"] / q, 2), "tri_tests_per_query": round(cst["tri_tests"] / q, 2),
                "sphere_tests_per_query": round(cst["sphere_tests"] / q, 2),
                "algorithmic_GBs": round((cst["box_tests"] * 32 + cst["tri_tests"] * 36 + cst["sphere_tests"] * 16 + cst["hit_nodes"] * 72
                                          + cst["rays_shadow"] * 28) / st["kernel_ms"] / 1e6, 1),
                "nonblack_fraction": round(float(((img & 0xFFFFFF) != 0).mean()), 3)}
        names = ["deal", "return", "closest", "on_closest", "aux", "shadow_setup", "shadow_query", "light_eval", "child_dirs", "fresnel",
                 "child_start"]
        tot = float(sum(cst["section_cycles"])) or 1.0
        line["cycle_share"] = {n: round(c / tot, 3) for n, c in zip(names, cst["section_cycles"])}
        line["subtrees_handed_over"] = cst["section_cycles"][15]
        line["walk_quanta"] = cst["trace_sections"]
        line["lanes_per_trace_section"] = round(cst["trace_lanes"] / max(1, cst["trace_sections"]), 1)
        line["lanes_per_light_section"] = round(cst["light_lanes"] / max(1, cst["light_sections"]), 1)
        if "--cpu" in sys.argv:
            from oracle import orc   # CPU baseline only
            import helpers
            spec_world = world.flatten()
            mats = [orc.material(m.albedo, m.metalness, m.smoothness, m.emission) for m in spec_world["materials"]]
            tris = orc.tris_from_arrays(spec_world["tri_pos"], spec_world["tri_nrm"], 0)
            tris["material"] = spec_world["tri_mat"]
            sph = [orc.Sphere(orc.f3(s[:3]), float(s[3]), int(m)) for s, m in zip(spec_world["spheres"], spec_world["sphere_mat"])]
            ol = [orc.point_light(l.position, l.brightness, None) for l in lights.lights]
            t = time.time()
            osc = orc.Scene(mats, spheres=sph, tris=tris, lights=ol)
            build_s = time.time() - t
            ocam = orc.Camera(orc.f3(cam.position), orc.f3(cam.lowerLeftCorner), orc.f3(cam.horizontal), orc.f3(cam.vertical))
            tile_x, tile_y = (W // 2) // 32, (H // 2) // 32           # the 32x32 tile at the image centre; 8 of its rows
            win = (tile_x * 32, tile_y * 32, tile_x * 32 + 32, tile_y * 32 + 8)
            ref, ost = osc.render(ocam, W, H, samples=1, max_depth=depth, seed=1, window=win, threads=min(16, os.cpu_count()))
            line.update({"cpu_Mrays_s": round(ost["rays_path"] / ost["seconds"] / 1e6, 5), "cpu_threads": ost["threads_used"],
                         "cpu_sample": "32x8 window, 1 spp, %d rays in %.1f s (reference BVH built in %.1f s)" % (ost["rays_path"], ost["seconds"], build_s),
                         "cpu_box_tests_per_query": round(ost["box_tests"] / max(1, ost["rays_path"] + ost["rays_shadow"] + ost["rays_redundant"]), 1)})
            # the GPU traced as the reference does (its BVH, its order) must reproduce the window exactly
            r.set_scene(world, lights, cam, world_kind=srt.WORLD_BVHNODE)
            # (the reference tree costs 10^4..10^5 box tests per ray: render only the window's tile, as the one tile of
            # "rank tile_id of tiles_x*tiles_y ranks")
            tiles_x, tiles_y = (W + 31) // 32, (H + 31) // 32
            g = r.render(W, H, samples=1, max_depth=depth, seed=1, layout=srt.LAYOUT_TILES, tile_w=32, tile_h=32,
                         rank=tile_number(tile_x, tile_y, tiles_x), world_size=tiles_x * tiles_y).reshape(32, 32)
            x0, y0, x1, y1 = win
            lsb, npx = helpers.max_lsb_diff(g[:8, :], ref[y0:y1, x0:x1])
            line.update({"bvhnode_window_max_lsb": lsb, "bvhnode_window_diff_pixels": npx, "bvhnode_1spp_kernel_ms": round(r.stats()["kernel_ms"], 1)})
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
