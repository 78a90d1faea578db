"""One-off check: larger renders of the mesh test scenes (the reference's teapot with spheres, its sphere mesh subdivided
twice) from the GPU against the oracle's linear SurfaceList.hit scan, depth 8 -- images and ray counts must be equal.

    python tools/mesh_parity_check.py
"""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, helpers, mesh_cases
import simple_raytracer_amd as srt
from oracle import orc
r = srt.Renderer(device=0)
for case in [("teapot-big", mesh_cases.teapot_with_spheres, 480, 270, 2, 8, 11), ("sphere-mesh-big", lambda w, h: mesh_cases.sphere_mesh(w, h, 2), 320, 180, 2, 8, 5)]:
    lsb, npx, st, ost = mesh_cases.run(r, orc, case)
    print(case[0], "max lsb", lsb, "diff pixels", npx, "rays", st["rays_path"], ost["rays_path"], "hits", st["hit_nodes"], ost["hit_nodes"], "kernel ms", round(st["kernel_ms"], 2), "oracle s", round(ost["seconds"], 1), flush=True)
