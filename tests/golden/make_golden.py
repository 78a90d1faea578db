"""Writes tests/golden/*.npz: images rendered by the CPU oracle on fixed seeds.

The reference program has no golden vectors of its own, so these are outputs of the oracle (see
oracle/srt_oracle.cpp).  They make two checks possible without re-running anything: that a later oracle
build still produces them (CPU suite), and that the GPU library produces them on a machine that has
neither the oracle's sources nor the reference (GPU suite).

Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers  # noqa: E402
from oracle import orc  # noqa: E402

# name -> (spec factory name, kwargs, width, height, samples, max_depth, mode, world_kind, seed)
GOLDEN = {
    "default_normal_64x36": ("default_spec", {}, 64, 36, 1, 4, orc.MODE_NORMAL, orc.WORLD_BVHNODE, 1),
    "default_depth_64x36": ("default_spec", {}, 64, 36, 1, 4, orc.MODE_DEPTH, orc.WORLD_BVHNODE, 1),
    "default_color_1spp_64x36": ("default_spec", {}, 64, 36, 1, 4, orc.MODE_COLOR, orc.WORLD_BVHNODE, 1),
    "default_color_16spp_64x36": ("default_spec", {}, 64, 36, 16, 4, orc.MODE_COLOR, orc.WORLD_BVHNODE, 1),
    "default_color_4spp_d8_64x36": ("default_spec", {}, 64, 36, 4, 8, orc.MODE_COLOR, orc.WORLD_BVHNODE, 1),
    "default_rect_list_2spp_64x36": ("default_spec", {"with_rect": True}, 64, 36, 2, 4, orc.MODE_COLOR, orc.WORLD_SURFACELIST, 1),
    "mesh_color_1spp_bvhnode_64x36": ("mesh_spec", {"n": 10}, 64, 36, 1, 4, orc.MODE_COLOR, orc.WORLD_BVHNODE, 1),
    "mesh_color_4spp_list_64x36": ("mesh_spec", {"n": 10, "with_rect": True}, 64, 36, 4, 6, orc.MODE_COLOR, orc.WORLD_SURFACELIST, 1),
    "teapot_dae_normal_list_64x36": ("teapot_spec", {}, 64, 36, 1, 4, orc.MODE_NORMAL, orc.WORLD_SURFACELIST, 1),
    "teapot_dae_color_1spp_list_64x36": ("teapot_spec", {}, 64, 36, 1, 4, orc.MODE_COLOR, orc.WORLD_SURFACELIST, 1),
    "teapot_dae_color_4spp_bvhnode_64x36": ("teapot_spec", {}, 64, 36, 4, 4, orc.MODE_COLOR, orc.WORLD_BVHNODE, 1),
    "tiny_obj_color_4spp_bvhnode_64x36": ("obj_spec", {}, 64, 36, 4, 4, orc.MODE_COLOR, orc.WORLD_BVHNODE, 1),
    "tiny_obj_normal_list_64x36": ("obj_spec", {}, 64, 36, 1, 4, orc.MODE_NORMAL, orc.WORLD_SURFACELIST, 1),
}


def render(entry):
    factory, kwargs, w, h, spp, depth, mode, kind, seed = entry
    spec = getattr(helpers, factory)(**kwargs)
    img, st, nodes = spec.oracle_scene().render(spec.oracle_camera(w, h), w, h, samples=spp, max_depth=depth, seed=seed,
                                                mode=mode, world_kind=kind, want_nodes=True)
    return img, nodes, np.array([st["rays_path"], st["rays_shadow"], st["hit_nodes"]], dtype=np.uint64)


# The steps after the render: one pass of each post-process filter over the 16-spp golden picture, and the mix-buffer
# technique (3 buffers, samples = 2) on the default scene.
POST_NAME = "post_filters_and_mix_buffers_64x36"
POST_ARGS = {"despeckle": (1.0, 120.0), "blur": (0.5,), "contrast_blur": (0.7, 20.0), "mix": (3, 2, 4, 1)}   # mix: buffers, samples, depth, seed


def render_post():
    picture = np.load(os.path.join(HERE, "default_color_16spp_64x36.npz"))["image"]
    spec = helpers.default_spec()
    mixed, _ = spec.oracle_scene().render_mix_buffers(spec.oracle_camera(64, 36), 64, 36, POST_ARGS["mix"][0], samples=POST_ARGS["mix"][1],
                                                      max_depth=POST_ARGS["mix"][2], seed=POST_ARGS["mix"][3])
    return {"picture": picture, "despeckle": orc.Despeckle(picture, *POST_ARGS["despeckle"]), "blur": orc.BlurArray(picture, *POST_ARGS["blur"]),
            "contrast_blur": orc.ContrastBlurArray(picture, *POST_ARGS["contrast_blur"]), "mix": mixed}


if __name__ == "__main__":
    only = sys.argv[1:]          # optional: names to (re)generate
    if not only or POST_NAME in only:
        np.savez_compressed(os.path.join(HERE, POST_NAME + ".npz"), **render_post())
        print(POST_NAME)
    for name, entry in GOLDEN.items():
        if only and name not in only:
            continue
        img, nodes, counts = render(entry)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), image=img, nodes_per_pixel=nodes, counts=counts)
        print(name, img.shape, counts.tolist())
