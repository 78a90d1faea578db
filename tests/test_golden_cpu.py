"""The committed golden fixtures (tests/golden/*.npz) must still be what the oracle renders, and what the kernel
source renders when run on the CPU."""
import os

import numpy as np
import pytest

import helpers
import simple_raytracer_amd as srt
from golden.make_golden import GOLDEN, POST_NAME, render, render_post


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_oracle_reproduces_golden(oracle, name):
    g = np.load(os.path.join(helpers.GOLDEN, name + ".npz"))
    img, nodes, counts = render(GOLDEN[name])
    assert (img == g["image"]).all()
    assert (nodes == g["nodes_per_pixel"]).all()
    assert counts.tolist() == g["counts"].tolist()
    assert int(g["nodes_per_pixel"].sum()) == int(g["counts"][0])   # rays_path = number of ray-tree nodes


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_kernel_source_reproduces_golden(emu_lib, name):
    g = np.load(os.path.join(helpers.GOLDEN, name + ".npz"))
    factory, kwargs, w, h, spp, depth, mode, kind, seed = GOLDEN[name]
    world, lights, cam = getattr(helpers, factory)(**kwargs).host(w, h)
    r = srt.Renderer(library=emu_lib, strict=False)
    r.set_scene(world, lights, cam, world_kind=kind)
    img = r.render(w, h, samples=spp, max_depth=depth, seed=seed, mode=mode)
    assert (img == g["image"]).all()
    if mode == srt.MODE_COLOR:
        st = r.stats()
        assert [st["rays_path"], st["rays_shadow"], st["hit_nodes"]] == g["counts"].tolist()


def test_oracle_reproduces_post_golden(oracle):
    g = np.load(os.path.join(helpers.GOLDEN, POST_NAME + ".npz"))
    now = render_post()
    for key in ("despeckle", "blur", "contrast_blur", "mix"):
        assert (now[key] == g[key]).all(), key
    assert (g["despeckle"] != g["picture"]).any() and (g["blur"] != g["picture"]).any()
