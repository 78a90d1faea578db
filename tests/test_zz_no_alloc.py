"""srt_render_device_static: the render-to-device-memory entry point for callers that cannot allocate.  Its sample
buffer, running sums and work counter live in a static per-GPU array instead of hipMalloc'ed buffers; everything else
is srt_render_device, so the two must write the same pixels and count the same rays."""
import ctypes as C

import numpy as np
import pytest

import helpers
import simple_raytracer_amd as srt

pytestmark = pytest.mark.gpu


def _both(r, torch, w, h, **kw):
    p = r.params(w, h, **kw)
    n = r.local_pixels(p)
    stream = torch.cuda.current_stream().cuda_stream
    a = torch.full((n,), 0x55, dtype=torch.int32, device="cuda")
    b = torch.full((n,), 0x77, dtype=torch.int32, device="cuda")
    r.stats()
    r.render_device(p, a.data_ptr(), n, stream)
    st_a = r.stats()
    r.render_device_static(p, b.data_ptr(), n, stream)
    st_b = r.stats()
    torch.cuda.synchronize()
    return a.cpu().numpy(), b.cpu().numpy(), st_a, st_b


def test_static_workspace_render_equals_the_allocating_one():
    import torch
    r = srt.Renderer(device=0)
    # two calls in a row on one context, the second one smaller: the array keeps the first call's contents
    for (w, h), kw in [((640, 360), dict(samples=6, max_depth=8, seed=11)),
                       ((200, 120), dict(samples=3, max_depth=4, seed=12, layout=srt.LAYOUT_TILES, tile_w=32, tile_h=16, rank=1, world_size=3))]:
        world, lights, cam = helpers.default_spec().host(w, h)
        r.set_scene(world, lights, cam)
        a, b, st_a, st_b = _both(r, torch, w, h, **kw)
        assert np.array_equal(a, b)
        for k in ("rays_path", "rays_shadow", "hit_nodes"):
            assert st_a[k] == st_b[k] and st_a[k] > 0, k
    # a mesh scene (the traversal kernel), and a frame that only fits the array in several passes with running sums:
    # 1920 x 1080 x 12 bytes = 24.9 MB per sample, 64 MiB hold two samples per pass next to the sums
    world, lights, cam = helpers.mesh_spec().host(160, 90)
    r.set_scene(world, lights, cam, world_kind=srt.WORLD_SURFACELIST)
    a, b, st_a, st_b = _both(r, torch, 160, 90, samples=2, max_depth=4, seed=5)
    assert np.array_equal(a, b) and st_a["rays_path"] == st_b["rays_path"]
    world, lights, cam = helpers.default_spec().host(1920, 1080)
    r.set_scene(world, lights, cam)
    a, b, st_a, st_b = _both(r, torch, 1920, 1080, samples=5, max_depth=4, seed=2)
    assert np.array_equal(a, b) and st_a["rays_path"] == st_b["rays_path"]
    # beyond the array: a clear refusal, nothing launched
    p = r.params(2560, 1440, samples=2, max_depth=2)
    big = torch.zeros(2560 * 1440, dtype=torch.int32, device="cuda")
    with pytest.raises(srt.SrtError) as e:
        r.render_device_static(p, big.data_ptr(), big.numel(), torch.cuda.current_stream().cuda_stream)
    assert e.value.code == srt._native.E_LIMIT and "static workspace" in str(e.value)
    r.close()
