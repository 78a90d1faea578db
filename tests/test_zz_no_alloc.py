"""srt_render_device_static: the render-to-device-memory entry point for callers that cannot allocate.  Its sample
buffer, running sums and work counter live in a static per-GPU array instead of hipMalloc'ed buffers; everything else
is srt_render_device, so the two must write the same pixels and count the same rays -- and the call must leave the
device's free memory where it was, also when it is the first render of a context, and it can be recorded into a HIP
graph (stream capture) and replayed."""
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


def test_static_workspace_render_allocates_nothing():
    """The first render of a fresh context through the static entry point -- nothing has been allocated for an earlier
    render that it could reuse -- including a 1920 x 1080 frame that needs several passes and running sums: the device's
    free memory is the same before and after, and the allocating entry point, called afterwards, does take memory."""
    import torch
    torch.cuda.init()
    r = srt.Renderer(device=0)
    world, lights, cam = helpers.default_spec().host(1920, 1080)
    r.set_scene(world, lights, cam)
    stream = torch.cuda.current_stream().cuda_stream
    out = torch.zeros(1920 * 1080, dtype=torch.int32, device="cuda")
    ref = torch.zeros(1920 * 1080, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    for (w, h), kw in [((320, 180), dict(samples=3, max_depth=4, seed=3)), ((1920, 1080), dict(samples=5, max_depth=4, seed=2))]:
        p = r.params(w, h, **kw)
        r.render_device_static(p, out.data_ptr(), w * h, stream)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before, (w, h)
        st = r.stats()      # (collecting the counters allocates nothing either)
        assert torch.cuda.mem_get_info()[0] == free_before
    r.render_device(p, ref.data_ptr(), 1920 * 1080, stream)      # the allocating twin: same pixels, and it does allocate
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] < free_before
    assert torch.equal(out, ref) and r.stats()["rays_path"] == st["rays_path"] > 0
    r.close()


def test_static_workspace_render_can_be_recorded_into_a_graph():
    """Stream capture: the call is recorded into a HIP graph, the graph is replayed twice, and each replay writes the
    picture srt_render_device writes.  Afterwards the context still renders and reports statistics as usual (a launch
    under capture must not leave the context waiting for an event that only exists inside the graph)."""
    import torch
    r = srt.Renderer(device=0)
    w, h = 320, 180
    world, lights, cam = helpers.default_spec().host(w, h)
    r.set_scene(world, lights, cam)
    p = r.params(w, h, samples=4, max_depth=6, seed=21)
    side = torch.cuda.Stream()
    ref = torch.zeros(w * h, dtype=torch.int32, device="cuda")
    out = torch.zeros(w * h, dtype=torch.int32, device="cuda")
    with torch.cuda.stream(side):
        r.render_device_static(p, ref.data_ptr(), w * h, side.cuda_stream)      # warm-up outside the capture, and the expected picture
    side.synchronize()
    st_ref = r.stats()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        r.render_device_static(p, out.data_ptr(), w * h, side.cuda_stream)
    assert int(out.abs().sum().item()) == 0        # recorded, not run
    for _ in range(2):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
    # the context is as usable as before: a blocking render, then the counters (the two replays are in them)
    img = r.render(w, h, samples=4, max_depth=6, seed=21)
    assert np.array_equal(img.ravel().view(np.int32), ref.cpu().numpy())
    st = r.stats()
    assert st["rays_path"] in (st_ref["rays_path"], 3 * st_ref["rays_path"])
    r.close()
